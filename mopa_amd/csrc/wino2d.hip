// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the image branch with >= 128 channels (layer2-4 of the ResNet34
// encoder, decoder stages 3-4): 2.25x fewer multiplies than the direct implicit GEMM; the 16 GEMMs run on the same kernel (f32 MFMA by default).
// Three steps, NHWC fp32:
//   1. k_wino_in :  V[p][t][ci] = (B^T d B)[p]   per 2x2-output tile t and channel (d = the tile's 4x4 input patch, zero padded)
//   2. 16 GEMMs  :  M[p] = V[p] (T x Cin) @ U[p] (Cin x Cout)   -- mopa_conv2d_igemm_batched (the implicit-GEMM kernel as a 1x1 conv)
//   3. k_wino_out:  out tile = A^T M[.][t][co] A  (+ bias, or accumulated into out)
// with U[p][ci][co] = (G g G^T)[p] from k_wino_w (cached per weight version by the caller).  Backward-data of such a conv is the
// same pipeline on the output gradient with the 180-degree-rotated, transposed filter.  The 64-channel full-resolution
// layers stay on the direct kernel: there the V / M traffic (4x the activations, written and read) costs more than the
// saved FMAs.  Replaces cuDNN's choice of algorithm behind mopa/models/resnet34_unet.py:97-110; oracle: oracle/net2d.py.
#include "common.h"

// U[p][r][c], p = 4*i + j.  dgrad = 0: r = input channel, c = output channel, g = w[c][r][.][.] (OIHW);
//                           dgrad = 1: r = output channel, c = input channel, g = w[r][c] rotated by 180 degrees.
__global__ void k_wino_w(const float* __restrict__ w, int O, int I, int dgrad, float* __restrict__ U) {
  const int R = dgrad ? O : I, C = dgrad ? I : O;
  const int n = R * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int r = i / C, c = i - r * C;
    const int o = dgrad ? r : c, ci = dgrad ? c : r;
    const float* g9 = w + ((int64_t)o * I + ci) * 9;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = dgrad ? g9[(2 - a) * 3 + (2 - b)] : g9[a * 3 + b];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
      U[(int64_t)(a * 4 + 0) * n + i] = u0;
      U[(int64_t)(a * 4 + 1) * n + i] = u1;
      U[(int64_t)(a * 4 + 2) * n + i] = u2;
      U[(int64_t)(a * 4 + 3) * n + i] = u3;
    }
  }
}

// thread = (tile, channel quad)
__global__ __launch_bounds__(256) void k_wino_in(const float* __restrict__ in, int ld_in, int B, int H, int W, int C, int th, int tw,
                                                  float* __restrict__ V) {
  const int CQ = C >> 2;
  const int64_t T = (int64_t)B * th * tw;
  const int64_t total = T * CQ;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / CQ;
    const int cq = (int)(i - t * CQ);
    const int b = (int)(t / (th * tw));
    const int rt = (int)(t - (int64_t)b * th * tw);
    const int ty = rt / tw, tx = rt - ty * tw;
    float4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int y = 2 * ty - 1 + a;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int x = 2 * tx - 1 + c;
        d[a][c] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                      ? *reinterpret_cast<const float4*>(in + ((int64_t)(b * H + y) * W + x) * ld_in + cq * 4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#define F4OP(R, X, OP, Y) R.x = X.x OP Y.x; R.y = X.y OP Y.y; R.z = X.z OP Y.z; R.w = X.w OP Y.w
    float4 m[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // B^T d
      F4OP(m[0][c], d[0][c], -, d[2][c]);
      F4OP(m[1][c], d[1][c], +, d[2][c]);
      F4OP(m[2][c], d[2][c], -, d[1][c]);
      F4OP(m[3][c], d[1][c], -, d[3][c]);
    }
    float* vp = V + t * C + cq * 4;
    const int64_t ps = T * C;
#pragma unroll
    for (int a = 0; a < 4; ++a) {  // (.) B
      float4 v0, v1, v2, v3;
      F4OP(v0, m[a][0], -, m[a][2]);
      F4OP(v1, m[a][1], +, m[a][2]);
      F4OP(v2, m[a][2], -, m[a][1]);
      F4OP(v3, m[a][1], -, m[a][3]);
      *reinterpret_cast<float4*>(vp + (a * 4 + 0) * ps) = v0;
      *reinterpret_cast<float4*>(vp + (a * 4 + 1) * ps) = v1;
      *reinterpret_cast<float4*>(vp + (a * 4 + 2) * ps) = v2;
      *reinterpret_cast<float4*>(vp + (a * 4 + 3) * ps) = v3;
    }
  }
}

__global__ __launch_bounds__(256) void k_wino_out(const float* __restrict__ M, int B, int H, int W, int C, int th, int tw,
                                                   const float* __restrict__ bias, float* __restrict__ out, int ld_out, int accumulate) {
  const int CQ = C >> 2;
  const int64_t T = (int64_t)B * th * tw;
  const int64_t total = T * CQ;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / CQ;
    const int cq = (int)(i - t * CQ);
    const int b = (int)(t / (th * tw));
    const int rt = (int)(t - (int64_t)b * th * tw);
    const int ty = rt / tw, tx = rt - ty * tw;
    const float* mp = M + t * C + cq * 4;
    const int64_t ps = T * C;
    float4 m[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) m[a][c] = *reinterpret_cast<const float4*>(mp + (a * 4 + c) * ps);
    float4 s[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // A^T m
      s[0][c].x = m[0][c].x + m[1][c].x + m[2][c].x; s[0][c].y = m[0][c].y + m[1][c].y + m[2][c].y;
      s[0][c].z = m[0][c].z + m[1][c].z + m[2][c].z; s[0][c].w = m[0][c].w + m[1][c].w + m[2][c].w;
      s[1][c].x = m[1][c].x - m[2][c].x - m[3][c].x; s[1][c].y = m[1][c].y - m[2][c].y - m[3][c].y;
      s[1][c].z = m[1][c].z - m[2][c].z - m[3][c].z; s[1][c].w = m[1][c].w - m[2][c].w - m[3][c].w;
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + cq * 4);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int y = 2 * ty + a;
      if (y >= H) continue;
      float4 y0, y1;  // (.) A
      y0.x = s[a][0].x + s[a][1].x + s[a][2].x + bv.x; y0.y = s[a][0].y + s[a][1].y + s[a][2].y + bv.y;
      y0.z = s[a][0].z + s[a][1].z + s[a][2].z + bv.z; y0.w = s[a][0].w + s[a][1].w + s[a][2].w + bv.w;
      y1.x = s[a][1].x - s[a][2].x - s[a][3].x + bv.x; y1.y = s[a][1].y - s[a][2].y - s[a][3].y + bv.y;
      y1.z = s[a][1].z - s[a][2].z - s[a][3].z + bv.z; y1.w = s[a][1].w - s[a][2].w - s[a][3].w + bv.w;
      float4* p0 = reinterpret_cast<float4*>(out + ((int64_t)(b * H + y) * W + 2 * tx) * ld_out + cq * 4);
      if (accumulate) { const float4 q = *p0; y0.x += q.x; y0.y += q.y; y0.z += q.z; y0.w += q.w; }
      *p0 = y0;
      if (2 * tx + 1 < W) {
        float4* p1 = reinterpret_cast<float4*>(out + ((int64_t)(b * H + y) * W + 2 * tx + 1) * ld_out + cq * 4);
        if (accumulate) { const float4 q = *p1; y1.x += q.x; y1.y += q.y; y1.z += q.z; y1.w += q.w; }
        *p1 = y1;
      }
    }
  }
}

// Weight gradient in the Winograd domain: dU[p] = V[p]^T dM[p] with dM = A dY A^T per 2x2 output tile (A = (A^T)^T, 4x2).
// thread = (tile, channel quad); output-gradient pixels beyond H, W are zero.
__global__ __launch_bounds__(256) void k_wino_dout(const float* __restrict__ dy, int ld, int B, int H, int W, int C, int th, int tw,
                                                    float* __restrict__ dM) {
  const int CQ = C >> 2;
  const int64_t T = (int64_t)B * th * tw;
  const int64_t total = T * CQ;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / CQ;
    const int cq = (int)(i - t * CQ);
    const int b = (int)(t / (th * tw));
    const int rt = (int)(t - (int64_t)b * th * tw);
    const int ty = rt / tw, tx = rt - ty * tw;
    float4 d[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int y = 2 * ty + a, x = 2 * tx + c;
        d[a][c] = (y < H && x < W) ? *reinterpret_cast<const float4*>(dy + ((int64_t)(b * H + y) * W + x) * ld + cq * 4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    float4 r[4][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // A dY
      r[0][c] = d[0][c];
      F4OP(r[1][c], d[0][c], +, d[1][c]);
      F4OP(r[2][c], d[0][c], -, d[1][c]);
      r[3][c] = make_float4(-d[1][c].x, -d[1][c].y, -d[1][c].z, -d[1][c].w);
    }
    float* mp = dM + t * C + cq * 4;
    const int64_t ps = T * C;
#pragma unroll
    for (int a = 0; a < 4; ++a) {  // (.) A^T
      float4 m1, m2;
      F4OP(m1, r[a][0], +, r[a][1]);
      F4OP(m2, r[a][0], -, r[a][1]);
      *reinterpret_cast<float4*>(mp + (a * 4 + 0) * ps) = r[a][0];
      *reinterpret_cast<float4*>(mp + (a * 4 + 1) * ps) = m1;
      *reinterpret_cast<float4*>(mp + (a * 4 + 2) * ps) = m2;
      *reinterpret_cast<float4*>(mp + (a * 4 + 3) * ps) = make_float4(-r[a][1].x, -r[a][1].y, -r[a][1].z, -r[a][1].w);
    }
  }
}
#undef F4OP

// U (16 * Cin * Cout floats) from the OIHW weight of a 3x3 convolution (dgrad = 1: filter of its backward-data).
MOPA_API int mopa_wino_weight(const float* weight, int32_t O, int32_t I, int32_t dgrad, float* U, void* stream) {
  if (O <= 0 || I <= 0) return MOPA_ERR_ARG;
  k_wino_w<<<stream_grid((int64_t)O * I, 256), 256, 0, (hipStream_t)stream>>>(weight, O, I, dgrad, U);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
// V (16 * T * C floats, T = B * ceil(H/2) * ceil(W/2)) from the NHWC input (row stride ld_in); padding 1.
MOPA_API int mopa_wino_input(const float* in, int32_t ld_in, int32_t B, int32_t H, int32_t W, int32_t C, float* V, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ld_in < C || (ld_in & 3) || (((uintptr_t)in | (uintptr_t)V) & 15)) return MOPA_ERR_ARG;
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  k_wino_in<<<stream_grid((int64_t)B * th * tw * (C >> 2), 256), 256, 0, (hipStream_t)stream>>>(in, ld_in, B, H, W, C, th, tw, V);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
// out (NHWC, row stride ld_out) = A^T M A (+ bias) (+= if accumulate) from M (16 * T * C floats).
MOPA_API int mopa_wino_output(const float* M, int32_t B, int32_t H, int32_t W, int32_t C, const float* bias, float* out, int32_t ld_out,
                              int32_t accumulate, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ld_out < C || (ld_out & 3) || (((uintptr_t)out | (uintptr_t)M) & 15)) return MOPA_ERR_ARG;
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  k_wino_out<<<stream_grid((int64_t)B * th * tw * (C >> 2), 256), 256, 0, (hipStream_t)stream>>>(M, B, H, W, C, th, tw, bias, out, ld_out, accumulate);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
// dM (16 * T * C floats) = A dY A^T from the NHWC output gradient (row stride ld) -- the second operand of the Winograd-domain
// weight gradient (mopa_wino_bwd_weight in conv2d.hip; the first is V from mopa_wino_input on the layer input).
MOPA_API int mopa_wino_dout(const float* dy, int32_t ld, int32_t B, int32_t H, int32_t W, int32_t C, float* dM, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ld < C || (ld & 3) || (((uintptr_t)dy | (uintptr_t)dM) & 15)) return MOPA_ERR_ARG;
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  k_wino_dout<<<stream_grid((int64_t)B * th * tw * (C >> 2), 256), 256, 0, (hipStream_t)stream>>>(dy, ld, B, H, W, C, th, tw, dM);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
