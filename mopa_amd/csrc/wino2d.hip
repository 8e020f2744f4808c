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
#include "weight_forms.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

// U[p][r][c], p = 4*i + j.  dgrad = 0: r = input channel, c = output channel, g = w[c][r][.][.] (OIHW);
//                           dgrad = 1: r = output channel, c = input channel, g = w[r][c] rotated by 180 degrees.
__global__ void k_wino_w(const float* __restrict__ w, int O, int I, int dgrad, float* __restrict__ U) {
  wf_wino2_body(w, O, I, dgrad, U, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
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

// ================================================================================================================
// Winograd F(4x4, 3x3): 6x6 input patch -> 4x4 outputs, 36 transform points, 4x fewer multiplies than the direct conv (F(2x2):
// 2.25x) and LESS transform traffic (V and M are 36/16 = 2.25x the activations instead of 4x).  Lavin & Gray's matrices:
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// fp32 error is about 10x that of the direct sum (still ~1e-6 relative per layer); used where it pays (dense2d.wino_tile).
// Thread = (tile, channel): consecutive lanes are consecutive channels, every access is a coalesced 256-byte run.  (2 or 4
// channels per thread were measured: the same times within 3 % -- the transforms move 3.25x the tensor and sit at 4.3-4.7 TB/s.)
// t = B^T d with every multiply-add written as ONE fused operation: the scalar instantiation (k_wino4_in, with or without the BatchNorm
// on the way in) and the vector ones (k_wino4_conv) then round alike whatever the compiler would have contracted in each context --
// "V has the same bits on every path" is a tested property (15 operations).
#include "wino4.h"

// U[p][r][c] (p = 6*i + j) = (G g G^T)[p]; dgrad as in k_wino_w
__global__ void k_wino4_w(const float* __restrict__ w, int O, int I, int dgrad, float* __restrict__ U, int transpose) {
  wf_wino4_body(w, O, I, dgrad, U, transpose, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// BN: the input is the raw output of the previous convolution and BatchNorm + ReLU are applied on the way in -- y = relu(x * scale +
// shift) with scale / shift of the image's BatchNorm group from stats[G][4][C] (mopa_bn_act_fwd_groups with y == null), the very
// expression of k_bn_relu_apply, so V is bit-identical to transforming the materialised y; the zero padding stays zero.
template <bool BN>
__global__ __launch_bounds__(256) void k_wino4_in(const float* __restrict__ in, int ld_in, int B, int H, int W, int C, int th, int tw,
                                                   float* __restrict__ V, const float* __restrict__ stats, int imgs_per_group, int c0) {
  const int64_t T = (int64_t)B * th * tw;
  const int64_t total = T * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / C;
    const int ch = (int)(i - t * C);
    const int b = (int)(t / (th * tw));
    const int rt = (int)(t - (int64_t)b * th * tw);
    const int ty = rt / tw, tx = rt - ty * tw;
    float sc = 1.f, sh = 0.f;
    // channels [c0, C) are the BatchNorm's (stats rows of C - c0 values); channels below c0 -- the skip half of a decoder join buffer,
    // already relu(batchnorm(.)) -- are left alone (a wave's 64 lanes are 64 consecutive channels: the branch is uniform for c0 % 64 == 0)
    const bool bn_ch = BN && ch >= c0;
    if (bn_ch) {
      const float* __restrict__ sg = stats + (int64_t)(b / imgs_per_group) * 4 * (C - c0);
      sc = sg[ch - c0]; sh = sg[C - c0 + ch - c0];
    }
    float m[6][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {   // B^T d, column by column
      const int x = 4 * tx - 1 + c;
      float d[6], tc[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int y = 4 * ty - 1 + a;
        const bool inside = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        d[a] = inside ? in[((int64_t)(b * H + y) * W + x) * ld_in + ch] : 0.f;
        if (bn_ch && inside) {
          const float o = fmaf(d[a], sc, sh);
          d[a] = o > 0.f ? o : o * 0.f;
        }
      }
      w4_bt(d, tc);
#pragma unroll
      for (int a = 0; a < 6; ++a) m[a][c] = tc[a];
    }
    float* vp = V + t * C + ch;
    const int64_t ps = T * C;
#pragma unroll
    for (int a = 0; a < 6; ++a) {   // (.) B
      float v[6];
      w4_bt(m[a], v);
#pragma unroll
      for (int c = 0; c < 6; ++c) vp[(a * 6 + c) * ps] = v[c];
    }
  }
}

__global__ __launch_bounds__(256) void k_wino4_out(const float* __restrict__ M, int B, int H, int W, int C, int th, int tw,
                                                    const float* __restrict__ bias, float* __restrict__ out, int ld_out, int accumulate) {
  const int64_t T = (int64_t)B * th * tw;
  const int64_t total = T * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / C;
    const int ch = (int)(i - t * C);
    const int b = (int)(t / (th * tw));
    const int rt = (int)(t - (int64_t)b * th * tw);
    const int ty = rt / tw, tx = rt - ty * tw;
    const float* mp = M + t * C + ch;
    const int64_t ps = T * C;
    float s[4][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {   // A^T m, column by column
      float m[6], y[4];
#pragma unroll
      for (int a = 0; a < 6; ++a) m[a] = mp[(a * 6 + c) * ps];
      w4_at(m, y);
#pragma unroll
      for (int a = 0; a < 4; ++a) s[a][c] = y[a];
    }
    const float bv = bias ? bias[ch] : 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {   // (.) A
      const int y = 4 * ty + a;
      if (y >= H) continue;
      float o[4];
      w4_at(s[a], o);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int x = 4 * tx + c;
        if (x >= W) continue;
        float* p = out + ((int64_t)(b * H + y) * W + x) * ld_out + ch;
        float v = o[c] + bv;
        if (accumulate) v += *p;
        *p = v;
      }
    }
  }
}

// dM[p][t][co] = (A dY A^T)[p] per 4x4 output-gradient tile (zero beyond H, W): second operand of the transform-domain weight gradient
__global__ __launch_bounds__(256) void k_wino4_dout(const float* __restrict__ dy, int ld, int B, int H, int W, int C, int th, int tw,
                                                     float* __restrict__ dM) {
  const int64_t T = (int64_t)B * th * tw;
  const int64_t total = T * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / C;
    const int ch = (int)(i - t * C);
    const int b = (int)(t / (th * tw));
    const int rt = (int)(t - (int64_t)b * th * tw);
    const int ty = rt / tw, tx = rt - ty * tw;
    float r[6][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // A dY, column by column
      const int x = 4 * tx + c;
      float d[4], rc[6];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int y = 4 * ty + a;
        d[a] = (y < H && x < W) ? dy[((int64_t)(b * H + y) * W + x) * ld + ch] : 0.f;
      }
      w4_a(d, rc);
#pragma unroll
      for (int a = 0; a < 6; ++a) r[a][c] = rc[a];
    }
    float* mp = dM + t * C + ch;
    const int64_t ps = T * C;
#pragma unroll
    for (int a = 0; a < 6; ++a) {   // (.) A^T
      float m[6];
      w4_a(r[a], m);
#pragma unroll
      for (int c = 0; c < 6; ++c) mp[(a * 6 + c) * ps] = m[c];
    }
  }
}

// F(4x4,3x3) entry points: same contracts as mopa_wino_weight / _input / _output / _dout with 36 transform points and
// T = B * ceil(H/4) * ceil(W/4) tiles.
MOPA_API int mopa_wino4_weight(const float* weight, int32_t O, int32_t I, int32_t dgrad, float* U, void* stream) {
  if (O <= 0 || I <= 0) return MOPA_ERR_ARG;
  k_wino4_w<<<stream_grid((int64_t)O * I, 256), 256, 0, (hipStream_t)stream>>>(weight, O, I, dgrad, U, 0);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_wino4_input(const float* in, int32_t ld_in, int32_t B, int32_t H, int32_t W, int32_t C, float* V, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ld_in < C) return MOPA_ERR_ARG;
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  k_wino4_in<false><<<stream_grid((int64_t)B * th * tw * C, 256), 256, 0, (hipStream_t)stream>>>(in, ld_in, B, H, W, C, th, tw, V, nullptr, 1, 0);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
// V of relu(batchnorm(in)): `in` is the BatchNorm's INPUT, stats = [n_groups][4][C] as mopa_bn_act_fwd_groups wrote it (scale, shift,
// mean, invstd), the B images are n_groups equal consecutive groups.  Bit-identical to mopa_wino4_input on the applied tensor.
// bn_c0 > 0: only channels [bn_c0, C) belong to the BatchNorm (stats = [n_groups][4][C - bn_c0]); the channels below are a tensor that is
// non-negative already and pass through -- a decoder join buffer [skip | raw up-convolution] whose second half is normalised on the way in.
MOPA_API int mopa_wino4_input_bn(const float* in, int32_t ld_in, int32_t B, int32_t H, int32_t W, int32_t C, const float* stats,
                                 int32_t n_groups, int32_t bn_c0, float* V, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ld_in < C || !stats || n_groups < 1 || B % n_groups || bn_c0 < 0 || bn_c0 >= C) return MOPA_ERR_ARG;
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  k_wino4_in<true><<<stream_grid((int64_t)B * th * tw * C, 256), 256, 0, (hipStream_t)stream>>>(in, ld_in, B, H, W, C, th, tw, V, stats,
                                                                                              B / n_groups, bn_c0);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_wino4_output(const float* M, int32_t B, int32_t H, int32_t W, int32_t C, const float* bias, float* out, int32_t ld_out,
                               int32_t accumulate, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ld_out < C) return MOPA_ERR_ARG;
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  k_wino4_out<<<stream_grid((int64_t)B * th * tw * C, 256), 256, 0, (hipStream_t)stream>>>(M, B, H, W, C, th, tw, bias, out, ld_out, accumulate);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_wino4_dout(const float* dy, int32_t ld, int32_t B, int32_t H, int32_t W, int32_t C, float* dM, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ld < C) return MOPA_ERR_ARG;
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  k_wino4_dout<<<stream_grid((int64_t)B * th * tw * C, 256), 256, 0, (hipStream_t)stream>>>(dy, ld, B, H, W, C, th, tw, dM);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ================================================================================================================
// Every stale weight form of the 2D network in ONE launch: after an optimizer step the first use of each conv re-laid out
// its weight with a tiny kernel of its own (igemm layouts, Winograd transforms: ~95 launches of 8-10 us per joint step, serialised
// on the main stream between the convolutions).  desc_host [n][8] int64: source, destination, O, I, KH, KW, kind (0 = igemm
// layout, 1 = F(2x2) transform, 2 = F(4x4) transform), arg (kind 0: mode 0-3; kinds 1, 2: bit 0 = dgrad, bits 1-2 = 1 transposed / 2 fragment layout of
// mopa_wino4_conv / 3 fragment layout of mopa_wino4_conv9 (F(4x4) only)).
#define WF_MAX 48
struct WeightFormDescs { int64_t src[WF_MAX], dst[WF_MAX]; int32_t O[WF_MAX], I[WF_MAX], KH[WF_MAX], KW[WF_MAX], kind[WF_MAX], arg[WF_MAX]; };
__global__ void k_weight_forms_batched(const WeightFormDescs d) {
  const int e = blockIdx.y;
  const float* src = reinterpret_cast<const float*>(d.src[e]);
  float* dst = reinterpret_cast<float*>(d.dst[e]);
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x, istep = gridDim.x * blockDim.x;
  if (d.kind[e] == 0) wf_relayout_body(src, dst, d.O[e], d.I[e], d.KH[e], d.KW[e], d.arg[e], i0, istep);
  else if (d.kind[e] == 1) wf_wino2_body(src, d.O[e], d.I[e], d.arg[e] & 1, dst, i0, istep);
  else wf_wino4_body(src, d.O[e], d.I[e], d.arg[e] & 1, dst, (d.arg[e] >> 1) & 3, i0, istep);
}
MOPA_API int mopa_conv2d_weight_forms_batched(const int64_t* desc_host, int32_t n, void* stream) {
  if (!desc_host || n <= 0 || n > WF_MAX) return MOPA_ERR_ARG;
  WeightFormDescs d;
  memset(&d, 0, sizeof(d));
  int64_t nmax = 0;
  for (int e = 0; e < n; ++e) {
    const int64_t* r = desc_host + (int64_t)e * 8;
    const int64_t O = r[2], I = r[3], KH = r[4], KW = r[5], kind = r[6], arg = r[7];
    if (!r[0] || !r[1] || O <= 0 || I <= 0 || kind < 0 || kind > 2) return MOPA_ERR_ARG;
    if (kind == 0 && (KH <= 0 || KW <= 0 || arg < 0 || arg > 3 || O * I * KH * KW >= (1ll << 31))) return MOPA_ERR_ARG;
    if (kind != 0 && (KH != 3 || KW != 3 || arg < 0 || arg > 7 || (kind == 1 && (arg & 6))
                      || ((arg & 6) == 4 && (O % 16 || I % 16))
                      || ((arg & 6) == 6 && (((arg & 1) ? O : I) % 16 || ((arg & 1) ? I : O) % 32)))) return MOPA_ERR_ARG;   // fragment layouts 2 / 3
    d.src[e] = r[0]; d.dst[e] = r[1]; d.O[e] = (int)O; d.I[e] = (int)I; d.KH[e] = (int)KH; d.KW[e] = (int)KW; d.kind[e] = (int)kind; d.arg[e] = (int)arg;
    const int64_t ne = kind == 0 ? O * I * KH * KW : O * I;
    if (ne > nmax) nmax = ne;
  }
  int bx = (int)cdiv64(nmax, 256);
  if (bx > 256) bx = 256;
  k_weight_forms_batched<<<dim3(bx, n), 256, 0, (hipStream_t)stream>>>(d);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ================================================================================================================
// F(4x4): the 36 GEMMs with the OUTPUT TRANSFORM IN THE EPILOGUE -- M (2.25x the activations) is never written or read.
// Why: at 64-128 channels the batched GEMM is bound by V in + M out (4.2 TB/s at 67 TFLOP/s for 64 -> 64 at 152x240) and the
// output transform re-reads M; together 129 us of that layer's 200 (profiles/by_layer_2d.py).
// How: a block owns 64 tiles x 32 output channels and walks the 36 transform points one after the other with an LDS-staged K
// loop per point (A = V[p] rows, B = U^T[p] rows, both k-contiguous).  The output transform is linear,
// Y = A^T M A = sum_i AT[.][i] (sum_j m[i][j] AT[.][j]): after point (i, j) the wave folds its 8 (tile, channel) results into a
// 4-value row sum, after the 6th j the row sum goes into the 16 output values of the pair -- 128 + 32 registers per lane
// instead of 36 x 8 accumulators.
// Waves: 2 (tiles) x 2 (channels), each 32 x 16 = two v_mfma_f32_16x16x4_f32 tiles; k is permuted consistently for both
// operands (MFMA step (g, s) takes k = 16 g + 4 q + s from lane group q) so one b128 LDS read feeds 4 MFMAs.
// Blocks are numbered so that the channel groups of one tile range run on the same XCD back to back (they share V through L2).
// Measured (profiles/bench_wino_fused.py, B = 8, whole conv incl. the input transform, batched GEMM + output transform -> fused):
//   64 -> 128 at 304x480  1175 -> 781 us    128 -> 64 at 304x480  1111 -> 927     64 -> 128 at 152x240  316 -> 234
//   64 -> 64 at 152x240    189 -> 175       128 -> 64 at 152x240   291 -> 328     128^2 at 76x120  109 -> 134, deeper: worse
// In-kernel cycle counters (-DW4G_PROFILE) per unit (point x 64-channel chunk) of a lone block: the 32 MFMAs per wave ~600-680,
// fragment-read waits + epilogue + drain ~750-900, DMA issue 340-830 (six LDS-DMA instructions per wave; it grows with the load on
// the memory pipeline), barrier + vmcnt < 200: the DMA's latency (~2 us) is hidden by the two units of lookahead, what is left is
// issue-side.  One wave per SIMD issues v_mfma_f32_16x16x4_f32 at the full rate (profiles/micro/mfma_rate.hip: 144-155 TFLOP/s),
// so the remaining headroom is in overlapping those segments, and with 72 KB of LDS only two blocks share a CU.  The kernel wins
// on grids of at least two full rounds of blocks (>= 1024) and, at 64 input channels, from one round on; it loses to the batched
// GEMM on ~570-block grids with more K per point (a 58-block tail round) and on the short deep levels: dense2d.wino4_fused.
typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef float f32x2w __attribute__((ext_vector_type(2)));
// A^T of F(4x4,3x3), transposed: c_w4_at[j][c] = AT[c][j]
__constant__ float c_w4_at[6][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f}, {1.f, 2.f, 4.f, 8.f}, {1.f, -2.f, 4.f, -8.f}, {0.f, 0.f, 0.f, 1.f}};
#define W4G_BM 64
#define W4G_BN 32
#define W4G_KB 64
#define W4G_NST 3   // LDS ring: the unit being multiplied + two in flight (LDS-DMA, no staging registers)

// Staging = global_load_lds_dwordx4 (LDS-DMA): one wave instruction moves 4 rows x 256 B into 1 KiB of LDS, lane-linear.  Rows are
// 256 B, so unpadded b128 fragment reads of 16 rows would hit one bank group 16 times: the 16-byte chunk c of row R lives in slot
// c ^ (R & 15) -- applied on the SOURCE address when staging (the LDS image of a DMA cannot be scattered) and on the read address.
// A unit = (transform point, 64-channel K chunk).  Per unit and wave: wait until its own DMA of this unit has landed (counted
// vmcnt: the next unit's 6 stay in flight), barrier (everyone's has; everyone is done reading the slot that is overwritten next),
// issue the DMA of unit u + 2, multiply unit u.  At 64 input channels a unit is only 32 MFMAs per wave (0.4 us): one unit of
// prefetch left the kernel waiting on HBM latency for most of every step (170 us at 64 -> 64, 152x240; round-2 first version).
__global__ __launch_bounds__(256, 2) void k_wino4_gemm_out(const float* __restrict__ V, const float* __restrict__ Ut,
                                                            const float* __restrict__ bias, float* __restrict__ out, int ld_out,
                                                            int B, int H, int W, int th, int tw, int Cin, int Cout, int accumulate,
                                                            int mtiles, int ntn, int64_t a_ps, int ldv
#ifdef W4G_PROFILE
                                                            , long long* prof
#endif
                                                            ) {
  __shared__ __attribute__((aligned(1024))) float As[W4G_NST][W4G_BM][W4G_KB];
  __shared__ __attribute__((aligned(1024))) float Bs[W4G_NST][W4G_BN][W4G_KB];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the DMA's LDS targets (M0) become scalar arithmetic
  const int bid = blockIdx.x;
  const int grp = bid / (8 * ntn), within = bid - grp * 8 * ntn;
  const int n_idx = within >> 3, m_idx = grp * 8 + (within & 7);
  if (m_idx >= mtiles) return;
  const int T = B * th * tw;
  const int m0 = m_idx * W4G_BM, n0 = n_idx * W4G_BN;
  const int wm0 = (wv & 1) * 32, wn0 = (wv >> 1) * 16;
  const int nkc = Cin / W4G_KB;
  const int NU = 36 * nkc;
  // DMA assignment: wave wv, instruction j moves rows 4 (4 j + wv) .. + 3; lane = (row in group: lane >> 4, slot: lane & 15).
  // Byte offsets inside one transform point stay below 2^32 (launcher): uniform 64-bit base + 32-bit lane offset.
  unsigned a_off[4], b_off[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 4 * (4 * j + wv) + (lane >> 4);
    a_off[j] = (unsigned)(((int64_t)min(m0 + row, T - 1) * ldv + (((lane & 15) ^ (row & 15)) << 2)) * 4);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 4 * (4 * j + wv) + (lane >> 4);
    b_off[j] = (unsigned)(((int64_t)(n0 + row) * Cin + (((lane & 15) ^ (row & 15)) << 2)) * 4);
  }
  // the unit the next DMA fetches: running pointers (no division per unit), its ring slot, and how many are left
  typedef const __attribute__((address_space(1))) char* gptr_t;
  gptr_t dma_a = (gptr_t)V, dma_b = (gptr_t)Ut;
  const int64_t a_step = (int64_t)W4G_KB * 4, a_wrap = (a_ps - (int64_t)(nkc - 1) * W4G_KB) * 4;
  const int64_t b_wrap = ((int64_t)Cout * Cin - (int64_t)(nkc - 1) * W4G_KB) * 4;
  int dma_kc = 0, dma_left = NU;
  unsigned dma_st = 0;
  const unsigned lds_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&As[0][0][0];
  const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Bs[0][0][0];
  const unsigned dma_la = lds_a + (unsigned)wv * 1024u, dma_lb = lds_b + (unsigned)wv * 1024u;   // + j * 4096 + slot * stage bytes
#define W4G_DMA()                                                                                                     \
  if (dma_left > 0) {                                                                                                 \
    const unsigned la_ = dma_la + dma_st * (W4G_BM * W4G_KB * 4), lb_ = dma_lb + dma_st * (W4G_BN * W4G_KB * 4);      \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                  \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_a + a_off[j_]),            \
                                       (__attribute__((address_space(3))) void*)(uintptr_t)(la_ + j_ * 4096u), 16, 0, 0); \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                  \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_b + b_off[j_]),            \
                                       (__attribute__((address_space(3))) void*)(uintptr_t)(lb_ + j_ * 4096u), 16, 0, 0); \
    --dma_left;                                                                                                       \
    dma_st = dma_st == W4G_NST - 1 ? 0u : dma_st + 1u;                                                                \
    if (++dma_kc == nkc) { dma_kc = 0; dma_a += a_wrap; dma_b += b_wrap; }                                            \
    else { dma_a += a_step; dma_b += a_step; }                                                                        \
  }
  float o[8][16];
#pragma unroll
  for (int pr = 0; pr < 8; ++pr)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[pr][e] = 0.f;

  W4G_DMA();
  W4G_DMA();
  int u = 0;
  unsigned st = 0;   // ring slot of unit u
  // Fragment reads are inline asm: the compiler cannot tell that the slot a DMA is filling is not the slot being read and would
  // put `s_waitcnt vmcnt(0)` in front of every compiler-visible LDS read (draining the ring).  LDS byte addresses: stage base +
  // row * 256 + 16 * ((4 g + q) ^ r); the second row tile is the first + 4096 (offset field).
  const unsigned ar = lds_a + (unsigned)(wm0 + r) * 256u, br = lds_b + (unsigned)(wn0 + r) * 256u;
  unsigned sl[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) sl[g] = (unsigned)(((4 * g + q) ^ r) << 4);
#define W4G_RD(XA0, XA1, XB, G)                                                                  \
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:4096\n\tds_read_b128 %2, %4" \
               : "=&v"(XA0), "=&v"(XA1), "=&v"(XB)                                                \
               : "v"(sa + sl[G]), "v"(sb + sl[G]))
#define W4G_WAIT(N, XA0, XA1, XB) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(XA0), "+v"(XA1), "+v"(XB))
#define W4G_MM(C0, C1, XA0, XA1, XB)                                            \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2) {                            \
    C0 = __builtin_amdgcn_mfma_f32_16x16x4f32(XA0[s2], XB[s2], C0, 0, 0, 0);    \
    C1 = __builtin_amdgcn_mfma_f32_16x16x4f32(XA1[s2], XB[s2], C1, 0, 0, 0);    \
  }
  // The point loop over pi is NOT unrolled (36 copies of the unit body were 60 KB of code: instruction-cache misses on every unit);
  // the transform coefficients are wave-uniform values from a constant table.  Two accumulator sets alternate between points:
  // while point p's MFMAs still drain, the wave is already through the wait / barrier / DMA issue / fragment reads of the next
  // unit, and the epilogue of point p -- fold its 8 results into the row sums, after the 6th point of a row the row sums into the
  // outputs -- runs between the MFMA groups of point p + 1's first unit, out of the OTHER set (no copy, no drain).  In-kernel
  // cycle counters of the first version, per unit of 32 MFMAs (~660): issue of the 6 DMAs ~500-800 (per-unit division, 64-bit
  // VALU address adds, readfirstlane for M0: now running pointers and scalar LDS targets), drain + epilogue ~800-1000.
  f32x4w eA0 = {0.f, 0.f, 0.f, 0.f}, eA1 = eA0, eB0 = eA0, eB1 = eA0;
  float srow_sum[8][4];
#pragma unroll
  for (int pr = 0; pr < 8; ++pr)
#pragma unroll
    for (int c = 0; c < 4; ++c) srow_sum[pr][c] = 0.f;
#ifdef W4G_PROFILE
  long long pt[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define W4G_T(K_) { const long long n_ = __builtin_readcyclecounter(); pt[K_] += n_ - tprev; tprev = n_; }
#else
#define W4G_T(K_)
#endif
#define W4G_UNIT_HEAD()                                                                                   \
  W4G_T(0) /* loop tail since the last MFMA issue */                                                      \
  if (u + 1 < NU) __builtin_amdgcn_s_waitcnt(0x0F76); /* vmcnt(6): this unit has landed, the next one's 6 DMAs may be in flight */ \
  else __builtin_amdgcn_s_waitcnt(0x0F70);                                                                \
  W4G_T(1)                                                                                                \
  __builtin_amdgcn_s_barrier();                                                                           \
  W4G_T(2)                                                                                                \
  W4G_DMA();                                                                                              \
  W4G_T(3)                                                                                                \
  const unsigned sa = ar + st * (W4G_BM * W4G_KB * 4), sb = br + st * (W4G_BN * W4G_KB * 4);              \
  st = st == W4G_NST - 1 ? 0u : st + 1u;                                                                  \
  f32x4w xa0, xa1, xb, ya0, ya1, yb;
#define W4G_ROWSUM(PJ_, P0, P1)                                                                           \
  {                                                                                                       \
    const float t0 = c_w4_at[PJ_][0], t1 = c_w4_at[PJ_][1], t2 = c_w4_at[PJ_][2], t3 = c_w4_at[PJ_][3];   \
    _Pragma("unroll") for (int j4 = 0; j4 < 4; ++j4) {                                                    \
      srow_sum[j4][0] = fmaf(t0, P0[j4], srow_sum[j4][0]);                                                \
      srow_sum[j4][1] = fmaf(t1, P0[j4], srow_sum[j4][1]);                                                \
      srow_sum[j4][2] = fmaf(t2, P0[j4], srow_sum[j4][2]);                                                \
      srow_sum[j4][3] = fmaf(t3, P0[j4], srow_sum[j4][3]);                                                \
      srow_sum[4 + j4][0] = fmaf(t0, P1[j4], srow_sum[4 + j4][0]);                                        \
      srow_sum[4 + j4][1] = fmaf(t1, P1[j4], srow_sum[4 + j4][1]);                                        \
      srow_sum[4 + j4][2] = fmaf(t2, P1[j4], srow_sum[4 + j4][2]);                                        \
      srow_sum[4 + j4][3] = fmaf(t3, P1[j4], srow_sum[4 + j4][3]);                                        \
    }                                                                                                     \
  }
#define W4G_COLSUM(PI_)                                                                                   \
  {                                                                                                       \
    const float w0 = c_w4_at[PI_][0], w1 = c_w4_at[PI_][1], w2 = c_w4_at[PI_][2], w3 = c_w4_at[PI_][3];   \
    _Pragma("unroll") for (int pr = 0; pr < 8; ++pr)                                                      \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                     \
        o[pr][0 + c] = fmaf(w0, srow_sum[pr][c], o[pr][0 + c]);                                           \
        o[pr][4 + c] = fmaf(w1, srow_sum[pr][c], o[pr][4 + c]);                                           \
        o[pr][8 + c] = fmaf(w2, srow_sum[pr][c], o[pr][8 + c]);                                           \
        o[pr][12 + c] = fmaf(w3, srow_sum[pr][c], o[pr][12 + c]);                                         \
        srow_sum[pr][c] = 0.f;                                                                            \
      }                                                                                                   \
  }
// one transform point into the set (C0, C1); (P0, P1) = the other set = the previous point's results
#define W4G_POINT(C0, C1, P0, P1)                                                                         \
  {                                                                                                       \
    {   /* first unit of the point, with the previous point's epilogue between its MFMA groups */        \
      W4G_UNIT_HEAD();                                                                                    \
      W4G_RD(xa0, xa1, xb, 0);                                                                            \
      W4G_RD(ya0, ya1, yb, 1);                                                                            \
      W4G_ROWSUM((pj + 5) % 6, P0, P1);        /* previous point = (pi, pj - 1), or (pi - 1, 5) */        \
      C0 = (f32x4w){0.f, 0.f, 0.f, 0.f};                                                                  \
      C1 = (f32x4w){0.f, 0.f, 0.f, 0.f};                                                                  \
      W4G_WAIT(3, xa0, xa1, xb);                                                                          \
      W4G_T(4)                                                                                            \
      W4G_MM(C0, C1, xa0, xa1, xb);                                                                       \
      W4G_RD(xa0, xa1, xb, 2);                                                                            \
      W4G_WAIT(3, ya0, ya1, yb);                                                                          \
      W4G_MM(C0, C1, ya0, ya1, yb);                                                                       \
      if (pj == 0) W4G_COLSUM((pi + 5) % 6);   /* the previous row of points is complete (zeros before the first) */ \
      W4G_RD(ya0, ya1, yb, 3);                                                                            \
      W4G_WAIT(3, xa0, xa1, xb);                                                                          \
      W4G_MM(C0, C1, xa0, xa1, xb);                                                                       \
      W4G_WAIT(0, ya0, ya1, yb);                                                                          \
      W4G_MM(C0, C1, ya0, ya1, yb);                                                                       \
      W4G_T(5)                                                                                            \
      ++u;                                                                                                \
    }                                                                                                     \
    _Pragma("unroll 1") for (int kc = 1; kc < nkc; ++kc, ++u) {                                           \
      W4G_UNIT_HEAD();                                                                                    \
      W4G_RD(xa0, xa1, xb, 0);                                                                            \
      W4G_RD(ya0, ya1, yb, 1);                                                                            \
      W4G_WAIT(3, xa0, xa1, xb);                                                                          \
      W4G_MM(C0, C1, xa0, xa1, xb);                                                                       \
      W4G_RD(xa0, xa1, xb, 2);                                                                            \
      W4G_WAIT(3, ya0, ya1, yb);                                                                          \
      W4G_MM(C0, C1, ya0, ya1, yb);                                                                       \
      W4G_RD(ya0, ya1, yb, 3);                                                                            \
      W4G_WAIT(3, xa0, xa1, xb);                                                                          \
      W4G_MM(C0, C1, xa0, xa1, xb);                                                                       \
      W4G_WAIT(0, ya0, ya1, yb);                                                                          \
      W4G_MM(C0, C1, ya0, ya1, yb);                                                                       \
    }                                                                                                     \
  }
#pragma unroll 1
  for (int pi = 0; pi < 6; ++pi) {
#pragma unroll
    for (int pj = 0; pj < 6; ++pj) {
      if (pj % 2 == 0) W4G_POINT(eA0, eA1, eB0, eB1)
      else W4G_POINT(eB0, eB1, eA0, eA1)
    }
  }
  W4G_ROWSUM(5, eB0, eB1);   // point (5, 5) went into set B
  W4G_COLSUM(5);
#undef W4G_UNIT_HEAD
#ifdef W4G_PROFILE
  if (blockIdx.x == 0 && tid == 0)
    for (int k = 0; k < 6; ++k) prof[k] = pt[k];
#endif
#undef W4G_ROWSUM
#undef W4G_COLSUM
#undef W4G_POINT
#undef W4G_DMA
#undef W4G_RD
#undef W4G_WAIT
#undef W4G_MM
  const int co = n0 + wn0 + r;
  const float bv = bias ? bias[co] : 0.f;
#pragma unroll
  for (int pr = 0; pr < 8; ++pr) {
    const int tile = m0 + wm0 + 16 * (pr >> 2) + 4 * q + (pr & 3);
    if (tile >= T) continue;
    const int b = tile / (th * tw), rt = tile - b * th * tw;
    const int ty = rt / tw, tx = rt - ty * tw;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int y = 4 * ty + a;
      if (y >= H) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int x = 4 * tx + c;
        if (x >= W) continue;
        float* pp = out + ((int64_t)(b * H + y) * W + x) * ld_out + co;
        float v = o[pr][a * 4 + c] + bv;
        if (accumulate) v += *pp;
        *pp = v;
      }
    }
  }
}

// U^T[36][C][R] (k-contiguous rows for the fused kernel) from the OIHW weight; dgrad as in mopa_wino4_weight.
MOPA_API int mopa_wino4_weight_t(const float* weight, int32_t O, int32_t I, int32_t dgrad, float* Ut, void* stream) {
  if (O <= 0 || I <= 0) return MOPA_ERR_ARG;
  k_wino4_w<<<stream_grid((int64_t)O * I, 256), 256, 0, (hipStream_t)stream>>>(weight, O, I, dgrad, Ut, 1);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// out (NHWC, row stride ld_out) = A^T (V[p] U[p]) A (+ bias) (+= if accumulate): mopa_conv2d_igemm_batched + mopa_wino4_output in
// one kernel.  V: [36][T][Cin] from mopa_wino4_input, Ut: [36][Cout][Cin] from mopa_wino4_weight_t.  Cin % 64 == 0, Cout % 32 == 0.
MOPA_API int mopa_wino4_gemm_output(const float* V, const float* Ut, const float* bias, float* out, int32_t ld_out, int32_t B, int32_t H,
                                    int32_t W, int32_t Cin, int32_t Cout, int32_t accumulate, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % W4G_KB || Cout % W4G_BN || ld_out < Cout) return MOPA_ERR_ARG;
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  const int64_t T = (int64_t)B * th * tw;
  if (T * Cin * 4 >= (1ll << 32) || (int64_t)Cout * Cin * 4 >= (1ll << 32) || T >= (1 << 30)) return MOPA_ERR_ARG;
  const int mtiles = (int)cdiv64(T, W4G_BM), ntn = Cout / W4G_BN;
  const int64_t nblk = cdiv64(mtiles, 8) * 8 * ntn;
#ifdef W4G_PROFILE
  static long long* prof = nullptr;
  if (!prof) hipMallocManaged(&prof, 64);
  k_wino4_gemm_out<<<(unsigned)nblk, 256, 0, (hipStream_t)stream>>>(V, Ut, bias, out, ld_out, B, H, W, th, tw, Cin, Cout, accumulate,
                                                                    mtiles, ntn, T * Cin, Cin, prof);
  hipStreamSynchronize((hipStream_t)stream);
  printf("[w4g profile] block 0 wave 0, cycles per unit: tail %.0f | vmcnt %.0f | barrier %.0f | dma issue %.0f | lds first wait %.0f | mfma groups %.0f  (units %d)\n",
         prof[0] / (36.0 * Cin / 64), prof[1] / (36.0 * Cin / 64), prof[2] / (36.0 * Cin / 64), prof[3] / (36.0 * Cin / 64), prof[4] / (36.0 * Cin / 64), prof[5] / (36.0 * Cin / 64), 36 * Cin / 64);
#else
  k_wino4_gemm_out<<<(unsigned)nblk, 256, 0, (hipStream_t)stream>>>(V, Ut, bias, out, ld_out, B, H, W, th, tw, Cin, Cout, accumulate,
                                                                    mtiles, ntn, T * Cin, Cin);
#endif
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}


// ================================================================================================================
// F(4x4): the WHOLE convolution in one kernel -- input transform, 36 GEMMs, output transform.  Neither V nor M (2.25x the
// activations each) exists in HBM: the layer reads its input once (plus the tile halos, from L2) and writes its output once.
// Why: the 64/128-channel layers at 152x240 and 304x480 are bound by that traffic (k_wino4_in writes V, k_wino4_gemm_out reads it:
// 672 MB per 64 -> 64 layer at 16 x 152 x 240 against 149 MB in + 149 MB out).
// How: a persistent workgroup of four waves, one per CU (144 KB of LDS, one wave per SIMD with the whole 512-register file), walks
// work items = (16 consecutive tiles = one MFMA M dimension) x (64 output channels); an item takes Cin / 64 steps of 64 input channels.
//   transform  wave w takes the 16 input channels [64 r + 16 w, + 16) of step r: lane (tile t = lane % 16, quad q = lane / 16) holds
//              the 6x6 patch of its tile for channels 4 q .. 4 q + 3 (36 dwordx4 loads, a safe address outside the image, zeroed
//              afterwards), applies B^T d B in registers -- the arithmetic of k_wino4_in, V has the same bits -- and writes the 36
//              points lane-linearly into LDS ([chunk][point][lane] float4).  The float4 a lane wrote IS an A-operand fragment:
//              v_mfma_f32_16x16x4_f32 step j takes A[m = t][k = q] = channel 4 q + j -- the k index is a permutation of the chunk's
//              channels, and the weight fragments (mopa_wino4_weight_f) use the same one.
//   multiply   wave w owns output channels [n0 + 16 w, + 16): per chunk and point one ds_read_b128 (A) and one global dwordx4 (B: a
//              1-KiB lane-linear run, L2-resident, two buffers of eighteen points: one in flight while the other multiplies) feed
//              four MFMAs into acc[point] -- 36 x 4 accumulator registers (AGPRs) that live across the steps of an item, so the
//              output transform runs ONCE per item.
//   fold       Y = A^T M A from the 36 accumulators (constants folded at compile time) + bias into an LDS tile [256 pixels][64
//              channels] (over V, behind a barrier), stored as whole 256-byte pixel rows (dwordx4, optional accumulate).  The lines
//              of the NEXT step's patch are touched in front of it (into L2).
// BN: the input is a BatchNorm's input and relu(x * scale + shift) is convolved (LazyImg, as k_wino4_in<true>).
template <bool BN, bool VOUT>   // (VOUT a template parameter: its few registers cost the others five spills at the 256-register limit)
__global__ __launch_bounds__(256, 2) void k_wino4_conv(const float* __restrict__ in, int ld_in, const float* __restrict__ Uf,
                                                     const float* __restrict__ bias, float* __restrict__ out, int ld_out, int B, int H,
                                                     int W, int th, int tw, int Cin, int Cout, int accumulate,
                                                     const float* __restrict__ stats, int imgs_per_group, int nitems,
                                                     float* __restrict__ Vout
#ifdef W4C_PROFILE
                                                     , long long* prof
#endif
                                                     ) {
#ifdef W4C_PROFILE
  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define W4C_T(K_) { const long long n_ = __builtin_readcyclecounter(); pt[K_] += n_ - tprev; tprev = n_; }
#else
#define W4C_T(K_)
#endif
  extern __shared__ __attribute__((aligned(16))) float w4c_lds[];   // [2][36][64] float4; the output tile [256][64] floats over it
  f32x4w* __restrict__ lds = reinterpret_cast<f32x4w*>(w4c_lds);
  const int tid = threadIdx.x, lane = tid & 63, t = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ny = Cout >> 6;
  const int T = B * th * tw, tt = th * tw;
  const int nround = Cin >> 5, nchunk = Cin >> 4, nnb = Cout >> 4;
  const f32x4w zero4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t pstride = (int64_t)nchunk * nnb * 64;   // float4 elements per transform point
  f32x4w acc[36];
#pragma unroll
  for (int p = 0; p < 36; ++p) acc[p] = zero4;
  const f32x2w zero2 = {0.f, 0.f};
  f32x2w sc = {1.f, 1.f}, sh = zero2;
  unsigned rmask = 0, cmask = 0;
  float pf_sink = 0.f;
  // transform phase: lane = (tile tx_ = lane / 4, channel pair pr_ = lane % 4 of the wave's 8 channels) -- four consecutive lanes read
  // 32 contiguous bytes (with the MFMA numbering t + 16 q every lane of a load would start a segment of its own).  Channel
  // 8 wv + 2 pr_ + e of the step = chunk wv / 2, quad qq_ = 2 (wv % 2) + pr_ / 2, half hh_ = pr_ % 2 of the float4 in slot tx_ + 16 qq_.
  const int tx_ = lane >> 2, pr_ = lane & 3;
  const int qq_ = ((wv & 1) << 1) + (pr_ >> 1), hh_ = pr_ & 1;
  f32x4w bA[6], bB[6];
  // the patch of step (IT_, R_): tile coordinates, in-image masks, BatchNorm constants, 36 loads.  PF_: touch the lines only (into
  // L2, one step ahead: the 144 patch registers of a real prefetch do not fit beside the accumulators and the weight buffers) --
  // a dword load per lane into one scratch register that is kept alive until the next real patch has been waited for.
#define W4C_PATCH(IT_, R_, PF_)                                                                               \
  {                                                                                                           \
    const int tile_ = ((IT_) / ny) * 16 + tx_;                                                                \
    const bool tv_ = tile_ < T;                                                                               \
    const int tb_ = tv_ ? tile_ / tt : 0, trt_ = tv_ ? tile_ - tb_ * tt : 0;                                  \
    const int tty_ = trt_ / tw, ttx_ = trt_ - tty_ * tw;                                                      \
    const int y0_ = 4 * tty_ - 1, x0_ = 4 * ttx_ - 1;                                                         \
    const int ch0_ = ((R_) << 5) + (wv << 3) + 2 * pr_;                                                       \
    const float* __restrict__ pin_ = in + ((int64_t)(tb_ * H + y0_) * W + x0_) * ld_in + ch0_;               \
    const float* __restrict__ psafe_ = in + ch0_;                                                             \
    unsigned rm_ = 0, cm_ = 0;                                                                                \
    _Pragma("unroll") for (int a_ = 0; a_ < 6; ++a_) {                                                        \
      rm_ |= (tv_ && (unsigned)(y0_ + a_) < (unsigned)H) ? (1u << a_) : 0u;                                   \
      cm_ |= ((unsigned)(x0_ + a_) < (unsigned)W) ? (1u << a_) : 0u;                                          \
    }                                                                                                         \
    if (!(PF_)) {                                                                                             \
      rmask = rm_; cmask = cm_;                                                                               \
      if (BN) {                                                                                               \
        /* imgs_per_group = images per BatchNorm group | bn_c0 << 16 (one kernel argument more costs this kernel its register budget); */ \
        /* channels below bn_c0 pass through (see mopa_wino4_input_bn) */                                     \
        const int bc0_ = imgs_per_group >> 16;                                                                \
        sc = (f32x2w){1.f, 1.f}; sh = zero2;                                                                  \
        if (ch0_ >= bc0_) {                                                                                   \
          const float* __restrict__ sg_ = stats + (int64_t)(tb_ / (imgs_per_group & 0xffff)) * 4 * (Cin - bc0_) + ch0_ - bc0_; \
          sc = *reinterpret_cast<const f32x2w*>(sg_);                                                         \
          sh = *reinterpret_cast<const f32x2w*>(sg_ + Cin - bc0_);                                            \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
    _Pragma("unroll") for (int a_ = 0; a_ < 6; ++a_)                                                          \
      _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) {                                                      \
        const bool ins_ = ((rm_ >> a_) & (cm_ >> c_) & 1u) != 0u;                                             \
        const float* __restrict__ src_ = ins_ ? pin_ + ((int64_t)a_ * W + c_) * ld_in : psafe_;              \
        if (PF_) { if (wv == 0 && pr_ == 0) asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(src_)); } /* "+": ONE register; one lane per 128-byte line */ \
        else d[a_][c_] = *reinterpret_cast<const f32x2w*>(src_);                                              \
      }                                                                                                       \
  }
#define W4C_LOADB(BQ, C_, P0_)                                                                                \
  {                                                                                                           \
    const f32x4w* __restrict__ ub_ = Ufw + (int64_t)((r << 1) + (C_)) * nnb * 64 + (int64_t)(P0_) * pstride;  \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) BQ[i_] = ub_[i_ * pstride];                             \
  }
#define W4C_MUL6(BQ, C_, P0_, I0_)                                                                            \
  {                                                                                                           \
    const f32x4w* __restrict__ la_ = lds + ((C_) * 36 + (P0_) + (I0_)) * 64 + lane;                           \
    f32x4w av_[6];                                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) av_[i_] = la_[i_ * 64];                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                                        \
      acc[(P0_) + (I0_) + i_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[i_][0], BQ[(I0_) + i_][0], acc[(P0_) + (I0_) + i_], 0, 0, 0); \
      acc[(P0_) + (I0_) + i_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[i_][1], BQ[(I0_) + i_][1], acc[(P0_) + (I0_) + i_], 0, 0, 0); \
      acc[(P0_) + (I0_) + i_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[i_][2], BQ[(I0_) + i_][2], acc[(P0_) + (I0_) + i_], 0, 0, 0); \
      acc[(P0_) + (I0_) + i_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[i_][3], BQ[(I0_) + i_][3], acc[(P0_) + (I0_) + i_], 0, 0, 0); \
    }                                                                                                         \
  }
#define W4C_MUL(BQ, C_, P0_) W4C_MUL6(BQ, C_, P0_, 0)
  constexpr float AT[6][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f}, {1.f, 2.f, 4.f, 8.f}, {1.f, -2.f, 4.f, -8.f}, {0.f, 0.f, 0.f, 1.f}};
  int item = blockIdx.x, r = 0;
  while (item < nitems) {
    const int tg = item / ny, n0 = (item - tg * ny) << 6;
    const f32x4w* __restrict__ Ufw = reinterpret_cast<const f32x4w*>(Uf) + (int64_t)((n0 >> 4) + wv) * 64 + lane;
    W4C_T(0)
    // ---- transform: patch -> V chunk wv (everyone is done with the LDS of the previous step: barrier at its end)
    f32x2w d[6][6];
    W4C_PATCH(item, r, false);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const bool inside = ((rmask >> a) & (cmask >> c) & 1u) != 0u;
        f32x2w v = d[a][c];
        if (BN) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float o = fmaf(v[e], sc[e], sh[e]);
            v[e] = o > 0.f ? o : o * 0.f;
          }
        }
        d[a][c] = inside ? v : zero2;
      }
    // the touch loads of the previous step were issued before this patch's loads, which have just been consumed: in-order vmcnt says
    // they have landed; the explicit wait makes that hold whatever the compiler's wait-count pass (blind to VMEM issued from inline
    // asm) emitted above -- it is free here, nothing younger is in flight
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(pf_sink));
    W4C_T(1)
#pragma unroll
    for (int c = 0; c < 6; ++c) {   // B^T d, column by column (the formulas of w4_bt)
      const f32x2w d0 = d[0][c], d1 = d[1][c], d2 = d[2][c], d3 = d[3][c], d4 = d[4][c], d5 = d[5][c];
      w4_bt6(d0, d1, d2, d3, d4, d5, d[0][c], d[1][c], d[2][c], d[3][c], d[4][c], d[5][c]);
    }
    {
      f32x2w* __restrict__ lw = reinterpret_cast<f32x2w*>(w4c_lds) + (((wv >> 1) * 36) * 64 + tx_ + 16 * qq_) * 2 + hh_;
#pragma unroll
      for (int a = 0; a < 6; ++a) {   // (.) B
        f32x2w v0, v1, v2, v3, v4, v5;
        w4_bt6(d[a][0], d[a][1], d[a][2], d[a][3], d[a][4], d[a][5], v0, v1, v2, v3, v4, v5);
        lw[(a * 6 + 0) * 128] = v0;
        lw[(a * 6 + 1) * 128] = v1;
        lw[(a * 6 + 2) * 128] = v2;
        lw[(a * 6 + 3) * 128] = v3;
        lw[(a * 6 + 4) * 128] = v4;
        lw[(a * 6 + 5) * 128] = v5;
      }
    }
    W4C_LOADB(bA, 0, 0);
    W4C_LOADB(bB, 0, 6);
    const bool last_round = r + 1 == nround;
    const int nitem = last_round ? item + (int)gridDim.x : item, nr = last_round ? 0 : r + 1;
    W4C_T(2)
    __syncthreads();   // V is complete
    W4C_T(3)
    if (VOUT && n0 == 0) {
      // training forward: V[36][T][Cin] goes to HBM as a by-product (the weight gradient multiplies it again) -- written once, not read
      // by this layer.  A point is 16 tiles x 32 channels = two 1-KiB wave stores of whole 128-byte (tile, channel block) rows.
      const int c8 = lane & 7, vt = lane >> 3;   // float4 c8 of the step's 32 channels = chunk c8 / 4, quad c8 % 4
      // (running pointers, a rolled loop: this block must not cost the multiplication its registers)
      const f32x4w* __restrict__ lp = lds + ((c8 >> 2) * 36 + wv) * 64 + vt + 16 * (c8 & 3);
      float* __restrict__ vp = Vout + ((int64_t)wv * T + (tg << 4) + vt) * Cin + (r << 5) + 4 * c8;
      const int64_t vstep = (int64_t)4 * T * Cin;
      const bool ok0 = (tg << 4) + vt < T, ok1 = (tg << 4) + vt + 8 < T;
#pragma unroll 1
      for (int k = 0; k < 9; ++k, lp += 4 * 64, vp += vstep) {
        const f32x4w v0 = lp[0], v1 = lp[8];
        if (ok0) *reinterpret_cast<f32x4w*>(vp) = v0;
        if (ok1) *reinterpret_cast<f32x4w*>(vp + 8 * Cin) = v1;
      }
    }
#pragma unroll 1
    for (int c = 0; c < 2; ++c) {   // (a loop, not 12 units of straight-line code: the scheduler would hoist every weight load)
      W4C_MUL(bA, c, 0);   W4C_LOADB(bA, c, 12);
      W4C_MUL(bB, c, 6);   W4C_LOADB(bB, c, 18);
      W4C_MUL(bA, c, 12);  W4C_LOADB(bA, c, 24);
      W4C_MUL(bB, c, 18);  W4C_LOADB(bB, c, 30);
      W4C_MUL(bA, c, 24);  if (c < 1) W4C_LOADB(bA, c + 1, 0);
      W4C_MUL(bB, c, 30);  if (c < 1) W4C_LOADB(bB, c + 1, 6);
    }
    W4C_T(4)
    __syncthreads();   // everyone is done reading V
    W4C_T(5)
    // touch the next step's patch: it comes from L2 instead of HBM when the loop comes round
    if (nitem < nitems) W4C_PATCH(nitem, nr, true);
    if (last_round) {
      // ---- output transform: lane holds M[point][tile 4 q + i][channel n0 + 16 wv + t], i = 0..3
      float o[4][16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
#pragma unroll
      for (int pi = 0; pi < 6; ++pi) {
        float srow[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) srow[i][c] = 0.f;
#pragma unroll
        for (int pj = 0; pj < 6; ++pj)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (AT[pj][c] != 0.f) srow[i][c] = fmaf(AT[pj][c], acc[pi * 6 + pj][i], srow[i][c]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (AT[pi][a] != 0.f) o[i][a * 4 + c] = fmaf(AT[pi][a], srow[i][c], o[i][a * 4 + c]);
      }
#pragma unroll
      for (int p = 0; p < 36; ++p) acc[p] = zero4;
      const float bvv = bias ? bias[n0 + (wv << 4) + t] : 0.f;
      // [tile 16][pixel 16][channel 64] with 4 floats between tiles (the four lane groups q write tiles 4 apart: 2-way instead of 4-way
      // bank conflicts)
      float* __restrict__ ot = w4c_lds + (wv << 4) + t;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[(4 * q + i) * 1028 + e * 64] = o[i][e] + bvv;
      __syncthreads();
      // whole pixel rows: thread = (pixel wv * 4 + q of the tile, channel quad t); one tile per iteration, its coordinates are uniform and
      // stepped, not divided
      {
        int otile = tg << 4;
        int ob = otile / tt, ort = otile - ob * tt;
        int oty = ort / tw, otx = ort - oty * tw;
        const f32x4w* __restrict__ sp = reinterpret_cast<const f32x4w*>(w4c_lds + (wv * 4 + q) * 64 + 4 * t);
#pragma unroll 4
        for (int k = 0; k < 16 && otile < T; ++k, ++otile) {
          const int y = 4 * oty + wv, x = 4 * otx + q;
          if (y < H && x < W) {
            f32x4w v = sp[k * 257];   // (1028 floats per tile)
            f32x4w* gp = reinterpret_cast<f32x4w*>(out + ((int64_t)(ob * H + y) * W + x) * ld_out + n0 + 4 * t);
            if (accumulate) v += *gp;
            *gp = v;
          }
          if (++otx == tw) { otx = 0; if (++oty == th) { oty = 0; ++ob; } }
        }
      }
      __syncthreads();   // the tile has been read: the next step may write V over it
    }
    W4C_T(6)
    item = nitem; r = nr;
  }
#undef W4C_PATCH
#undef W4C_LOADB
#undef W4C_MUL
#undef W4C_MUL6
#ifdef W4C_PROFILE
  if (blockIdx.x == 100 && tid == 0)
    for (int k = 0; k < 8; ++k) prof[k] = pt[k];
#endif
}

// k_wino4_conv with 32 tiles per item (round-4 review, item 2): 8 waves = (16 output channels wv) x (point half ph).  A wave multiplies
// BOTH 16-tile halves by its 18 points' weight fragments -- 8 MFMAs per fragment instead of 4, half the weight bytes from L2 per tile
// -- and holds 2 x 18 accumulator blocks = the 144 registers of k_wino4_conv, so two waves share a SIMD as before (one wave per SIMD
// with all 36 points would need 288 accumulators: the compiler spills 157 registers of it).  V of both halves: 144 KB of LDS.
template <bool BN, bool VOUT>
__global__ __launch_bounds__(512, 2) void k_wino4_conv32(const float* __restrict__ in, int ld_in, const float* __restrict__ Uf,
                                                     const float* __restrict__ bias, float* __restrict__ out, int ld_out, int B, int H,
                                                     int W, int th, int tw, int Cin, int Cout, int accumulate,
                                                     const float* __restrict__ stats, int imgs_per_group, int nitems,
                                                     float* __restrict__ Vout
#ifdef W4C_PROFILE
                                                     , long long* prof
#endif
                                                     ) {
#ifdef W4C_PROFILE
  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define W4C_T(K_) { const long long n_ = __builtin_readcyclecounter(); pt[K_] += n_ - tprev; tprev = n_; }
#else
#define W4C_T(K_)
#endif
  extern __shared__ __attribute__((aligned(16))) float w4c_lds[];   // [2 tile halves][2 chunks][36][64] float4 = 144 KB; two output tiles [16][1028] floats over it
  f32x4w* __restrict__ lds = reinterpret_cast<f32x4w*>(w4c_lds);
  const int tid = threadIdx.x, lane = tid & 63, t = lane & 15, q = lane >> 4;
  const int wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);   // 8 waves
  const int wv = wv8 & 3;     // transform: the wave's 8 channels of the step; multiply: its 16 output channels
  const int ph = wv8 >> 2;    // transform / by-product: tile half; multiply / fold: point half (transform rows pi = 3 ph .. 3 ph + 2)
  const int ny = Cout >> 6;
  const int T = B * th * tw, tt = th * tw;
  const int nround = Cin >> 5, nchunk = Cin >> 4, nnb = Cout >> 4;
  const f32x4w zero4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t pstride = (int64_t)nchunk * nnb * 64;   // float4 elements per transform point
  f32x4w acc[2][18];   // [tile half][point of the wave's point half]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int p = 0; p < 18; ++p) acc[h][p] = zero4;
  const f32x2w zero2 = {0.f, 0.f};
  f32x2w sc = {1.f, 1.f}, sh = zero2;
  unsigned rmask = 0, cmask = 0;
  float pf_sink = 0.f;
  // transform phase: lane = (tile tx_ = lane / 4, channel pair pr_ = lane % 4 of the wave's 8 channels) -- four consecutive lanes read
  // 32 contiguous bytes (with the MFMA numbering t + 16 q every lane of a load would start a segment of its own).  Channel
  // 8 wv + 2 pr_ + e of the step = chunk wv / 2, quad qq_ = 2 (wv % 2) + pr_ / 2, half hh_ = pr_ % 2 of the float4 in slot tx_ + 16 qq_.
  const int tx_ = lane >> 2, pr_ = lane & 3;
  const int qq_ = ((wv & 1) << 1) + (pr_ >> 1), hh_ = pr_ & 1;
  f32x4w bA[6], bB[6];
  // the patch of step (IT_, R_): tile coordinates, in-image masks, BatchNorm constants, 36 loads.  PF_: touch the lines only (into
  // L2, one step ahead: the 144 patch registers of a real prefetch do not fit beside the accumulators and the weight buffers) --
  // a dword load per lane into one scratch register that is kept alive until the next real patch has been waited for.
#define W4C_PATCH(IT_, R_, PF_)   /* the wave's tile half: ph */                                                                               \
  {                                                                                                           \
    const int tile_ = ((IT_) / ny) * 32 + ph * 16 + tx_;                                                      \
    const bool tv_ = tile_ < T;                                                                               \
    const int tb_ = tv_ ? tile_ / tt : 0, trt_ = tv_ ? tile_ - tb_ * tt : 0;                                  \
    const int tty_ = trt_ / tw, ttx_ = trt_ - tty_ * tw;                                                      \
    const int y0_ = 4 * tty_ - 1, x0_ = 4 * ttx_ - 1;                                                         \
    const int ch0_ = ((R_) << 5) + (wv << 3) + 2 * pr_;                                                       \
    const float* __restrict__ pin_ = in + ((int64_t)(tb_ * H + y0_) * W + x0_) * ld_in + ch0_;               \
    const float* __restrict__ psafe_ = in + ch0_;                                                             \
    unsigned rm_ = 0, cm_ = 0;                                                                                \
    _Pragma("unroll") for (int a_ = 0; a_ < 6; ++a_) {                                                        \
      rm_ |= (tv_ && (unsigned)(y0_ + a_) < (unsigned)H) ? (1u << a_) : 0u;                                   \
      cm_ |= ((unsigned)(x0_ + a_) < (unsigned)W) ? (1u << a_) : 0u;                                          \
    }                                                                                                         \
    if (!(PF_)) {                                                                                             \
      rmask = rm_; cmask = cm_;                                                                               \
      if (BN) {                                                                                               \
        /* imgs_per_group = images per BatchNorm group | bn_c0 << 16 (one kernel argument more costs this kernel its register budget); */ \
        /* channels below bn_c0 pass through (see mopa_wino4_input_bn) */                                     \
        const int bc0_ = imgs_per_group >> 16;                                                                \
        sc = (f32x2w){1.f, 1.f}; sh = zero2;                                                                  \
        if (ch0_ >= bc0_) {                                                                                   \
          const float* __restrict__ sg_ = stats + (int64_t)(tb_ / (imgs_per_group & 0xffff)) * 4 * (Cin - bc0_) + ch0_ - bc0_; \
          sc = *reinterpret_cast<const f32x2w*>(sg_);                                                         \
          sh = *reinterpret_cast<const f32x2w*>(sg_ + Cin - bc0_);                                            \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
    _Pragma("unroll") for (int a_ = 0; a_ < 6; ++a_)                                                          \
      _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) {                                                      \
        const bool ins_ = ((rm_ >> a_) & (cm_ >> c_) & 1u) != 0u;                                             \
        const float* __restrict__ src_ = ins_ ? pin_ + ((int64_t)a_ * W + c_) * ld_in : psafe_;              \
        if (PF_) { if (wv == 0 && pr_ == 0) asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(src_)); } /* "+": ONE register; one lane per 128-byte line */ \
        else d[a_][c_] = *reinterpret_cast<const f32x2w*>(src_);                                              \
      }                                                                                                       \
  }
#define W4C_LOADB(BQ, C_, P0_)                                                                                \
  {                                                                                                           \
    const f32x4w* __restrict__ ub_ = Ufw + (int64_t)((r << 1) + (C_)) * nnb * 64 + (int64_t)(ph * 18 + (P0_)) * pstride;  \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) BQ[i_] = ub_[i_ * pstride];                             \
  }
#define W4C_MUL6(BQ, C_, P0_, I0_)                                                                            \
  _Pragma("unroll") for (int hf_ = 0; hf_ < 2; ++hf_) {   /* the same weight fragments for both tile halves: 8 MFMAs per fragment */ \
    const f32x4w* __restrict__ la_ = lds + ((hf_ * 2 + (C_)) * 36 + ph * 18 + (P0_) + (I0_)) * 64 + lane;     \
    f32x4w av_[6];                                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) av_[i_] = la_[i_ * 64];                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                                        \
      acc[hf_][(P0_) + (I0_) + i_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[i_][0], BQ[(I0_) + i_][0], acc[hf_][(P0_) + (I0_) + i_], 0, 0, 0); \
      acc[hf_][(P0_) + (I0_) + i_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[i_][1], BQ[(I0_) + i_][1], acc[hf_][(P0_) + (I0_) + i_], 0, 0, 0); \
      acc[hf_][(P0_) + (I0_) + i_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[i_][2], BQ[(I0_) + i_][2], acc[hf_][(P0_) + (I0_) + i_], 0, 0, 0); \
      acc[hf_][(P0_) + (I0_) + i_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[i_][3], BQ[(I0_) + i_][3], acc[hf_][(P0_) + (I0_) + i_], 0, 0, 0); \
    }                                                                                                         \
  }
#define W4C_MUL(BQ, C_, P0_) W4C_MUL6(BQ, C_, P0_, 0)
  constexpr float AT[6][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f}, {1.f, 2.f, 4.f, 8.f}, {1.f, -2.f, 4.f, -8.f}, {0.f, 0.f, 0.f, 1.f}};
  int item = blockIdx.x, r = 0;
  while (item < nitems) {
    const int tg = item / ny, n0 = (item - tg * ny) << 6;
    const f32x4w* __restrict__ Ufw = reinterpret_cast<const f32x4w*>(Uf) + (int64_t)((n0 >> 4) + wv) * 64 + lane;
    W4C_T(0)
    // ---- transform: patch -> V chunk wv (everyone is done with the LDS of the previous step: barrier at its end)
    f32x2w d[6][6];
    W4C_PATCH(item, r, false);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const bool inside = ((rmask >> a) & (cmask >> c) & 1u) != 0u;
        f32x2w v = d[a][c];
        if (BN) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float o = fmaf(v[e], sc[e], sh[e]);
            v[e] = o > 0.f ? o : o * 0.f;
          }
        }
        d[a][c] = inside ? v : zero2;
      }
    // the touch loads of the previous step were issued before this patch's loads, which have just been consumed: in-order vmcnt says
    // they have landed; the explicit wait makes that hold whatever the compiler's wait-count pass (blind to VMEM issued from inline
    // asm) emitted above -- it is free here, nothing younger is in flight
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(pf_sink));
    W4C_T(1)
#pragma unroll
    for (int c = 0; c < 6; ++c) {   // B^T d, column by column (the formulas of w4_bt)
      const f32x2w d0 = d[0][c], d1 = d[1][c], d2 = d[2][c], d3 = d[3][c], d4 = d[4][c], d5 = d[5][c];
      w4_bt6(d0, d1, d2, d3, d4, d5, d[0][c], d[1][c], d[2][c], d[3][c], d[4][c], d[5][c]);
    }
    {
      f32x2w* __restrict__ lw = reinterpret_cast<f32x2w*>(w4c_lds) + (((ph * 2 + (wv >> 1)) * 36) * 64 + tx_ + 16 * qq_) * 2 + hh_;
#pragma unroll
      for (int a = 0; a < 6; ++a) {   // (.) B
        f32x2w v0, v1, v2, v3, v4, v5;
        w4_bt6(d[a][0], d[a][1], d[a][2], d[a][3], d[a][4], d[a][5], v0, v1, v2, v3, v4, v5);
        lw[(a * 6 + 0) * 128] = v0;
        lw[(a * 6 + 1) * 128] = v1;
        lw[(a * 6 + 2) * 128] = v2;
        lw[(a * 6 + 3) * 128] = v3;
        lw[(a * 6 + 4) * 128] = v4;
        lw[(a * 6 + 5) * 128] = v5;
      }
    }
    W4C_LOADB(bA, 0, 0);
    W4C_LOADB(bB, 0, 6);
    const bool last_round = r + 1 == nround;
    const int nitem = last_round ? item + (int)gridDim.x : item, nr = last_round ? 0 : r + 1;
    W4C_T(2)
    __syncthreads();   // V is complete
    W4C_T(3)
    if (VOUT && n0 == 0) {
      // training forward: V[36][T][Cin] goes to HBM as a by-product (the weight gradient multiplies it again) -- written once, not read
      // by this layer.  A point is 16 tiles x 32 channels = two 1-KiB wave stores of whole 128-byte (tile, channel block) rows.
      const int c8 = lane & 7, vt = lane >> 3;   // float4 c8 of the step's 32 channels = chunk c8 / 4, quad c8 % 4
      // (running pointers, a rolled loop: this block must not cost the multiplication its registers)
      const f32x4w* __restrict__ lp = lds + ((ph * 2 + (c8 >> 2)) * 36 + wv) * 64 + vt + 16 * (c8 & 3);
      float* __restrict__ vp = Vout + ((int64_t)wv * T + (tg << 5) + (ph << 4) + vt) * Cin + (r << 5) + 4 * c8;
      const int64_t vstep = (int64_t)4 * T * Cin;
      const bool ok0 = (tg << 5) + (ph << 4) + vt < T, ok1 = (tg << 5) + (ph << 4) + vt + 8 < T;
#pragma unroll 1
      for (int k = 0; k < 9; ++k, lp += 4 * 64, vp += vstep) {
        const f32x4w v0 = lp[0], v1 = lp[8];
        if (ok0) *reinterpret_cast<f32x4w*>(vp) = v0;
        if (ok1) *reinterpret_cast<f32x4w*>(vp + 8 * Cin) = v1;
      }
    }
    // six units of (6 points x 2 tile halves x 4 MFMAs): the weight fragments of the unit after next are loaded behind each
    W4C_MUL(bA, 0, 0);   W4C_LOADB(bA, 0, 12);
    W4C_MUL(bB, 0, 6);   W4C_LOADB(bB, 1, 0);
    W4C_MUL(bA, 0, 12);  W4C_LOADB(bA, 1, 6);
    W4C_MUL(bB, 1, 0);   W4C_LOADB(bB, 1, 12);
    W4C_MUL(bA, 1, 6);
    W4C_MUL(bB, 1, 12);
    W4C_T(4)
    __syncthreads();   // everyone is done reading V
    W4C_T(5)
    // touch the next step's patch: it comes from L2 instead of HBM when the loop comes round
    if (nitem < nitems) W4C_PATCH(nitem, nr, true);
    if (last_round) {
      // ---- output transform: lane holds M[point][tile 4 q + i][channel n0 + 16 wv + t], i = 0..3, for the transform rows pi = 3 ph ..
      // 3 ph + 2 of both tile halves.  Y = A^T M A is linear in the rows: each wave folds its three rows in registers and the two
      // partial sums meet in the LDS tiles -- wave (wv, ph) writes tile half ph and keeps its share of the other half in the (dead)
      // accumulators; after a barrier it adds that share to tile half 1 - ph, which wave (wv, 1 - ph) wrote.
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float o[4][16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          float srow[4][4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) srow[i][c] = 0.f;
#pragma unroll
          for (int pj = 0; pj < 6; ++pj)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (AT[pj][c] != 0.f) srow[i][c] = fmaf(AT[pj][c], acc[hf][pl * 6 + pj][i], srow[i][c]);
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const float ca = ph ? AT[3 + pl][a] : AT[pl][a];   // (wave-uniform)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int c = 0; c < 4; ++c) o[i][a * 4 + c] = fmaf(ca, srow[i][c], o[i][a * 4 + c]);
          }
        }
#pragma unroll
        for (int p = 0; p < 18; ++p) acc[hf][p] = zero4;
        if (hf == ph) {
          // [tile half][tile 16][pixel 16][channel 64] with 4 floats between tiles (the four lane groups q write tiles 4 apart)
          float* __restrict__ ot = w4c_lds + hf * (16 * 1028) + (wv << 4) + t;
          const float bvv = bias ? bias[n0 + (wv << 4) + t] : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) ot[(4 * q + i) * 1028 + e * 64] = o[i][e] + bvv;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[hf][i * 4 + (e >> 2)][e & 3] = o[i][e];
        }
      }
      __syncthreads();   // each tile half holds one point half (+ bias)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
        if (hf != ph) {
          float* __restrict__ ot = w4c_lds + hf * (16 * 1028) + (wv << 4) + t;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) ot[(4 * q + i) * 1028 + e * 64] += acc[hf][i * 4 + (e >> 2)][e & 3];
#pragma unroll
          for (int p = 0; p < 16; ++p) acc[hf][p] = zero4;
        }
      __syncthreads();
      // whole pixel rows: thread = (pixel wv * 4 + q of the tile, channel quad t) of tile half ph; one tile per iteration, its coordinates
      // are uniform and stepped, not divided
      {
        int otile = (tg << 5) + (ph << 4);
        int ob = otile / tt, ort = otile - ob * tt;
        int oty = ort / tw, otx = ort - oty * tw;
        const f32x4w* __restrict__ sp = reinterpret_cast<const f32x4w*>(w4c_lds + ph * (16 * 1028) + (wv * 4 + q) * 64 + 4 * t);
#pragma unroll 4
        for (int k = 0; k < 16 && otile < T; ++k, ++otile) {
          const int y = 4 * oty + wv, x = 4 * otx + q;
          if (y < H && x < W) {
            f32x4w v = sp[k * 257];   // (1028 floats per tile)
            f32x4w* gp = reinterpret_cast<f32x4w*>(out + ((int64_t)(ob * H + y) * W + x) * ld_out + n0 + 4 * t);
            if (accumulate) v += *gp;
            *gp = v;
          }
          if (++otx == tw) { otx = 0; if (++oty == th) { oty = 0; ++ob; } }
        }
      }
      __syncthreads();   // the tiles have been read: the next step may write V over them
    }
    W4C_T(6)
    item = nitem; r = nr;
  }
#undef W4C_PATCH
#undef W4C_LOADB
#undef W4C_MUL
#undef W4C_MUL6
#ifdef W4C_PROFILE
  if (blockIdx.x == 100 && tid == 0)
    for (int k = 0; k < 8; ++k) prof[k] = pt[k];
#endif
}

// B-operand fragments of k_wino4_conv from the OIHW weight (weight_forms.h: transpose = 2); dgrad as in mopa_wino4_weight.
MOPA_API int mopa_wino4_weight_f(const float* weight, int32_t O, int32_t I, int32_t dgrad, float* Uf, void* stream) {
  if (O <= 0 || I <= 0 || O % 16 || I % 16) return MOPA_ERR_ARG;
  k_wino4_w<<<stream_grid((int64_t)O * I, 256), 256, 0, (hipStream_t)stream>>>(weight, O, I, dgrad, Uf, 2);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// B-operand fragments of k_wino4_conv9 (wino4c9.hip) from the OIHW weight (weight_forms.h: transpose = 3); dgrad as in mopa_wino4_weight:
// Uq[p][ci / 16][co / 32][(ci % 16) / 8][lane = 32 (ci % 2) + co % 32][(ci % 8) / 2].  (Here, not in wino4c9.hip: the same kernel and
// translation unit as the other forms, so that the batched refresh rebuilds the same bits.)
MOPA_API int mopa_wino4_weight_q(const float* weight, int32_t O, int32_t I, int32_t dgrad, float* Uq, void* stream) {
  const int R = dgrad ? O : I, C = dgrad ? I : O;
  if (O <= 0 || I <= 0 || R % 16 || C % 32) return MOPA_ERR_ARG;
  k_wino4_w<<<stream_grid((int64_t)O * I, 256), 256, 0, (hipStream_t)stream>>>(weight, O, I, dgrad, Uq, 3);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// Which kernel mopa_wino4_conv launches: -1 = by shape (the default), 0 = k_wino4_conv (16 tiles per item), 1 = k_wino4_conv32 (32 tiles);
// the environment variable MOPA_WINO4_CONV32 gives the initial value (tests and profiles/ set it).
static std::atomic<int>& w4c_tiles32_mode() {
  static std::atomic<int> mode([] { const char* e = getenv("MOPA_WINO4_CONV32"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }());
  return mode;
}
MOPA_API int mopa_wino4_conv_tiles32(int32_t mode) {
  w4c_tiles32_mode().store(mode < 0 ? -1 : mode != 0 ? 1 : 0, std::memory_order_relaxed);
  return MOPA_OK;
}

// out (NHWC, row stride ld_out) = conv3x3(in) (+ bias) (+= if accumulate) through F(4x4,3x3) in ONE kernel: mopa_wino4_input +
// mopa_wino4_gemm_output without V.  Uf: mopa_wino4_weight_f.  Cin % 64 == 0, Cout % 64 == 0; `in` / `out` 16-byte aligned, ld_in % 4 ==
// ld_out % 4 == 0.
// stats != null: `in` is a BatchNorm's input, stats = [n_groups][4][Cin] (mopa_bn_act_fwd_groups with y == null), the B images are
// n_groups equal consecutive groups and relu(batchnorm(in)) is what is convolved (as mopa_wino4_input_bn, bn_c0 likewise).
// V != null: the transformed input [36][T][Cin] (what mopa_wino4_input would write: the same bits) is stored as a by-product -- the
// forward pass of a training step keeps it for mopa_wino4_bwd_weight.
MOPA_API int mopa_wino4_conv(const float* in, int32_t ld_in, const float* Uf, const float* bias, float* out, int32_t ld_out, int32_t B,
                             int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t accumulate, const float* stats, int32_t n_groups,
                             int32_t bn_c0, float* V, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64 || ld_in < Cin || (ld_in & 3) || ld_out < Cout ||
      (ld_out & 3) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || ((uintptr_t)V & 15))
    return MOPA_ERR_ARG;
  if (stats && (n_groups < 1 || B % n_groups || B / n_groups > 0xffff || bn_c0 < 0 || bn_c0 >= Cin || (bn_c0 & 1))) return MOPA_ERR_ARG;
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  const int64_t T = (int64_t)B * th * tw;
  const int ncu = mopa_cu_count();   // of the current device
  const int dev = mopa_device_index();
  if (ncu <= 0 || dev < 0) return MOPA_ERR_LAUNCH;
  // k_wino4_conv32 (32 tiles per item, one 8-wave workgroup per CU: half the weight bytes per tile, but the CU's transform and
  // multiply phases no longer overlap) instead of k_wino4_conv (16 tiles, two 4-wave workgroups per CU) where it measured faster:
  // from 8 items per CU on (profiles/r5_wino4_conv32.md; 16 x 304 x 480: 64->128 1487 -> 1393 us, 128->64 1477 -> 1320 us).
  // mopa_wino4_conv_tiles32 / MOPA_WINO4_CONV32 = 0 / 1: never / always.
  const int conv32_mode = w4c_tiles32_mode().load(std::memory_order_relaxed);
  const bool conv32 = conv32_mode >= 0 ? conv32_mode != 0 : cdiv64(T, 32) * (Cout / 64) >= 8 * (int64_t)ncu;
  const int64_t nitems = cdiv64(T, conv32 ? 32 : 16) * (Cout / 64);
  if (T >= (1 << 30) || nitems >= (1ll << 31)) return MOPA_ERR_ARG;
  const size_t ldsb = (size_t)(conv32 ? 4 : 2) * 36 * 64 * 16;   // 72 KB: two workgroups per CU; 144 KB: one
  const int v = (stats ? 1 : 0) | (V ? 2 : 0), va = v | (conv32 ? 4 : 0);
  typedef void (*kern_t)(const float*, int, const float*, const float*, float*, int, int, int, int, int, int, int, int, int, const float*, int,
                         int, float*
#ifdef W4C_PROFILE
                         , long long*
#endif
                         );
  static const kern_t kerns16[4] = {k_wino4_conv<false, false>, k_wino4_conv<true, false>, k_wino4_conv<false, true>, k_wino4_conv<true, true>};
  static const kern_t kerns32[4] = {k_wino4_conv32<false, false>, k_wino4_conv32<true, false>, k_wino4_conv32<false, true>,
                                    k_wino4_conv32<true, true>};
  const kern_t* kerns = conv32 ? kerns32 : kerns16;
  const int per_cu = conv32 ? 1 : 2, nthr = conv32 ? 512 : 256;
  static std::atomic<bool> attr[64][8];   // per device and variant: the attribute belongs to the device's code object
  if (!attr[dev][va].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[v]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess)
      return MOPA_ERR_LAUNCH;
    attr[dev][va].store(true, std::memory_order_release);
  }
  const unsigned nblk = (unsigned)(nitems < per_cu * ncu ? nitems : per_cu * ncu);   // persistent
  const int ipg = stats ? (B / n_groups) | (bn_c0 << 16) : 1;
#ifdef W4C_PROFILE
  static long long* prof = nullptr;
  if (!prof) hipMallocManaged(&prof, 128);
  kerns[v]<<<nblk, nthr, ldsb, (hipStream_t)stream>>>(in, ld_in, Uf, bias, out, ld_out, B, H, W, th, tw, Cin, Cout, accumulate, stats, ipg,
                                                  (int)nitems, V, prof);
  hipStreamSynchronize((hipStream_t)stream);
  {
    const double n_ = (double)nitems / nblk * (Cin / 32);
    printf("[w4c profile] block 100 wave 0, cycles per step: loop head %.0f | patch wait + apply %.0f | transform, lds write, B + patch issue %.0f | "
           "barrier %.0f | multiply %.0f | barrier %.0f | fold + store %.0f   (steps per block %.1f)\n",
           prof[0] / n_, prof[1] / n_, prof[2] / n_, prof[3] / n_, prof[4] / n_, prof[5] / n_, prof[6] / n_, n_);
  }
#else
  kerns[v]<<<nblk, nthr, ldsb, (hipStream_t)stream>>>(in, ld_in, Uf, bias, out, ld_out, B, H, W, th, tw, Cin, Cout, accumulate, stats, ipg,
                                                  (int)nitems, V);
#endif
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
