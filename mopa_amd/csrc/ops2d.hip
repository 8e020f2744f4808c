// Elementwise / small-reduction ops of the image branch on NHWC fp32: MaxPool 3x3 s2 p1, Dropout, the full-image
// linear head, column sums (conv bias gradients).  All HBM-bound: float4 access, channels on the fast axis.
//
// Reference call sites: mopa/models/resnet34_unet.py:148 (maxpool), :154,:159 (nn.Dropout p=0.4), decoder conv
// biases :104-110; mopa/models/xmuda_arch.py:58-60 (full-image head).  Oracle: oracle/net2d.py.
#include "common.h"

#define PH_MAXNC 16  // classes handled by the full-image head kernels (reference configs: 5 and 10)

// ------------------------------------------------------------------------------------------ maxpool 3x3 s2 p1
// y[b][oy][ox][c] = max over the 3x3 window (first maximum in (ky,kx) scan order wins, like torch); idx stores the
// winning tap 0..8 (uint8) for the backward.  x may be a channel slice (ldx) of a wider buffer.
// BN: x is a BatchNorm's input and the pooled tensor is relu(x * scale + shift) (stats[G][4][C] of mopa_bn_act_fwd_groups with y ==
// null; the expression of k_bn_relu_apply, so values and winners are those of pooling the materialised tensor).
template <bool BN>
__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* __restrict__ x, int ldx, int B, int H, int W, int C,
                                                      float* __restrict__ y, int ldy, unsigned char* __restrict__ idx,
                                                      const float* __restrict__ stats, int imgs_per_group) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2, CQ = C >> 2;
  const int64_t total = (int64_t)B * OH * OW * CQ;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % CQ);
    int64_t r = i / CQ;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH), b = (int)(r / OH);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (BN) {
      const float* __restrict__ sg = stats + (int64_t)(b / imgs_per_group) * 4 * C + cq * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) { sc[j] = sg[j]; sh[j] = sg[C + j]; }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + ((int64_t)(b * H + iy) * W + ix) * ldx + cq * 4);
        float vs[4] = {v.x, v.y, v.z, v.w};
        if (BN) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float o = fmaf(vs[j], sc[j], sh[j]);
            vs[j] = o > 0.f ? o : o * 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (vs[j] > best[j] || vs[j] != vs[j]) { best[j] = vs[j]; bi[j] = ky * 3 + kx; }
      }
    const int64_t o = ((int64_t)(b * OH + oy) * OW + ox);
    *reinterpret_cast<float4*>(y + o * ldy + cq * 4) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<uchar4*>(idx + o * C + cq * 4) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
  }
}
// Gather-form backward (no atomics).  One thread per 2 x 2 block of input pixels and channel quad: the block (rows 2a, 2a+1, columns
// 2b, 2b+1) lies under exactly the four windows (a, b), (a, b+1), (a+1, b), (a+1, b+1) -- an even row or column under one of them,
// an odd one under two -- so four (argmax, dy) pairs serve four outputs (one thread per input pixel loaded nine pairs for the same
// four: 2.2 TB/s).  The sums are taken in the same window order as before (oy then ox ascending): same bits.
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ dy, int ld_dy, const unsigned char* __restrict__ idx,
                                                      int B, int H, int W, int C, float* __restrict__ dx, int ld_dx, int accumulate) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2, CQ = C >> 2;
  const int HB = (H + 1) / 2, WB = (W + 1) / 2;   // 2 x 2 blocks
  const int64_t total = (int64_t)B * HB * WB * CQ;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % CQ);
    int64_t r = i / CQ;
    const int bx = (int)(r % WB); r /= WB;
    const int by = (int)(r % HB), b = (int)(r / HB);
    float4 g[2][2];
    uchar4 w[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int oy = by + a, ox = bx + c;
        if (oy < OH && ox < OW) {
          const int64_t o = ((int64_t)(b * OH + oy) * OW + ox);
          w[a][c] = *reinterpret_cast<const uchar4*>(idx + o * C + cq * 4);
          g[a][c] = *reinterpret_cast<const float4*>(dy + o * ld_dy + cq * 4);
        } else {
          w[a][c] = make_uchar4(255, 255, 255, 255);
          g[a][c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int iy = 2 * by + py, ix = 2 * bx + px;
        if (iy >= H || ix >= W) continue;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a <= py; ++a)        // window rows containing iy: by (tap row 1 + py) and, for the odd row, by + 1 (tap row 0)
#pragma unroll
          for (int c = 0; c <= px; ++c) {
            const int tap = (a == 0 ? 1 + py : 0) * 3 + (c == 0 ? 1 + px : 0);
            if (w[a][c].x == tap) s[0] += g[a][c].x;
            if (w[a][c].y == tap) s[1] += g[a][c].y;
            if (w[a][c].z == tap) s[2] += g[a][c].z;
            if (w[a][c].w == tap) s[3] += g[a][c].w;
          }
        float4* p = reinterpret_cast<float4*>(dx + ((int64_t)(b * H + iy) * W + ix) * ld_dx + cq * 4);
        if (accumulate) { const float4 q = *p; s[0] += q.x; s[1] += q.y; s[2] += q.z; s[3] += q.w; }
        *p = make_float4(s[0], s[1], s[2], s[3]);
      }
  }
}
MOPA_API int mopa_maxpool3x3s2_fwd(const float* x, int32_t ldx, int32_t B, int32_t H, int32_t W, int32_t C, float* y,
                                   int32_t ldy, uint8_t* argmax, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ldx < C || ldy < C || ((ldx | ldy) & 3)) return MOPA_ERR_ARG;
  const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C >> 2);
  k_maxpool_fwd<false><<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(x, ldx, B, H, W, C, y, ldy, argmax, nullptr, 1);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
// The same pooling of relu(batchnorm(x)) read from the BatchNorm's input x: stats = [n_groups][4][C] (scale, shift, ...) of
// mopa_bn_act_fwd_groups with y == null, the B images are n_groups equal consecutive groups.  Values and argmax are those of
// mopa_maxpool3x3s2_fwd on the applied tensor (the stem's BatchNorm never writes its output: dense2d, MOPA_DEFER_STEM_BN).
MOPA_API int mopa_maxpool3x3s2_fwd_bn(const float* x, int32_t ldx, int32_t B, int32_t H, int32_t W, int32_t C, const float* stats,
                                      int32_t n_groups, float* y, int32_t ldy, uint8_t* argmax, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ldx < C || ldy < C || ((ldx | ldy) & 3) || !stats || n_groups < 1 || B % n_groups)
    return MOPA_ERR_ARG;
  const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C >> 2);
  k_maxpool_fwd<true><<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(x, ldx, B, H, W, C, y, ldy, argmax, stats, B / n_groups);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_maxpool3x3s2_bwd(const float* dy, int32_t ld_dy, const uint8_t* argmax, int32_t B, int32_t H, int32_t W,
                                   int32_t C, float* dx, int32_t ld_dx, int32_t accumulate, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ld_dx < C || ld_dy < C || ((ld_dx | ld_dy) & 3)) return MOPA_ERR_ARG;
  k_maxpool_bwd<<<stream_grid((int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C >> 2), 256), 256, 0, (hipStream_t)stream>>>(
      dy, ld_dy, argmax, B, H, W, C, dx, ld_dx, accumulate);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ dropout
// Counter-based keep mask: element i of call `seed` is kept iff hash(seed, i) < (1-p)*2^32; kept values are scaled by
// 1/(1-p).  The same function regenerates the mask in the backward (nothing is stored).  Not torch's RNG stream.
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t i) {
  uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 16);
}
__global__ void k_dropout_rows(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int64_t rows, int C,
                               uint64_t seed, uint32_t keep_thresh, float scale, int identity,
                               const int64_t* __restrict__ seed_dev, int site) {
  const int CQ = C >> 2;
  const int64_t total = rows * CQ;
  if (seed_dev) seed = (uint64_t)seed_dev[0] * 2 + (uint64_t)site;   // the seed lives in device memory (HIP-graph replays)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / CQ;
    const int cq = (int)(i - row * CQ);
    float4 v = *reinterpret_cast<const float4*>(x + row * ldx + cq * 4);
    if (!identity) {
      const uint64_t e = (uint64_t)(row * C + cq * 4);
      v.x = hash_u32(seed, e) < keep_thresh ? v.x * scale : 0.f;
      v.y = hash_u32(seed, e + 1) < keep_thresh ? v.y * scale : 0.f;
      v.z = hash_u32(seed, e + 2) < keep_thresh ? v.z * scale : 0.f;
      v.w = hash_u32(seed, e + 3) < keep_thresh ? v.w * scale : 0.f;
    }
    *reinterpret_cast<float4*>(y + row * ldy + cq * 4) = v;
  }
}
// y = dropout(x) over a [rows, C] slice (also the backward: dx = dropout(dy) with the same seed).  p in [0,1);
// p == 0 is an exact copy.
MOPA_API int mopa_dropout_rows(const float* x, int32_t ldx, float* y, int32_t ldy, int64_t rows, int32_t C, float p,
                               uint64_t seed, void* stream) {
  if (rows <= 0 || C <= 0 || (C & 3) || ldx < C || ldy < C || ((ldx | ldy) & 3) || p < 0.f || p >= 1.f) return MOPA_ERR_ARG;
  const double keep = 1.0 - (double)p;
  const uint32_t thresh = keep >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(keep * 4294967296.0);
  k_dropout_rows<<<stream_grid(rows * (C >> 2), 256), 256, 0, (hipStream_t)stream>>>(x, ldx, y, ldy, rows, C, seed, thresh,
                                                                                     (float)(1.0 / keep), p == 0.f, nullptr, 0);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// The same op with the call's seed read from device memory at run time: mask of (seed_dev[0] * 2 + site), i.e. what
// mopa_dropout_rows(..., seed = seed_dev[0] * 2 + site) draws.  For launches recorded once into a HIP graph and replayed with a
// new seed per iteration (mopa_amd/dense2d.py, graph replay of the 2D backbone).
MOPA_API int mopa_dropout_rows_dseed(const float* x, int32_t ldx, float* y, int32_t ldy, int64_t rows, int32_t C, float p,
                                     const int64_t* seed_dev, int32_t site, void* stream) {
  if (rows <= 0 || C <= 0 || (C & 3) || ldx < C || ldy < C || ((ldx | ldy) & 3) || p < 0.f || p >= 1.f || !seed_dev || site < 0 || site > 1)
    return MOPA_ERR_ARG;
  const double keep = 1.0 - (double)p;
  const uint32_t thresh = keep >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(keep * 4294967296.0);
  k_dropout_rows<<<stream_grid(rows * (C >> 2), 256), 256, 0, (hipStream_t)stream>>>(x, ldx, y, ldy, rows, C, 0, thresh,
                                                                                     (float)(1.0 / keep), p == 0.f, seed_dev, site);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ full-image linear head
// pred[b][h][w][k] = x[b][h][w][:] . W[k][:] + bias[k] for h < H, w < W of the padded (Hp, Wp) feature map
// (xmuda_arch.py:58-60 on the cropped map, resnet34_unet.py:185-186).  M / 4 lanes per pixel (16 for the 64-channel map): a
// pixel's row is ONE coalesced 256-byte read across its lanes, each lane keeps its float4 of every W[k] in registers and a
// 4-step shuffle tree sums the partial dot products (round 1: one thread per pixel walked the 256-B row by itself and the
// kernel moved 5.5x its algorithmic bytes -- profiles/r1_joint_hbm_traffic.json).
// Sum over the MQ (power of two) consecutive lanes of a pixel, result in all of them.  Up to 16 lanes = one DPP row: four VALU
// instructions with a data-parallel-primitive operand (quad swaps, half-row mirror, row mirror) instead of four ds_bpermute round
// trips through the LDS crossbar per class -- the round-1..3 kernel issued 20 of those per pixel and ran at 1.8 TB/s.
template <int CTRL> __device__ __forceinline__ float ph_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float ph_lane_sum(float v, int MQ) {
  if (MQ >= 2) v = ph_dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  if (MQ >= 4) v = ph_dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  if (MQ >= 8) v = ph_dpp_add<0x141>(v);   // row_half_mirror
  if (MQ >= 16) v = ph_dpp_add<0x140>(v);  // row_mirror
  for (int o = 16; o < MQ; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void k_pixel_head_fwd(const float* __restrict__ x, int ld, int B, int Hp, int Wp, int H, int W,
                                                         int M, int NC, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ pred) {
  const int MQ = M >> 2;                       // lanes per pixel: 4, 8 or 16 (power of two, checked by the launcher)
  const int cq = threadIdx.x % MQ;
  float4 wr[PH_MAXNC];
  float bk[PH_MAXNC];
#pragma unroll
  for (int k = 0; k < PH_MAXNC; ++k) {
    wr[k] = k < NC ? *reinterpret_cast<const float4*>(w + k * M + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    bk[k] = k < NC ? bias[k] : 0.f;
  }
  const int PPB = 256 / MQ;                    // pixels per block and iteration
  const int64_t total = (int64_t)B * H * W;
  // PH_UN pixels per lane group and iteration: their row reads are issued back to back (one 16-byte load in flight per lane left
  // the kernel waiting on HBM latency: 16 KB per CU in flight)
  constexpr int PH_UN = 4;
  const int64_t step = (int64_t)gridDim.x * PPB;
  for (int64_t i0 = (int64_t)blockIdx.x * PPB + threadIdx.x / MQ; i0 < total; i0 += step * PH_UN) {
    float4 v[PH_UN];
#pragma unroll
    for (int u = 0; u < PH_UN; ++u) {
      const int64_t i = i0 + u * step < total ? i0 + u * step : total - 1;   // (clamped: an in-bounds read, result not stored)
      const int wq = (int)(i % W);
      const int64_t r = i / W;
      const int h = (int)(r % H), b = (int)(r / H);
      v[u] = *reinterpret_cast<const float4*>(x + ((int64_t)(b * Hp + h) * Wp + wq) * ld + cq * 4);
    }
#pragma unroll
    for (int u = 0; u < PH_UN; ++u) {
      const int64_t i = i0 + u * step;
      float out = 0.f;
#pragma unroll
      for (int k = 0; k < PH_MAXNC; ++k) {
        if (k < NC) {   // uniform
          const float a = ph_lane_sum(fmaf(v[u].x, wr[k].x, fmaf(v[u].y, wr[k].y, fmaf(v[u].z, wr[k].z, v[u].w * wr[k].w))), MQ);
          out = cq == k ? a + bk[k] : out;
        }
      }
      if (cq < NC && i < total) pred[i * NC + cq] = out;
    }
  }
}
// dx[pix][c] (+)= sum_k dpred[pix][k] W[k][c] inside the H x W window (rows outside are left untouched).
__global__ __launch_bounds__(256) void k_pixel_head_bwd_x(const float* __restrict__ dpred, int B, int Hp, int Wp, int H, int W, int M,
                                                           int NC, const float* __restrict__ w, float* __restrict__ dx, int ld,
                                                           int accumulate) {
  extern __shared__ float lw[];
  for (int i = threadIdx.x; i < NC * M; i += 256) lw[i] = w[i];
  __syncthreads();
  const int MQ = M >> 2;
  const int64_t total = (int64_t)B * H * W * MQ;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % MQ);
    int64_t r = i / MQ;
    const int64_t pix = r;
    const int wq = (int)(r % W); r /= W;
    const int h = (int)(r % H), b = (int)(r / H);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < NC; ++k) {
      const float g = dpred[pix * NC + k];
      const float* wr = lw + k * M + cq * 4;
      s[0] = fmaf(g, wr[0], s[0]); s[1] = fmaf(g, wr[1], s[1]); s[2] = fmaf(g, wr[2], s[2]); s[3] = fmaf(g, wr[3], s[3]);
    }
    float4* p = reinterpret_cast<float4*>(dx + ((int64_t)(b * Hp + h) * Wp + wq) * ld + cq * 4);
    if (accumulate) { const float4 q = *p; s[0] += q.x; s[1] += q.y; s[2] += q.z; s[3] += q.w; }
    *p = make_float4(s[0], s[1], s[2], s[3]);
  }
}
// dW[k][c] = sum_pix dpred[pix][k] x[pix][c], db[k] = sum_pix dpred[pix][k]; block partials [nblk][NC][M+1].
#define PH_PIX_PER_BLOCK 1024
__global__ __launch_bounds__(256) void k_pixel_head_wgrad_partial(const float* __restrict__ dpred, const float* __restrict__ x, int ld,
                                                                   int B, int Hp, int Wp, int H, int W, int M, int NC,
                                                                   float* __restrict__ partial) {
  // thread = (pixel lane pl = t / 16, channel quad cq = t % 16) for M = 64; generic: MQ = M/4 quads, PL = 256/MQ lanes
  extern __shared__ float red[];  // [PL][NC][M+1]
  const int MQ = M >> 2, PL = 256 / MQ;
  const int cq = threadIdx.x % MQ, pl = threadIdx.x / MQ;
  const int64_t total = (int64_t)B * H * W;
  const int64_t p0 = (int64_t)blockIdx.x * PH_PIX_PER_BLOCK, p1 = min(total, p0 + PH_PIX_PER_BLOCK);
  float acc[PH_MAXNC][4];
  float accb[PH_MAXNC];
#pragma unroll
  for (int k = 0; k < PH_MAXNC; ++k) { acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f; accb[k] = 0.f; }
  if (pl < PL)
    for (int64_t pix = p0 + pl; pix < p1; pix += PL) {
      const int wq = (int)(pix % W);
      const int64_t r = pix / W;
      const int h = (int)(r % H), b = (int)(r / H);
      const float4 v = *reinterpret_cast<const float4*>(x + ((int64_t)(b * Hp + h) * Wp + wq) * ld + cq * 4);
#pragma unroll
      for (int k = 0; k < PH_MAXNC; ++k)
        if (k < NC) {
          const float g = dpred[pix * NC + k];
          acc[k][0] = fmaf(g, v.x, acc[k][0]); acc[k][1] = fmaf(g, v.y, acc[k][1]);
          acc[k][2] = fmaf(g, v.z, acc[k][2]); acc[k][3] = fmaf(g, v.w, acc[k][3]);
          if (cq == 0) accb[k] += g;
        }
    }
  const int stride = NC * (M + 1);
  if (pl < PL)
#pragma unroll
    for (int k = 0; k < PH_MAXNC; ++k)
      if (k < NC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl * stride + k * (M + 1) + cq * 4 + j] = acc[k][j];
        if (cq == 0) red[pl * stride + k * (M + 1) + M] = accb[k];
      }
  __syncthreads();
  for (int i = threadIdx.x; i < stride; i += 256) {
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * stride + i];
    partial[(int64_t)blockIdx.x * stride + i] = s;
  }
}
__global__ __launch_bounds__(256) void k_head_partial_reduce(const float* __restrict__ partial, int nblk, int M, int NC,
                                                              float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  const int nout = NC * (M + 1);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wv; i < nout; i += gridDim.x * 4) {
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)partial[(int64_t)b * nout + i];
    s = wave_sum_d(s);
    if (lane == 0) {
      const int k = i / (M + 1), c = i - k * (M + 1);
      float* dst = (c < M) ? &dw[k * M + c] : &db[k];
      *dst = (accumulate ? *dst : 0.f) + (float)s;
    }
  }
}

MOPA_API int mopa_pixel_head_fwd(const float* x, int32_t ld, int32_t B, int32_t Hp, int32_t Wp, int32_t H, int32_t W, int32_t M,
                                 int32_t num_classes, const float* w, const float* bias, float* pred, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || H > Hp || W > Wp || M <= 0 || (M & 3) || num_classes <= 0 || num_classes > PH_MAXNC || ld < M || (ld & 3))
    return MOPA_ERR_ARG;
  const int MQ = M >> 2;   // lanes per pixel: a power of two <= 64 that holds every class (lane k writes class k)
  if ((MQ & (MQ - 1)) != 0 || MQ > 64 || MQ < num_classes || (((uintptr_t)x | (uintptr_t)w) & 15)) return MOPA_ERR_ARG;
  k_pixel_head_fwd<<<stream_grid(cdiv64((int64_t)B * H * W, 256 / MQ) * 256, 256), 256, 0, (hipStream_t)stream>>>(
      x, ld, B, Hp, Wp, H, W, M, num_classes, w, bias, pred);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API size_t mopa_pixel_head_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t M, int32_t num_classes) {
  return align_up((size_t)cdiv64((int64_t)B * H * W, PH_PIX_PER_BLOCK) * num_classes * (M + 1) * sizeof(float), 256);
}
MOPA_API int mopa_pixel_head_bwd(const float* dpred, const float* x, int32_t ld, int32_t B, int32_t Hp, int32_t Wp, int32_t H,
                                 int32_t W, int32_t M, int32_t num_classes, const float* w, float* dx, int32_t ld_dx,
                                 int32_t accumulate_dx, float* dw, float* db, int32_t accumulate_params, void* ws, size_t ws_bytes,
                                 void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || H > Hp || W > Wp || M <= 0 || (M & 3) || 256 % (M >> 2) != 0 || num_classes <= 0 ||
      num_classes > PH_MAXNC || ld < M || (ld & 3) || ld_dx < M || (ld_dx & 3))
    return MOPA_ERR_ARG;
  if (ws_bytes < mopa_pixel_head_bwd_workspace_bytes(B, H, W, M, num_classes)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  k_pixel_head_bwd_x<<<stream_grid((int64_t)B * H * W * (M >> 2), 256), 256, (size_t)num_classes * M * sizeof(float), st>>>(
      dpred, B, Hp, Wp, H, W, M, num_classes, w, dx, ld_dx, accumulate_dx);
  const int nblk = (int)cdiv64((int64_t)B * H * W, PH_PIX_PER_BLOCK);
  const int PL = 256 / (M >> 2);
  k_pixel_head_wgrad_partial<<<nblk, 256, (size_t)PL * num_classes * (M + 1) * sizeof(float), st>>>(dpred, x, ld, B, Hp, Wp, H, W, M,
                                                                                                    num_classes, (float*)ws);
  k_head_partial_reduce<<<32, 256, 0, st>>>((const float*)ws, nblk, M, num_classes, dw, db, accumulate_params);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ column sums (bias grads)
// out[c] (+)= sum_rows x[row][c]
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ x, int ld, int A, int C, int rows_per_block,
                                                         float* __restrict__ partial) {
  extern __shared__ float lds[];  // [RL][C]
  const int CQ = C >> 2, RL = 256 / CQ;
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const int rbeg = blockIdx.x * rows_per_block, rend = min(A, rbeg + rows_per_block);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    for (int row = rbeg + rl; row < rend; row += RL) {
      const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)row * ld + cq * 4);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
    for (int j = 0; j < 4; ++j) lds[rl * C + cq * 4 + j] = s[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int k = 0; k < RL; ++k) t += lds[k * C + c];
    partial[(int64_t)blockIdx.x * C + c] = t;
  }
}
// one block of 256 per column (like k_bn_finalize): <= 8 independent partial loads per thread, wave sums in double, the four wave
// sums added in order.  (16 blocks of one wave per column walked the partials 32 deep: 28 us of latency per call.)
__global__ __launch_bounds__(256) void k_colsum_reduce(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out,
                                                        int accumulate) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double s = 0.0;
#pragma unroll 8
  for (int b = threadIdx.x; b < nblk; b += 256) s += (double)partial[(int64_t)b * C + c];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (accumulate ? out[c] : 0.f) + (float)(((red[0] + red[1]) + red[2]) + red[3]);
}
static inline int colsum_rows(int64_t num_rows) {
  int64_t r = cdiv64(num_rows, 2048);
  if (r < 32) r = 32;
  if (r > 1024) r = 1024;
  return (int)r;
}
MOPA_API size_t mopa_colsum_workspace_bytes(int64_t num_rows, int32_t C) {
  return align_up((size_t)cdiv64(num_rows, colsum_rows(num_rows)) * C * sizeof(float), 256);
}
MOPA_API int mopa_colsum(const float* x, int32_t ld, int64_t num_rows, int32_t C, float* out, int32_t accumulate, void* ws,
                         size_t ws_bytes, void* stream) {
  if (num_rows <= 0 || C <= 0 || (C & 3) || C > 1024 || ld < C || (ld & 3)) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_colsum_workspace_bytes(num_rows, C)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int rpb = colsum_rows(num_rows);
  const int nblk = (int)cdiv64(num_rows, rpb);
  const int RL = 256 / (C >> 2);
  k_colsum_partial<<<nblk, 256, (size_t)RL * C * sizeof(float), st>>>(x, ld, (int)num_rows, C, rpb, (float*)ws);
  k_colsum_reduce<<<C, 256, 0, st>>>((const float*)ws, nblk, C, out, accumulate);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
