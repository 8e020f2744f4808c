// Dense 2D convolution of the image branch as an LDS-tiled implicit GEMM.  Two inner products over the same staging and
// index map: f32-operand MFMA (default; exact fp32, measured 95-119 TFLOP/s) and the fp32 vector pipe (v_pk_fma_f32,
// 60-94 TFLOP/s; north_star's literal choice, kept selectable with MOPA_CONV2D_MFMA=0 -- see DESIGN.md section 3 for the
// measurements behind the default).  NHWC activations, one kernel for
//   * forward conv (3x3 s1/s2, 1x1 s2, the 7x7 stem through a 16-wide tap trick),
//   * backward-data (stride 1 directly; stride 2 as 4 output-parity classes so no FLOP is wasted on zeros),
//   * ConvTranspose2d k2 s2 (4 output-parity classes of a 1x1 conv) and its backward-data (a 2x2 s2 conv),
// selected by an index map, plus a split-K backward-weight kernel.
//
// Replaces the cuDNN calls behind mopa/models/resnet34_unet.py:93-110,144-182 (Conv2d / ConvTranspose2d of
// UNetResNet34).  Oracle: oracle/net2d.py (torch-CPU conv2d / conv_transpose2d), golden fixture G1.
//
// GEMM view:  out[m][n] = sum_{tap, c} A[m][tap][c] * Wt[tap][c][n]
//   m = (b, oy, ox) over a LOGICAL output grid [B][OHl][OWl]; the element lands at (oy*OS+OOY, ox*OS+OOX) of the
//   actual output image [OHa][OWa];  tap = (ty, tx) over [TH][TW] logical taps; its input pixel is
//   (oy*IS + IY0 + ty*IDY, ox*IS + IX0 + tx*IDX), zero outside [IH][IW]; its weight slice is
//   w[(KH0 + ty*KS) * KWF + (KW0 + tx*KS)] of shape [Cin][Cout].
#include "common.h"
#include "wino4.h"
#include "weight_forms.h"
#include <stdlib.h>
#include <string.h>

struct ConvGeom {
  int B, IH, IW, OHl, OWl, OHa, OWa;
  int OS, OOY, OOX;
  int IS, IY0, IX0, IDY, IDX;
  int TH, TW, KH0, KW0, KS, KWF;
  int Cin, Cout, ld_in, ld_out;
};

#define BK 16
#define APAD 4

// BM x BN output tile per 256-thread block, 8x8 register micro-tile per thread (TX = BN/8 threads across n).
// LDS: A is kept k-major ([k][m], transposed while staging) so that each k-step is 2+2 ds_read_b128 feeding
// 64 FMAs with only 16 operand registers live; double-buffered, one barrier per 16-deep K-chunk, the next chunk's
// global loads are issued before the FMAs of the current one.
template <int BM, int BN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void k_conv2d_igemm(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          const ConvGeom g, int accumulate, int64_t bs_in, int64_t bs_w,
                                                          int64_t bs_out) {
  // blockIdx.z: independent problems of one shape (the 16 Winograd points); strides in floats, 0 for a single conv
  in += blockIdx.z * bs_in;
  w += blockIdx.z * bs_w;
  out += blockIdx.z * bs_out;
  constexpr int TX = BN / TN, TY = 256 / TX;
  static_assert(TY * TM == BM && (TM == 4 || TM == 8) && (TN == 4 || TN == 8), "tile shape");
  constexpr int AROWS = BM / 64;             // A float4 loads per thread per K-chunk
  constexpr int BVEC = (BK * BN / 4) / 256;  // B float4 loads per thread per K-chunk
  constexpr int BMP = BM + APAD;
  __shared__ __attribute__((aligned(16))) float As[2][BK][BMP];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
  const int t = threadIdx.x, tx = t % TX, ty = t / TX;
  const int M = g.B * g.OHl * g.OWl;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int ohw = g.OHl * g.OWl;

  // staging assignment: A row = t/4 + 64*j, k-quad = t%4
  const int akq = t & 3, arow = t >> 2;
  int ab[AROWS], aoy[AROWS], aox[AROWS];
#pragma unroll
  for (int j = 0; j < AROWS; ++j) {
    const int m = m0 + arow + 64 * j;
    if (m < M) {
      const int b = m / ohw, r = m - b * ohw;
      ab[j] = b; aoy[j] = r / g.OWl; aox[j] = r - aoy[j] * g.OWl;
    } else {
      ab[j] = -1; aoy[j] = 0; aox[j] = 0;
    }
  }
  const int cchunks = g.Cin / BK;
  const int niter = g.TH * g.TW * cchunks;
  float4 ra[AROWS], rb[BVEC];

#define LOAD_TILE(IT)                                                                                              \
  {                                                                                                                \
    const int tap_ = (IT) / cchunks, c0_ = ((IT) - tap_ * cchunks) * BK;                                           \
    const int tyy_ = tap_ / g.TW, txx_ = tap_ - tyy_ * g.TW;                                                       \
    _Pragma("unroll") for (int j = 0; j < AROWS; ++j) {                                                            \
      const int iy = aoy[j] * g.IS + g.IY0 + tyy_ * g.IDY, ix = aox[j] * g.IS + g.IX0 + txx_ * g.IDX;              \
      if (ab[j] >= 0 && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW)                            \
        ra[j] = *reinterpret_cast<const float4*>(in + ((int64_t)(ab[j] * g.IH + iy) * g.IW + ix) * g.ld_in + c0_ + akq * 4); \
      else                                                                                                         \
        ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                   \
    }                                                                                                              \
    const float* wt_ = w + ((int64_t)((g.KH0 + tyy_ * g.KS) * g.KWF + g.KW0 + txx_ * g.KS) * g.Cin + c0_) * g.Cout + n0; \
    _Pragma("unroll") for (int j = 0; j < BVEC; ++j) {                                                             \
      const int idx = t + 256 * j, k = idx / (BN / 4), c4 = idx - k * (BN / 4);                                    \
      rb[j] = *reinterpret_cast<const float4*>(wt_ + (int64_t)k * g.Cout + c4 * 4);                                \
    }                                                                                                              \
  }
#define STORE_TILE(BUF)                                                                                            \
  {                                                                                                                \
    _Pragma("unroll") for (int j = 0; j < AROWS; ++j) {                                                            \
      As[BUF][akq * 4 + 0][arow + 64 * j] = ra[j].x;                                                               \
      As[BUF][akq * 4 + 1][arow + 64 * j] = ra[j].y;                                                               \
      As[BUF][akq * 4 + 2][arow + 64 * j] = ra[j].z;                                                               \
      As[BUF][akq * 4 + 3][arow + 64 * j] = ra[j].w;                                                               \
    }                                                                                                              \
    _Pragma("unroll") for (int j = 0; j < BVEC; ++j) {                                                             \
      const int idx = t + 256 * j, k = idx / (BN / 4), c4 = idx - k * (BN / 4);                                    \
      *reinterpret_cast<float4*>(&Bs[BUF][k][c4 * 4]) = rb[j];                                                     \
    }                                                                                                              \
  }

  float acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = 0.f;

  LOAD_TILE(0);
  STORE_TILE(0);
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int cur = it & 1;
    if (it + 1 < niter) LOAD_TILE(it + 1);  // global loads in flight while this tile is consumed
#pragma unroll 4
    for (int k = 0; k < BK; ++k) {
      float av[TM], bv[TN];
      {
        const float4 a0 = *reinterpret_cast<const float4*>(&As[cur][k][ty * TM]);
        av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w;
        if (TM == 8) {
          const float4 a1 = *reinterpret_cast<const float4*>(&As[cur][k][ty * TM + 4]);
          av[TM - 4] = a1.x; av[TM - 3] = a1.y; av[TM - 2] = a1.z; av[TM - 1] = a1.w;
        }
        const float4 b0 = *reinterpret_cast<const float4*>(&Bs[cur][k][tx * 4]);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
        if (TN == 8) {
          const float4 b1 = *reinterpret_cast<const float4*>(&Bs[cur][k][BN / 2 + tx * 4]);
          bv[TN - 4] = b1.x; bv[TN - 3] = b1.y; bv[TN - 2] = b1.z; bv[TN - 1] = b1.w;
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    if (it + 1 < niter) {
      STORE_TILE(cur ^ 1);
      __syncthreads();
    }
  }
#undef LOAD_TILE
#undef STORE_TILE
  // epilogue
  float4 bv0 = make_float4(0.f, 0.f, 0.f, 0.f), bv1 = bv0;
  if (bias) {
    bv0 = *reinterpret_cast<const float4*>(bias + n0 + tx * 4);
    if (TN == 8) bv1 = *reinterpret_cast<const float4*>(bias + n0 + BN / 2 + tx * 4);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + ty * TM + i;
    if (m >= M) continue;
    const int b = m / ohw, r = m - b * ohw;
    const int oy = r / g.OWl, ox = r - oy * g.OWl;
    float* o = out + ((int64_t)(b * g.OHa + oy * g.OS + g.OOY) * g.OWa + ox * g.OS + g.OOX) * g.ld_out + n0;
    float4 v0 = make_float4(acc[i][0] + bv0.x, acc[i][1] + bv0.y, acc[i][2] + bv0.z, acc[i][3] + bv0.w);
    float4* p0 = reinterpret_cast<float4*>(o + tx * 4);
    if (accumulate) {
      const float4 q0 = *p0;
      v0.x += q0.x; v0.y += q0.y; v0.z += q0.z; v0.w += q0.w;
    }
    *p0 = v0;
    if (TN == 8) {
      float4 v1 = make_float4(acc[i][TN - 4] + bv1.x, acc[i][TN - 3] + bv1.y, acc[i][TN - 2] + bv1.z, acc[i][TN - 1] + bv1.w);
      float4* p1 = reinterpret_cast<float4*>(o + BN / 2 + tx * 4);
      if (accumulate) {
        const float4 q1 = *p1;
        v1.x += q1.x; v1.y += q1.y; v1.z += q1.z; v1.w += q1.w;
      }
      *p1 = v1;
    }
  }
}

// ----------------------------------------------------------------------------------------------
// The same implicit GEMM on the matrix cores with f32 operands (v_mfma_f32_32x32x2_f32: exact fp32 products and fp32
// accumulation -- the arithmetic of the fmaf chain above, in a different summation order -- at the fp32 VECTOR peak rate,
// 157 TFLOP/s).  Why: the 8x8 register-tile kernel needs 64 B of LDS operands per 64 FMAs and ~1.3 instructions per 2 FMAs;
// it tops out at 60 % of that peak (94 TFLOP/s on its best shape, 40-50 % on the others).  An MFMA takes its 64+64 operand
// floats from ONE ds_read_b32 per lane each and keeps the SIMD busy for 64 cycles, so staging and addressing hide behind it.
// Same staging (k-major A tile, double-buffered LDS, next chunk's global loads in flight), same tile shapes and the same
// index map; the four waves of a block own WM x WN sub-tiles of 32x32 MFMA tiles.  MOPA_CONV2D_MFMA=0 selects the
// vector-FMA kernel above (A/B measurements in DESIGN.md).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int WM, int WN, int KB>
__global__ __launch_bounds__(256, 2) void k_conv2d_igemm_mfma(const float* __restrict__ in, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               const ConvGeom g, int accumulate, int64_t bs_in, int64_t bs_w,
                                                               int64_t bs_out) {
  in += blockIdx.z * bs_in;
  w += blockIdx.z * bs_w;
  out += blockIdx.z * bs_out;
  constexpr int MT = WM / 32, NT = WN / 32;
  static_assert((BM / WM) * (BN / WN) == 4 && WM % 32 == 0 && WN % 32 == 0, "four waves per block");
  constexpr int KQ = KB / 4;                  // k-quads per row: the 256 threads stage RPP = 256 / KQ rows per pass
  constexpr int RPP = 256 / KQ;
  constexpr int AROWS = BM / RPP;             // A float4 loads per thread per K-chunk
  constexpr int BVEC = (KB * BN / 4) / 256;   // B float4 loads per thread per K-chunk
  constexpr int BMP = BM + APAD;
  __shared__ __attribute__((aligned(16))) float As[2][KB][BMP];
  __shared__ __attribute__((aligned(16))) float Bs[2][KB][BN];
  __shared__ int64_t rowoff[BM];             // output offset of each tile row (-1: beyond M)
  const int t = threadIdx.x;
  const int M = g.B * g.OHl * g.OWl;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int ohw = g.OHl * g.OWl;
  // staging assignment: A row = t/KQ + RPP*j, k-quad = t%KQ.  Per row: input coordinates and element offset at tap (0,0);
  // a tap adds a uniform delta, a K-chunk a uniform channel offset -> the loads of the loop are `uniform base + 32-bit
  // per-lane offset` (saddr form, no 64-bit VALU), unconditional (rows outside the image read element 0 and are zeroed
  // when staged), and the per-lane offsets change once per tap only.
  const int akq = t % KQ, arow = t / KQ;
  int ay0[AROWS], ax0[AROWS], abase[AROWS];
#pragma unroll
  for (int j = 0; j < AROWS; ++j) {
    const int m = m0 + arow + RPP * j;
    if (m < M) {
      const int b = m / ohw, r = m - b * ohw;
      const int oy = r / g.OWl, ox = r - oy * g.OWl;
      ay0[j] = oy * g.IS + g.IY0;
      ax0[j] = ox * g.IS + g.IX0;
      abase[j] = ((b * g.IH + ay0[j]) * g.IW + ax0[j]) * g.ld_in + akq * 4;
    } else {
      ay0[j] = -(1 << 28); ax0[j] = 0; abase[j] = 0;   // never inside the image
    }
  }
  uint32_t boff[BVEC];
#pragma unroll
  for (int j = 0; j < BVEC; ++j) {
    const int idx = t + 256 * j, k = idx / (BN / 4), c4 = idx - k * (BN / 4);
    boff[j] = (uint32_t)(k * g.Cout + c4 * 4) * 4u;
  }
  for (int r = t; r < BM; r += 256) {
    const int m = m0 + r;
    int64_t off = -1;
    if (m < M) {
      const int b = m / ohw, q = m - b * ohw;
      const int oy = q / g.OWl, ox = q - oy * g.OWl;
      off = ((int64_t)(b * g.OHa + oy * g.OS + g.OOY) * g.OWa + ox * g.OS + g.OOX) * g.ld_out;
    }
    rowoff[r] = off;
  }
  const int niter = g.TH * g.TW * (g.Cin / KB);
  float4 ra0[AROWS], rb0[BVEC];
  bool ok0[AROWS];
  uint32_t aoff[AROWS];
  bool aok[AROWS];
  int tyy = 0, txx = 0, c0 = 0;
  const float* wtap;
  auto set_tap = [&]() {
    const int dy = tyy * g.IDY, dx = txx * g.IDX;
    const int delta = (dy * g.IW + dx) * g.ld_in;
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
      aok[j] = (unsigned)(ay0[j] + dy) < (unsigned)g.IH && (unsigned)(ax0[j] + dx) < (unsigned)g.IW;
      aoff[j] = aok[j] ? (uint32_t)(abase[j] + delta) * 4u : 0u;
    }
    wtap = w + (int64_t)((g.KH0 + tyy * g.KS) * g.KWF + g.KW0 + txx * g.KS) * g.Cin * g.Cout + n0;
  };
  auto load_tile = [&](float4* ra, float4* rb, bool* ok) {
    const char* ic = reinterpret_cast<const char*>(in + c0);
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
      ra[j] = *reinterpret_cast<const float4*>(ic + aoff[j]);
      ok[j] = aok[j];
    }
    const char* wc = reinterpret_cast<const char*>(wtap + (int64_t)c0 * g.Cout);
#pragma unroll
    for (int j = 0; j < BVEC; ++j) rb[j] = *reinterpret_cast<const float4*>(wc + boff[j]);
  };
  auto advance = [&]() {   // uniform: next K-chunk, next tap after the last chunk of this one
    c0 += KB;
    if (c0 == g.Cin) {
      c0 = 0;
      if (++txx == g.TW) { txx = 0; ++tyy; }
      set_tap();
    }
  };
  auto store_tile = [&](int buf, const float4* ra, const float4* rb, const bool* ok) {
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
      As[buf][akq * 4 + 0][arow + RPP * j] = ok[j] ? ra[j].x : 0.f;
      As[buf][akq * 4 + 1][arow + RPP * j] = ok[j] ? ra[j].y : 0.f;
      As[buf][akq * 4 + 2][arow + RPP * j] = ok[j] ? ra[j].z : 0.f;
      As[buf][akq * 4 + 3][arow + RPP * j] = ok[j] ? ra[j].w : 0.f;
    }
#pragma unroll
    for (int j = 0; j < BVEC; ++j) {
      const int idx = t + 256 * j, k = idx / (BN / 4), c4 = idx - k * (BN / 4);
      *reinterpret_cast<float4*>(&Bs[buf][k][c4 * 4]) = rb[j];
    }
  };

  const int lane = t & 63, wv = t >> 6;
  const int wm0 = (wv % (BM / WM)) * WM, wn0 = (wv / (BM / WM)) * WN;
  const int l32 = lane & 31, lk = lane >> 5;   // MFMA operand: row / column l32, k index lk
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto multiply = [&](int cur) {
#pragma unroll
    for (int kk = 0; kk < KB; kk += 2) {
      float av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = As[cur][kk + lk][wm0 + i * 32 + l32];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = Bs[cur][kk + lk][wn0 + j * 32 + l32];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };

  set_tap();
  load_tile(ra0, rb0, ok0);
  store_tile(0, ra0, rb0, ok0);
  __syncthreads();
  // PMC (profiles/pmc_igemm.sh): the MFMA pipe is busy 73 % of the time at 5 waves per SIMD; the waves wait on the loaded
  // L2 / Infinity-Cache latency of the next chunk (with the loads removed the loop reaches 83 %).  A second register stage
  // (loads two chunks ahead) was measured twice: +8 % on small-M direct shapes with 64x64 tiles, -4 % on the batched Winograd
  // GEMMs that now carry those layers, +-0 on the 128x64 long-K convs (joint step 222.1 vs 222.4 scans/s): not kept.
#pragma unroll 1
  for (int it = 0; it < niter; ++it) {
    const int cur = it & 1;
    if (it + 1 < niter) {   // global loads in flight while this tile is consumed
      advance();
      load_tile(ra0, rb0, ok0);
    }
    multiply(cur);
    if (it + 1 < niter) {
      store_tile(cur ^ 1, ra0, rb0, ok0);
      __syncthreads();
    }
  }
  // epilogue: accumulator register e of a 32x32 tile = row 8*(e/4) + 4*lk + e%4, column l32 -> every store instruction
  // writes two 128-byte runs (32 consecutive output channels of two pixels)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn0 + j * 32 + l32;
    const float bvj = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t off = rowoff[wm0 + i * 32 + 8 * (e >> 2) + 4 * lk + (e & 3)];
        if (off < 0) continue;
        float v = acc[i][j][e] + bvj;
        if (accumulate) v += out[off + n];
        out[off + n] = v;
      }
    }
  }
}

// Tile choice, from measurements on MI355X at the network's own shapes (profiles/bench_igemm.py, B=8):
//   M >= 200k pixels: 256x64 (8x8 register tile) 68-82 TF/s; with Cout % 128 == 0 and M >= 1M: 128x128 84 TF/s;
//   M <  200k pixels (layer2-4, decoder stages 3-5): 64x64 (4x4 register tile) fills the 256 CUs -- 50-69 TF/s
//   where the large tiles reach 34-60 (4,560 pixels x 512 channels is 144 blocks of 256x64 on 256 CUs).
struct TileCfg { int bm, bn; };
static const TileCfg kTiles[4] = {{256, 64}, {128, 128}, {128, 64}, {64, 64}};
static int pick_tile(int64_t M, int cout, bool mfma) {
  // MFMA kernel (same sweep): 128x64 wins or ties everywhere above 200k pixels (115-119 TF/s at full resolution), 64x64 below
  if (mfma) return M >= 200000 ? 2 : 3;
  if (M >= 200000) return (cout % 128 == 0 && M >= 1000000) ? 1 : 0;
  return 3;
}

static int igemm_launch(const float* in, const float* weight, const float* bias, float* out, const int32_t* geom_host,
                        int32_t flags, int nbatch, int64_t bs_in, int64_t bs_w, int64_t bs_out, void* stream) {
  ConvGeom g;
  static_assert(sizeof(ConvGeom) == 25 * sizeof(int), "ConvGeom layout");
  memcpy(&g, geom_host, sizeof(g));
  if (g.Cin % BK != 0 || g.Cout % 64 != 0 || g.ld_in % 4 != 0 || g.ld_out % 4 != 0 || g.B <= 0 || g.TH * g.TW <= 0) return MOPA_ERR_ARG;
  if ((((uintptr_t)in | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)bias) & 15) != 0) return MOPA_ERR_ARG;
  if (nbatch < 1 || nbatch > 65535 || ((bs_in | bs_w | bs_out) & 3)) return MOPA_ERR_ARG;
  const int64_t M = (int64_t)g.B * g.OHl * g.OWl;
  if (M <= 0) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int accumulate = flags & 1;
  static const bool env_mfma = [] { const char* e = getenv("MOPA_CONV2D_MFMA"); return !e || atoi(e) != 0; }();
  // the MFMA kernel addresses its operands with 32-bit byte offsets
  const bool use_mfma = env_mfma && (int64_t)g.B * g.IH * g.IW * g.ld_in < (1ll << 30) && (int64_t)BK * g.Cout < (1ll << 28);
  int tile = ((flags >> 8) & 0xff) ? ((flags >> 8) & 0xff) - 1 : pick_tile(M * nbatch, g.Cout, use_mfma);
  if (tile < 0 || tile > 3 || g.Cout % kTiles[tile].bn) return MOPA_ERR_ARG;
  dim3 grid((unsigned)cdiv64(M, kTiles[tile].bm), g.Cout / kTiles[tile].bn, nbatch);
  if (use_mfma) {
    // K-chunk depth 16.  (32 -- half the barriers per FLOP, twice the LDS and staging registers -- was measured: 128x64 tiles
    // 115 -> 100 TFLOP/s, joint step 214 -> 196 scans/s: resident blocks per CU matter more than barriers here; 8 -- twice the
    // resident blocks, half the MFMAs per barrier -- 222 -> 217.)
    switch (tile) {
      case 0: k_conv2d_igemm_mfma<256, 64, 64, 64, 16><<<grid, 256, 0, st>>>(in, weight, bias, out, g, accumulate, bs_in, bs_w, bs_out); break;
      case 1: k_conv2d_igemm_mfma<128, 128, 64, 64, 16><<<grid, 256, 0, st>>>(in, weight, bias, out, g, accumulate, bs_in, bs_w, bs_out); break;
      case 2: k_conv2d_igemm_mfma<128, 64, 64, 32, 16><<<grid, 256, 0, st>>>(in, weight, bias, out, g, accumulate, bs_in, bs_w, bs_out); break;
      default: k_conv2d_igemm_mfma<64, 64, 32, 32, 16><<<grid, 256, 0, st>>>(in, weight, bias, out, g, accumulate, bs_in, bs_w, bs_out); break;
    }
  } else {
    switch (tile) {
      case 0: k_conv2d_igemm<256, 64, 8, 8><<<grid, 256, 0, st>>>(in, weight, bias, out, g, accumulate, bs_in, bs_w, bs_out); break;
      case 1: k_conv2d_igemm<128, 128, 8, 8><<<grid, 256, 0, st>>>(in, weight, bias, out, g, accumulate, bs_in, bs_w, bs_out); break;
      case 2: k_conv2d_igemm<128, 64, 8, 4><<<grid, 256, 0, st>>>(in, weight, bias, out, g, accumulate, bs_in, bs_w, bs_out); break;
      default: k_conv2d_igemm<64, 64, 4, 4><<<grid, 256, 0, st>>>(in, weight, bias, out, g, accumulate, bs_in, bs_w, bs_out); break;
    }
  }
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// geom: 25 int32 in the ConvGeom order.  bias may be null.
// flags: bit 0 = accumulate (out += result, used for gradient accumulation); bits 8-15 = tile override + 1 (tuning only).
MOPA_API int mopa_conv2d_igemm(const float* in, const float* weight, const float* bias, float* out,
                               const int32_t* geom_host, int32_t flags, void* stream) {
  return igemm_launch(in, weight, bias, out, geom_host, flags, 1, 0, 0, 0, stream);
}

// nbatch independent convolutions of one geometry: problem z reads in + z*in_stride, weight + z*w_stride and writes
// out + z*out_stride (strides in floats).  Used for the 16 transformed points of the Winograd path (wino2d.hip).
MOPA_API int mopa_conv2d_igemm_batched(const float* in, const float* weight, float* out, const int32_t* geom_host,
                                       int32_t nbatch, int64_t in_stride, int64_t w_stride, int64_t out_stride, int32_t flags,
                                       void* stream) {
  return igemm_launch(in, weight, nullptr, out, geom_host, flags, nbatch, in_stride, w_stride, out_stride, stream);
}

// ----------------------------------------------------------------------------------------------
// Backward-weight: dW[tap][ci][co] = sum_m A[m][tap][ci] * dY[m][co]   (same index map as above; dY plays "out").
// Block = a 64x64 (ci, co) tile of NTAP horizontally adjacent taps (one filter row: the dY tile is staged once and
// reused by all of them) over a slice of the pixels; pixels are the GEMM K dimension, staged 16 at a time; each
// thread owns NTAP x 4 x 4 outputs.  Slice partials go to slabs [split][taps*Cin*Cout] and are summed in a fixed
// order by k_reduce_slabs2 (deterministic, no float atomics).
#define WBK 16
template <int WM, int NTAP>  // WM: 64, or 16 for the stem's 16-wide tap trick; NTAP: taps per block (divides TW)
__global__ __launch_bounds__(256) void k_conv2d_wgrad(const float* __restrict__ in, const float* __restrict__ dy,
                                                       float* __restrict__ slabs, const ConvGeom g, int m_per_split) {
  constexpr int WN = 64;
  constexpr int MT = WM / 16;  // micro rows per thread (ty has 16 values)
  __shared__ __attribute__((aligned(16))) float As[2][NTAP][WBK][WM + 4];
  __shared__ __attribute__((aligned(16))) float Bs[2][WBK][WN + 4];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int tap0 = blockIdx.x * NTAP;
  const int tiles_n = g.Cout / WN;
  const int ci0 = (blockIdx.y / tiles_n) * WM, co0 = (blockIdx.y % tiles_n) * WN;
  const int tyy = tap0 / g.TW, txx0 = tap0 - tyy * g.TW;
  const int M = g.B * g.OHl * g.OWl;
  const int mbeg = blockIdx.z * m_per_split, mend = min(M, mbeg + m_per_split);
  // staging: pixel k = t / 16, 4-float group = t % 16 (covers 64 floats per pixel row)
  const int sk = t >> 4, sq = t & 15;
  float acc[NTAP][MT][4];
#pragma unroll
  for (int n = 0; n < NTAP; ++n)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[n][i][j] = 0.f;
  float4 ra[NTAP], rb;
  // this thread's pixel (b, oy, ox) of the NEXT tile to load: decomposed once, then advanced by WBK pixels per tile
  int pb, poy, pox;
  {
    const int m = mbeg + sk;
    pb = m / (g.OHl * g.OWl);
    const int r = m - pb * g.OHl * g.OWl;
    poy = r / g.OWl;
    pox = r - poy * g.OWl;
  }
  auto load_tile = [&](int mb) {
    const int m = mb + sk;
    rb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < NTAP; ++n) ra[n] = rb;
    const int b = pb, oy = poy, ox = pox;
    pox += WBK;
    while (pox >= g.OWl) {
      pox -= g.OWl;
      if (++poy == g.OHl) { poy = 0; ++pb; }
    }
    if (m < mend) {
      const int iy = oy * g.IS + g.IY0 + tyy * g.IDY;
#pragma unroll
      for (int n = 0; n < NTAP; ++n) {
        const int ix = ox * g.IS + g.IX0 + (txx0 + n) * g.IDX;
        if (sq * 4 < WM && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW)
          ra[n] = *reinterpret_cast<const float4*>(in + ((int64_t)(b * g.IH + iy) * g.IW + ix) * g.ld_in + ci0 + sq * 4);
      }
      rb = *reinterpret_cast<const float4*>(dy + ((int64_t)(b * g.OHa + oy * g.OS + g.OOY) * g.OWa + ox * g.OS + g.OOX) * g.ld_out + co0 + sq * 4);
    }
  };
  auto store_tile = [&](int buf) {
    if (sq * 4 < WM) {
#pragma unroll
      for (int n = 0; n < NTAP; ++n) *reinterpret_cast<float4*>(&As[buf][n][sk][sq * 4]) = ra[n];
    }
    *reinterpret_cast<float4*>(&Bs[buf][sk][sq * 4]) = rb;
  };
  int buf = 0;
  if (mbeg < mend) {
    load_tile(mbeg);
    store_tile(0);
  }
  __syncthreads();
  for (int mb = mbeg; mb < mend; mb += WBK) {
    const bool more = mb + WBK < mend;
    if (more) load_tile(mb + WBK);
    // operands of pixel k+1 are read from LDS (into a second register set) before the FMAs of pixel k: the compiler's own
    // schedule reused one register quad for every read and waited for each of them (lgkmcnt(0) per ds_read)
    float4 bq[2], aq[2][NTAP];
#define WG_READ(SET, K)                                                                        \
  {                                                                                            \
    bq[SET] = *reinterpret_cast<const float4*>(&Bs[buf][K][tx * 4]);                           \
    _Pragma("unroll") for (int n = 0; n < NTAP; ++n) {                                         \
      if (MT == 4) aq[SET][n] = *reinterpret_cast<const float4*>(&As[buf][n][K][ty * 4]);      \
      else aq[SET][n].x = As[buf][n][K][ty];                                                   \
    }                                                                                          \
  }
    WG_READ(0, 0);
#pragma unroll
    for (int k = 0; k < WBK; ++k) {
      const int cur = k & 1;
      if (k + 1 < WBK) WG_READ(cur ^ 1, k + 1);
      const float4 b4 = bq[cur];
#pragma unroll
      for (int n = 0; n < NTAP; ++n) {
        const float a[4] = {aq[cur][n].x, aq[cur][n].y, aq[cur][n].z, aq[cur][n].w};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[n][i][0] = fmaf(a[i], b4.x, acc[n][i][0]); acc[n][i][1] = fmaf(a[i], b4.y, acc[n][i][1]);
          acc[n][i][2] = fmaf(a[i], b4.z, acc[n][i][2]); acc[n][i][3] = fmaf(a[i], b4.w, acc[n][i][3]);
        }
      }
    }
#undef WG_READ
    if (more) {
      store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
  const int64_t wsz = (int64_t)g.TH * g.TW * g.Cin * g.Cout;
#pragma unroll
  for (int n = 0; n < NTAP; ++n) {
    float* dst = slabs + (int64_t)blockIdx.z * wsz + ((int64_t)(tap0 + n) * g.Cin + ci0) * g.Cout + co0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int ci = (MT == 4) ? ty * 4 + i : ty;
      *reinterpret_cast<float4*>(dst + (int64_t)ci * g.Cout + tx * 4) = make_float4(acc[n][i][0], acc[n][i][1], acc[n][i][2], acc[n][i][3]);
    }
  }
}

// The same on the matrix cores (f32 operands, see k_conv2d_igemm_mfma): the LDS tiles are already pixel-major, which is the
// MFMA K dimension.  Block = 64 input channels x 64*NCO output channels x NTAP taps; wave w owns the (ci half, co half)
// quadrant: NTAP x NCO 32x32 tiles -> per pixel pair NTAP + NCO ds_read_b32 and NTAP * NCO v_mfma_f32_32x32x2_f32.
// Loads are `uniform base + 32-bit per-lane offset`, unconditional (pixels outside the image / the slice read element 0 and
// are zeroed when staged).  Cin % 64 == 0 only (the stem keeps the vector kernel).
template <int NTAP, int NCO>
__global__ __launch_bounds__(256) void k_conv2d_wgrad_mfma(const float* __restrict__ in, const float* __restrict__ dy,
                                                            float* __restrict__ slabs, const ConvGeom g, int m_per_split,
                                                            int64_t bs_in, int64_t bs_dy) {
  // bs_in != 0: blockIdx.x counts independent 1x1 problems (the 16 Winograd points) instead of filter taps
  const bool batched = bs_in != 0;
  in += batched ? blockIdx.x * bs_in : 0;
  dy += batched ? blockIdx.x * bs_dy : 0;
  constexpr int WM = 64, WN = 64 * NCO;
  __shared__ __attribute__((aligned(16))) float As[2][NTAP][WBK][WM + 4];
  __shared__ __attribute__((aligned(16))) float Bs[2][WBK][WN + 4];
  const int t = threadIdx.x;
  const int slab_tap = blockIdx.x * NTAP;
  const int tap0 = batched ? 0 : slab_tap;
  const int tiles_n = g.Cout / WN;
  const int ci0 = (blockIdx.y / tiles_n) * WM, co0 = (blockIdx.y % tiles_n) * WN;
  const int tyy = tap0 / g.TW, txx0 = tap0 - tyy * g.TW;
  const int M = g.B * g.OHl * g.OWl;
  const int mbeg = blockIdx.z * m_per_split, mend = min(M, mbeg + m_per_split);
  const int sk = t >> 4, sq = t & 15;   // staging: pixel k = t / 16, 4-float group = t % 16
  const int lane = t & 63, wv = t >> 6;
  const int wci = (wv & 1) * 32, wco = (wv >> 1) * 32 * NCO;
  const int l32 = lane & 31, lk = lane >> 5;
  f32x16 acc[NTAP][NCO];
#pragma unroll
  for (int n = 0; n < NTAP; ++n)
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][c][e] = 0.f;
  float4 ra[NTAP], rb[NCO];
  bool oka[NTAP], okb;
  int pb, poy, pox;
  {
    const int m = mbeg + sk;
    pb = m / (g.OHl * g.OWl);
    const int r = m - pb * g.OHl * g.OWl;
    poy = r / g.OWl;
    pox = r - poy * g.OWl;
  }
  const char* inb = reinterpret_cast<const char*>(in + ci0 + sq * 4);
  const char* dyb = reinterpret_cast<const char*>(dy + co0 + sq * 4);
  auto load_tile = [&](int mb) {
    const int b = pb, oy = poy, ox = pox;
    pox += WBK;
    while (pox >= g.OWl) {
      pox -= g.OWl;
      if (++poy == g.OHl) { poy = 0; ++pb; }
    }
    okb = mb + sk < mend;
    const int iy = oy * g.IS + g.IY0 + tyy * g.IDY;
    const bool rowok = okb && (unsigned)iy < (unsigned)g.IH;
    const int rowbase = (b * g.IH + iy) * g.IW;
#pragma unroll
    for (int n = 0; n < NTAP; ++n) {
      const int ix = ox * g.IS + g.IX0 + (txx0 + n) * g.IDX;
      oka[n] = rowok && (unsigned)ix < (unsigned)g.IW;
      const uint32_t off = oka[n] ? (uint32_t)((rowbase + ix) * g.ld_in) * 4u : 0u;
      ra[n] = *reinterpret_cast<const float4*>(inb + off);
    }
    const uint32_t offb = okb ? (uint32_t)(((b * g.OHa + oy * g.OS + g.OOY) * g.OWa + ox * g.OS + g.OOX) * g.ld_out) * 4u : 0u;
#pragma unroll
    for (int c = 0; c < NCO; ++c) rb[c] = *reinterpret_cast<const float4*>(dyb + offb + c * 256);
  };
  auto store_tile = [&](int buf) {
    // (component-wise selects: `ok ? ra[n] : zero` on the float4 becomes a pointer select and sends the arrays to scratch)
#pragma unroll
    for (int n = 0; n < NTAP; ++n)
      *reinterpret_cast<float4*>(&As[buf][n][sk][sq * 4]) =
          make_float4(oka[n] ? ra[n].x : 0.f, oka[n] ? ra[n].y : 0.f, oka[n] ? ra[n].z : 0.f, oka[n] ? ra[n].w : 0.f);
#pragma unroll
    for (int c = 0; c < NCO; ++c)
      *reinterpret_cast<float4*>(&Bs[buf][sk][c * 64 + sq * 4]) =
          make_float4(okb ? rb[c].x : 0.f, okb ? rb[c].y : 0.f, okb ? rb[c].z : 0.f, okb ? rb[c].w : 0.f);
  };
  int buf = 0;
  if (mbeg < mend) {
    load_tile(mbeg);
    store_tile(0);
  }
  __syncthreads();
  for (int mb = mbeg; mb < mend; mb += WBK) {
    const bool more = mb + WBK < mend;
    if (more) load_tile(mb + WBK);
#pragma unroll
    for (int kk = 0; kk < WBK; kk += 2) {
      float av[NTAP], bv[NCO];
#pragma unroll
      for (int c = 0; c < NCO; ++c) bv[c] = Bs[buf][kk + lk][wco + c * 32 + l32];
#pragma unroll
      for (int n = 0; n < NTAP; ++n) av[n] = As[buf][n][kk + lk][wci + l32];
#pragma unroll
      for (int n = 0; n < NTAP; ++n)
#pragma unroll
        for (int c = 0; c < NCO; ++c) acc[n][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[n], bv[c], acc[n][c], 0, 0, 0);
    }
    if (more) {
      store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
  const int64_t wsz = (int64_t)(batched ? (int)gridDim.x : g.TH * g.TW) * g.Cin * g.Cout;
#pragma unroll
  for (int n = 0; n < NTAP; ++n) {
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
      float* dst = slabs + (int64_t)blockIdx.z * wsz + ((int64_t)(slab_tap + n) * g.Cin + ci0 + wci) * g.Cout + co0 + wco + c * 32 + l32;
#pragma unroll
      for (int e = 0; e < 16; ++e) dst[(int64_t)(8 * (e >> 2) + 4 * lk + (e & 3)) * g.Cout] = acc[n][c][e];
    }
  }
}

// Weight gradient of the 7x7 stem (16-wide taps: Cin = 16, TH x TW = 7 x 2, Cout = 64) on the matrix cores.  One block owns ALL
// 14 taps over a slice of the pixels, so the output-gradient tile is read once (the per-filter-row blocks of the generic kernel
// read it 7 times: that kernel is bound by those 2 GB, not by its FMAs).  GEMM rows = (tap, ci) = 224 = 7 row tiles of 32 (one
// filter row each), 2 column tiles; wave w: column tile w & 1, row tiles (w >> 1), +2, +4(, +6).
// BNB: dy is the gradient of relu(batchnorm(conv)) -- the gradient w.r.t. the convolution's output, dx = scale * (dz - mean(dz) -
// xhat * mean(dz * xhat)) with dz = dy where the BatchNorm output was positive (the expressions of k_bn_bwd_apply), is formed while the
// tile is staged, from dy, the BatchNorm's input xbn, its stats[G][4][64] and coef[G][2][64] (mopa_bn_bwd_sums_groups): the stem's
// BatchNorm backward needs no apply pass (read dy + x, write dx: 1.8 GB at 16 images) and dx is never stored.
#define STEM_ROWS 224
struct StemBn { const float* xbn; const float* stats; const float* coef; int ld_x, ld_dy, imgs_per_group, n_groups, training; };
template <bool BNB>
__global__ __launch_bounds__(256) void k_stem_wgrad_mfma(const float* __restrict__ in, const float* __restrict__ dy,
                                                          float* __restrict__ slabs, const ConvGeom g, int m_per_split, const StemBn bn) {
  __shared__ __attribute__((aligned(16))) float As[2][WBK][STEM_ROWS + 4];
  __shared__ __attribute__((aligned(16))) float Bs[2][WBK][64 + 4];
  __shared__ __attribute__((aligned(16))) float cst[BNB ? 3 : 1][6][BNB ? 64 : 4];   // per group: scale, shift, mean, invstd, coef0, coef1
  const int t = threadIdx.x;
  if (BNB) {
    for (int i = t; i < bn.n_groups * 6 * 64; i += 256) {
      const int gi = i / 384, k = (i - gi * 384) >> 6, c = i & 63;
      cst[gi][k][c] = k < 4 ? bn.stats[(gi * 4 + k) * 64 + c] : bn.coef[(gi * 2 + (k - 4)) * 64 + c];
    }
    __syncthreads();
  }
  const int M = g.B * g.OHl * g.OWl, ohw = g.OHl * g.OWl;
  const int mbeg = blockIdx.x * m_per_split, mend = min(M, mbeg + m_per_split);
  const int lane = t & 63, wv = t >> 6;
  const int wco = (wv & 1) * 32, rt0 = wv >> 1;
  const int l32 = lane & 31, lk = lane >> 5;
  f32x16 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  // staging: A = 16 pixels x 56 float4 (14 taps x 4 quads) = 896 float4 -> thread t takes idx = t + 256 j; B = 16 x 16 float4
  int apx[4], aq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = t + 256 * j;
    apx[j] = idx / 56;
    aq[j] = idx - apx[j] * 56;
  }
  const int bpx = t >> 4, bq = t & 15;
  float4 ra[4], rb, rx;
  int rgi = 0;
  // pixel (b, oy, ox) of each of this thread's 5 loads in the NEXT chunk: decomposed once, then advanced by WBK pixels per chunk
  int pb[5], py[5], px[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int m = mbeg + (j < 4 ? apx[j] : bpx);
    pb[j] = m / ohw;
    const int r = m - pb[j] * ohw;
    py[j] = r / g.OWl;
    px[j] = r - py[j] * g.OWl;
  }
  auto load_tile = [&](int mb) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int b = pb[j], oy = py[j], ox = px[j];
      px[j] += WBK;
      while (px[j] >= g.OWl) {
        px[j] -= g.OWl;
        if (++py[j] == g.OHl) { py[j] = 0; ++pb[j]; }
      }
      if (j < 4) {
        ra[j] = z;
        if (t + 256 * j < 896 && mb + apx[j] < mend) {
          const int tap = aq[j] >> 2, ty = tap >> 1, tx = tap & 1;
          const int iy = oy * g.IS + g.IY0 + ty * g.IDY, ix = ox * g.IS + g.IX0 + tx * g.IDX;
          if ((unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW)
            ra[j] = *reinterpret_cast<const float4*>(in + ((int64_t)(b * g.IH + iy) * g.IW + ix) * g.ld_in + (aq[j] & 3) * 4);
        }
      } else {
        rb = z;
        if (BNB) { rx = z; rgi = b / bn.imgs_per_group; }
        if (mb + bpx < mend) {
          const int64_t pix = (int64_t)(b * g.OHa + oy * g.OS + g.OOY) * g.OWa + ox * g.OS + g.OOX;
          rb = *reinterpret_cast<const float4*>(dy + pix * (BNB ? bn.ld_dy : g.ld_out) + bq * 4);
          if (BNB) rx = *reinterpret_cast<const float4*>(bn.xbn + pix * bn.ld_x + bq * 4);
        } else if (BNB) rgi = -1;
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (t + 256 * j < 896) *reinterpret_cast<float4*>(&As[buf][apx[j]][aq[j] * 4]) = ra[j];
    if (BNB && rgi >= 0) {
      const float gs[4] = {rb.x, rb.y, rb.z, rb.w}, xs[4] = {rx.x, rx.y, rx.z, rx.w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = bq * 4 + j;
        const float sc = cst[rgi][0][c], sh = cst[rgi][1][c], mean = cst[rgi][2][c], inv = cst[rgi][3][c], c0 = cst[rgi][4][c], c1 = cst[rgi][5][c];
        const float yv = fmaf(xs[j], sc, sh);
        const float dz = yv > 0.f ? gs[j] : gs[j] * 0.f;
        if (bn.training) {
          const float xhat = (xs[j] - mean) * inv;
          o[j] = sc * (dz - c0 - xhat * c1);
        } else {
          o[j] = sc * dz;
        }
      }
      rb = make_float4(o[0], o[1], o[2], o[3]);
    }
    *reinterpret_cast<float4*>(&Bs[buf][bpx][bq * 4]) = rb;
  };
  int buf = 0;
  if (mbeg < mend) {
    load_tile(mbeg);
    store_tile(0);
  }
  __syncthreads();
  for (int mb = mbeg; mb < mend; mb += WBK) {
    const bool more = mb + WBK < mend;
    if (more) load_tile(mb + WBK);
#pragma unroll
    for (int kk = 0; kk < WBK; kk += 2) {
      const float bv = Bs[buf][kk + lk][wco + l32];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int rt = rt0 + 2 * n;
        if (rt < 7) {   // uniform per wave
          const float av = As[buf][kk + lk][rt * 32 + l32];
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
        }
      }
    }
    if (more) {
      store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
  float* dst = slabs + (int64_t)blockIdx.x * STEM_ROWS * 64 + wco + l32;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int rt = rt0 + 2 * n;
    if (rt < 7) {
#pragma unroll
      for (int e = 0; e < 16; ++e) dst[(int64_t)(rt * 32 + 8 * (e >> 2) + 4 * lk + (e & 3)) * 64] = acc[n][e];
    }
  }
}
static bool stem_wgrad_mfma(const ConvGeom& g) {
  static const bool env_mfma = [] { const char* e = getenv("MOPA_CONV2D_MFMA"); return !e || atoi(e) != 0; }();
  return env_mfma && g.Cin == 16 && g.TW == 2 && g.TH == 7 && g.Cout == 64;
}

// dw[i] (+)= sum_c slabs[c][i]: block = EW consecutive elements x 256 / EW slab lanes + fixed-order LDS reduction.  EW = 64 reads whole
// 256-byte runs of a slab row (16: 64-byte pieces -- 1 TB/s; the reduction of the 256- and 512-channel layers' slabs took 60 us a call);
// the narrower forms keep >= 512 blocks on the small gradients (reduce_ew).
template <int EW>
__global__ __launch_bounds__(256) void k_reduce_slabs2(const float* __restrict__ slabs, int nsplit, int64_t n, float* __restrict__ dw,
                                                        int accumulate, int oihw, int taps, int Cin, int Cout) {
  constexpr int NL = 256 / EW;
  __shared__ float red[NL][EW + 1];
  const int el = threadIdx.x % EW, cl = threadIdx.x / EW;
  const int64_t i = (int64_t)blockIdx.x * EW + el;
  float s = 0.f;
  if (i < n)
#pragma unroll 8
    for (int c = cl; c < nsplit; c += NL) s += slabs[(int64_t)c * n + i];
  red[cl][el] = s;
  __syncthreads();
  if (cl == 0 && i < n) {
    int64_t o = i;
    if (oihw) {   // i = (tap * Cin + ci) * Cout + co  ->  torch's parameter layout [co][ci][tap]
      const int co = (int)(i % Cout);
      const int64_t r = i / Cout;
      const int ci = (int)(r % Cin), tap = (int)(r / Cin);
      o = ((int64_t)co * Cin + ci) * taps + tap;
    }
    float t = accumulate ? dw[o] : 0.f;
#pragma unroll
    for (int k = 0; k < NL; ++k) t += red[k][el];
    dw[o] = t;
  }
}
static inline int reduce_ew(int64_t n) { return n >= 64 * 512 ? 64 : n >= 32 * 512 ? 32 : 16; }
static void reduce_slabs2_launch(const float* slabs, int ns, int64_t n, float* dw, int accumulate, int oihw, int taps, int Cin, int Cout,
                                 hipStream_t st) {
  const int ew = reduce_ew(n);
  const unsigned nb = (unsigned)cdiv64(n, ew);
  if (ew == 64) k_reduce_slabs2<64><<<nb, 256, 0, st>>>(slabs, ns, n, dw, accumulate, oihw, taps, Cin, Cout);
  else if (ew == 32) k_reduce_slabs2<32><<<nb, 256, 0, st>>>(slabs, ns, n, dw, accumulate, oihw, taps, Cin, Cout);
  else k_reduce_slabs2<16><<<nb, 256, 0, st>>>(slabs, ns, n, dw, accumulate, oihw, taps, Cin, Cout);
}

static int wgrad_ntap(const ConvGeom& g) {
  if (g.Cin >= 64 && g.TW % 3 == 0) return 3;
  if (g.Cin == 16 && g.TW == 2) return 2;  // stem: both 16-wide taps of a filter row share the dY tile
  return 1;
}

static bool wgrad_use_mfma(const ConvGeom& g) {
  static const bool env_mfma = [] { const char* e = getenv("MOPA_CONV2D_MFMA"); return !e || atoi(e) != 0; }();
  // 32-bit byte offsets into the input and the output-gradient images
  return env_mfma && g.Cin >= 64 && (int64_t)g.B * g.IH * g.IW * g.ld_in < (1ll << 30) &&
         (int64_t)g.B * g.OHa * g.OWa * g.ld_out < (1ll << 30);
}
// Output-channel tiles of 32 per wave in the MFMA kernel.  NCO = 2 (block = 64 x 128 channels: half the dY traffic, twice the
// MFMAs per barrier) was measured on the >= 128-channel layers: same kernel time at 248 VGPRs, and the split-K reduction
// grows with the larger split count -> 1.
static int wgrad_nco(const ConvGeom& g) { return 1; }

static void wgrad_split(const ConvGeom& g, int* nsplit, int* m_per_split) {
  const int64_t M = (int64_t)g.B * g.OHl * g.OWl;
  if (stem_wgrad_mfma(g)) {   // one block per pixel slice covers all taps and channels
    int64_t ns = cdiv64(M, 1152);
    if (ns > 1024) ns = 1024;
    const int64_t mps = cdiv64(cdiv64(M, ns), WBK) * WBK;
    *nsplit = (int)cdiv64(M, mps);
    *m_per_split = (int)mps;
    return;
  }
  const int wm = g.Cin >= 64 ? 64 : 16;
  const int64_t tiles = (int64_t)(g.TH * g.TW / wgrad_ntap(g)) * (g.Cin / wm) * (g.Cout / (64 * wgrad_nco(g)));
  int64_t ns = cdiv64(2048, tiles);   // (768 ... 3072 target blocks measured: +-1 % on the joint step)
  const int64_t maxs = cdiv64(M, 256);
  if (ns > maxs) ns = maxs;
  if (ns > 512) ns = 512;
  if (ns < 1) ns = 1;
  int64_t mps = cdiv64(cdiv64(M, ns), WBK) * WBK;
  *nsplit = (int)cdiv64(M, mps);
  *m_per_split = (int)mps;
}

MOPA_API size_t mopa_conv2d_wgrad_workspace_bytes(const int32_t* geom_host) {
  ConvGeom g;
  memcpy(&g, geom_host, sizeof(g));
  int ns, mps;
  wgrad_split(g, &ns, &mps);
  return align_up((size_t)ns * g.TH * g.TW * g.Cin * g.Cout * sizeof(float), 256);
}

// dweight[TH*TW][Cin][Cout] (logical taps, already in the igemm weight layout) (+)= sum_m A^T dY.
// flags: bit 0 = accumulate into dweight; bit 1 = dweight is torch's OIHW parameter (gradient) tensor [Cout][Cin][TH][TW]
// (only when the logical taps are the whole filter: forward geometry of a plain convolution).
MOPA_API int mopa_conv2d_bwd_weight(const float* in, const float* dy, float* dweight, const int32_t* geom_host,
                                    int32_t flags, void* ws, size_t ws_bytes, void* stream) {
  ConvGeom g;
  memcpy(&g, geom_host, sizeof(g));
  const int accumulate = flags & 1, oihw = (flags >> 1) & 1;
  if (oihw && (g.KS != 1 || g.KH0 != 0 || g.KW0 != 0 || g.KWF != g.TW)) return MOPA_ERR_ARG;
  if ((g.Cin % 64 != 0 && g.Cin != 16) || g.Cout % 64 != 0 || g.ld_in % 4 != 0 || g.ld_out % 4 != 0) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_conv2d_wgrad_workspace_bytes(geom_host)) return MOPA_ERR_WORKSPACE;
  int ns, mps;
  wgrad_split(g, &ns, &mps);
  hipStream_t st = (hipStream_t)stream;
  float* slabs = (float*)ws;
  const bool use_mfma = wgrad_use_mfma(g);
  const int nco = wgrad_nco(g);
  if (g.Cin >= 64) {
    const int ntap = wgrad_ntap(g) == 3 ? 3 : 1;
    dim3 grid(g.TH * g.TW / ntap, (g.Cin / 64) * (g.Cout / (64 * nco)), ns);
    if (use_mfma) {
      if (ntap == 3) k_conv2d_wgrad_mfma<3, 1><<<grid, 256, 0, st>>>(in, dy, slabs, g, mps, 0, 0);
      else k_conv2d_wgrad_mfma<1, 1><<<grid, 256, 0, st>>>(in, dy, slabs, g, mps, 0, 0);
    } else if (ntap == 3) {
      k_conv2d_wgrad<64, 3><<<grid, 256, 0, st>>>(in, dy, slabs, g, mps);
    } else {
      k_conv2d_wgrad<64, 1><<<grid, 256, 0, st>>>(in, dy, slabs, g, mps);
    }
  } else if (stem_wgrad_mfma(g)) {
    k_stem_wgrad_mfma<false><<<ns, 256, 0, st>>>(in, dy, slabs, g, mps, StemBn{});
  } else if (wgrad_ntap(g) == 2) {
    dim3 grid(g.TH * g.TW / 2, g.Cout / 64, ns);
    k_conv2d_wgrad<16, 2><<<grid, 256, 0, st>>>(in, dy, slabs, g, mps);
  } else {
    dim3 grid(g.TH * g.TW, g.Cout / 64, ns);
    k_conv2d_wgrad<16, 1><<<grid, 256, 0, st>>>(in, dy, slabs, g, mps);
  }
  const int64_t n = (int64_t)g.TH * g.TW * g.Cin * g.Cout;
  reduce_slabs2_launch(slabs, ns, n, dweight, accumulate, oihw, g.TH * g.TW, g.Cin, g.Cout, st);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// The stem's weight gradient with its BatchNorm's backward apply folded into the load of dy (k_stem_wgrad_mfma<true>): `dy` is the
// gradient of relu(batchnorm(conv)) (row stride ld_dy), xbn the BatchNorm's input = the convolution's output (row stride ld_x),
// stats = [n_groups][4][64] of the forward pass, coef = [n_groups][2][64] from mopa_bn_bwd_sums_groups; the B images are n_groups equal
// consecutive groups.  Same geometry, flags and workspace as mopa_conv2d_bwd_weight; the 7x7 / 16-wide-tap stem only.
MOPA_API int mopa_stem_bwd_weight_bn(const float* in, const float* dy, int32_t ld_dy, const float* xbn, int32_t ld_x, const float* stats,
                                     const float* coef, int32_t n_groups, int32_t training, float* dweight, const int32_t* geom_host,
                                     int32_t flags, void* ws, size_t ws_bytes, void* stream) {
  ConvGeom g;
  memcpy(&g, geom_host, sizeof(g));
  const int accumulate = flags & 1, oihw = (flags >> 1) & 1;
  if (!stem_wgrad_mfma(g) || g.OS != 1 || g.OOY != 0 || g.OOX != 0 || n_groups < 1 || n_groups > 3 || g.B % n_groups || ld_dy < 64 ||
      ld_x < 64 || ((ld_dy | ld_x | g.ld_in) & 3) || !xbn || !stats || !coef)
    return MOPA_ERR_ARG;
  if (oihw && (g.KS != 1 || g.KH0 != 0 || g.KW0 != 0 || g.KWF != g.TW)) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_conv2d_wgrad_workspace_bytes(geom_host)) return MOPA_ERR_WORKSPACE;
  int ns, mps;
  wgrad_split(g, &ns, &mps);
  hipStream_t st = (hipStream_t)stream;
  float* slabs = (float*)ws;
  StemBn bn{xbn, stats, coef, ld_x, ld_dy, g.B / n_groups, n_groups, training};
  k_stem_wgrad_mfma<true><<<ns, 256, 0, st>>>(in, dy, slabs, g, mps, bn);
  const int64_t n = (int64_t)g.TH * g.TW * g.Cin * g.Cout;
  reduce_slabs2_launch(slabs, ns, n, dweight, accumulate, oihw, g.TH * g.TW, g.Cin, g.Cout, st);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ----------------------------------------------------------------------------------------------
// Weight gradient of a Winograd-eligible 3x3 convolution in the transform domain (2.25x fewer multiplies than the direct
// form):  dU[p][ci][co] = sum_t V[p][t][ci] * dM[p][t][co]  -- 16 independent 1x1 weight gradients over the T tiles, run as
// ONE launch of the MFMA weight-gradient kernel (blockIdx.x = p), split over the tiles into slabs -- then
// dW[a][b] = (G^T dU G)[a][b] while the slabs are summed in order (k_wino_dw; deterministic).  V = mopa_wino_input(x),
// dM = mopa_wino_dout(dy) (wino2d.hip).  dweight: [3][3][Cin][Cout] (the igemm layout of mopa_conv2d_bwd_weight) or OIHW.
// block = 16 elements x 16 transform points: thread (e, p) sums its point's slabs in split order, then 16 threads transform
__global__ __launch_bounds__(256) void k_wino_dw(const float* __restrict__ slabs, int nsplit, int64_t n, float* __restrict__ dw,
                                                  int accumulate, int oihw, int Cin, int Cout) {
  __shared__ float red[16][17];
  const int e = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + e;
  float s = 0.f;
  if (i < n)
#pragma unroll 8
    for (int c = 0; c < nsplit; ++c) s += slabs[((int64_t)c * 16 + p) * n + i];
  red[p][e] = s;
  __syncthreads();
  if (p != 0 || i >= n) return;
  float u[4][4];
#pragma unroll
  for (int q = 0; q < 16; ++q) u[q >> 2][q & 3] = red[q][e];
  float t[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // G^T u
    t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
    t[1][j] = 0.5f * (u[1][j] - u[2][j]);
    t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
  }
  // igemm layout [tap][ci][co] (i = ci * Cout + co), or torch's OIHW [co][ci][tap]
  const int co = (int)(i % Cout), ci = (int)(i / Cout);
  const int64_t base = oihw ? ((int64_t)co * Cin + ci) * 9 : i, step = oihw ? 1 : n;
#pragma unroll
  for (int a = 0; a < 3; ++a) {  // (.) G
    float g0 = t[a][0] + 0.5f * (t[a][1] + t[a][2]), g1 = 0.5f * (t[a][1] - t[a][2]), g2 = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
    float* d = dw + base + (int64_t)(a * 3) * step;
    if (accumulate) { g0 += d[0]; g1 += d[step]; g2 += d[2 * step]; }
    d[0] = g0; d[step] = g1; d[2 * step] = g2;
  }
}

// F(4x4,3x3): 36 points, dW = G^T dU G with the 6x3 G.  block = EW consecutive elements x 256 / EW point lanes: thread (e, q) sums points
// q, q + 256 / EW, ... over the slabs in split order, then the first EW threads transform.  EW as in k_reduce_slabs2 (whole 256-byte runs
// of a slab row on the 256- / 512-channel layers: this kernel averaged 52 us a call with 64-byte pieces).
template <int EW>
__global__ __launch_bounds__(256) void k_wino4_dw(const float* __restrict__ slabs, int nsplit, int64_t n, float* __restrict__ dw,
                                                   int accumulate, int oihw, int Cin, int Cout) {
  constexpr int NQ = 256 / EW;
  __shared__ float red[36][EW + 1];
  const int e = threadIdx.x % EW, q = threadIdx.x / EW;
  const int64_t i = (int64_t)blockIdx.x * EW + e;
#pragma unroll
  for (int k = 0; k < (36 + NQ - 1) / NQ; ++k) {
    const int p = q + k * NQ;
    if (p < 36) {
      float s = 0.f;
      if (i < n) {
#pragma unroll 8
        for (int c = 0; c < nsplit; ++c) s += slabs[((int64_t)c * 36 + p) * n + i];
      }
      red[p][e] = s;
    }
  }
  __syncthreads();
  if (q != 0 || i >= n) return;
  float t[3][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {   // G^T u, column by column
    const float u0 = red[0 * 6 + j][e], u1 = red[1 * 6 + j][e], u2 = red[2 * 6 + j][e], u3 = red[3 * 6 + j][e], u4 = red[4 * 6 + j][e],
                u5 = red[5 * 6 + j][e];
    t[0][j] = 0.25f * u0 - (1.f / 6.f) * (u1 + u2) + (1.f / 24.f) * (u3 + u4);
    t[1][j] = (1.f / 6.f) * (u2 - u1) + (1.f / 12.f) * (u3 - u4);
    t[2][j] = -(1.f / 6.f) * (u1 + u2) + (1.f / 6.f) * (u3 + u4) + u5;
  }
  const int co = (int)(i % Cout), ci = (int)(i / Cout);
  const int64_t base = oihw ? ((int64_t)co * Cin + ci) * 9 : i, step = oihw ? 1 : n;
#pragma unroll
  for (int a = 0; a < 3; ++a) {   // (.) G
    float g0 = 0.25f * t[a][0] - (1.f / 6.f) * (t[a][1] + t[a][2]) + (1.f / 24.f) * (t[a][3] + t[a][4]);
    float g1 = (1.f / 6.f) * (t[a][2] - t[a][1]) + (1.f / 12.f) * (t[a][3] - t[a][4]);
    float g2 = -(1.f / 6.f) * (t[a][1] + t[a][2]) + (1.f / 6.f) * (t[a][3] + t[a][4]) + t[a][5];
    float* d = dw + base + (int64_t)(a * 3) * step;
    if (accumulate) { g0 += d[0]; g1 += d[step]; g2 += d[2 * step]; }
    d[0] = g0; d[step] = g1; d[2 * step] = g2;
  }
}

// (for wino4wg.hip: the slab sum + G^T dU G epilogue of an F(4x4) weight gradient whose slabs another kernel produced)
int wino4_dw_launch(const float* slabs, int nsplit, int Cin, int Cout, float* dweight, int flags, hipStream_t st) {
  const int64_t n = (int64_t)Cin * Cout;
  const int ew = reduce_ew(n), acc = flags & 1, oihw = (flags >> 1) & 1;
  const unsigned nb = (unsigned)cdiv64(n, ew);
  if (ew == 64) k_wino4_dw<64><<<nb, 256, 0, st>>>(slabs, nsplit, n, dweight, acc, oihw, Cin, Cout);
  else if (ew == 32) k_wino4_dw<32><<<nb, 256, 0, st>>>(slabs, nsplit, n, dweight, acc, oihw, Cin, Cout);
  else k_wino4_dw<16><<<nb, 256, 0, st>>>(slabs, nsplit, n, dweight, acc, oihw, Cin, Cout);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

static void wino_wgrad_split(int np, int64_t T, int Cin, int Cout, int* nsplit, int* m_per_split) {
  if (wgemm_tn_ok(np, T, Cin, Cout)) {   // the ring-buffered GEMM (wgemm.hip) has its own K ranges
    wgemm_tn_split(np, T, Cin, Cout, nsplit, m_per_split);
    return;
  }
  const int64_t tiles = (int64_t)np * (Cin / 64) * (Cout / 64);
  int64_t ns = cdiv64(2048, tiles);   // (768 ... 2048 target blocks: same joint step within 0.5 %)
  const int64_t maxs = cdiv64(T, 128);
  if (ns > maxs) ns = maxs;
  if (ns < 1) ns = 1;
  const int64_t mps = cdiv64(cdiv64(T, ns), WBK) * WBK;
  *nsplit = (int)cdiv64(T, mps);
  *m_per_split = (int)mps;
}

static size_t wino_wgrad_ws(int np, int32_t T, int32_t Cin, int32_t Cout) {
  if (T <= 0 || Cin < 64 || Cout < 64 || Cin % 64 || Cout % 64) return 0;   // (shapes wino_bwd_weight refuses)
  int ns, mps;
  wino_wgrad_split(np, T, Cin, Cout, &ns, &mps);
  return align_up((size_t)ns * np * Cin * Cout * sizeof(float), 256);
}

static int wino_bwd_weight(int np, const float* V, const float* dM, int32_t T, int32_t Cin, int32_t Cout, float* dweight, int32_t flags,
                           void* ws, size_t ws_bytes, void* stream) {
  if (T <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return MOPA_ERR_ARG;
  if ((int64_t)T * Cin >= (1ll << 30) || (int64_t)T * Cout >= (1ll << 30)) return MOPA_ERR_ARG;   // 32-bit byte offsets per point
  if (ws_bytes < wino_wgrad_ws(np, T, Cin, Cout)) return MOPA_ERR_WORKSPACE;
  int ns, mps;
  wino_wgrad_split(np, T, Cin, Cout, &ns, &mps);
  ConvGeom g;
  memset(&g, 0, sizeof(g));
  g.B = 1; g.IH = 1; g.IW = T; g.OHl = 1; g.OWl = T; g.OHa = 1; g.OWa = T;
  g.OS = 1; g.IS = 1; g.IDY = 1; g.IDX = 1; g.TH = 1; g.TW = 1; g.KS = 1; g.KWF = 1;
  g.Cin = Cin; g.Cout = Cout; g.ld_in = Cin; g.ld_out = Cout;
  hipStream_t st = (hipStream_t)stream;
  float* slabs = (float*)ws;
  if (wgemm_tn_ok(np, T, Cin, Cout)) {
    if (wgemm_tn_launch(np, V, dM, T, Cin, Cout, slabs, ns, mps, st) != MOPA_OK) return MOPA_ERR_LAUNCH;
  } else {
    dim3 grid(np, (Cin / 64) * (Cout / 64), ns);
    k_conv2d_wgrad_mfma<1, 1><<<grid, 256, 0, st>>>(V, dM, slabs, g, mps, (int64_t)T * Cin, (int64_t)T * Cout);
  }
  const int64_t n = (int64_t)Cin * Cout;
  if (np == 16) k_wino_dw<<<(unsigned)cdiv64(n, 16), 256, 0, st>>>(slabs, ns, n, dweight, flags & 1, (flags >> 1) & 1, Cin, Cout);
  else return wino4_dw_launch(slabs, ns, Cin, Cout, dweight, flags, st);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

MOPA_API size_t mopa_wino_wgrad_workspace_bytes(int32_t T, int32_t Cin, int32_t Cout) { return wino_wgrad_ws(16, T, Cin, Cout); }
// flags: bit 0 = accumulate into dweight; bit 1 = dweight is the OIHW parameter (gradient) tensor [Cout][Cin][3][3].
MOPA_API int mopa_wino_bwd_weight(const float* V, const float* dM, int32_t T, int32_t Cin, int32_t Cout, float* dweight, int32_t flags,
                                  void* ws, size_t ws_bytes, void* stream) {
  return wino_bwd_weight(16, V, dM, T, Cin, Cout, dweight, flags, ws, ws_bytes, stream);
}
// The same for F(4x4,3x3): V / dM hold 36 points (mopa_wino4_input / mopa_wino4_dout), T = B * ceil(H/4) * ceil(W/4).
MOPA_API size_t mopa_wino4_wgrad_workspace_bytes(int32_t T, int32_t Cin, int32_t Cout) { return wino_wgrad_ws(36, T, Cin, Cout); }
MOPA_API int mopa_wino4_bwd_weight(const float* V, const float* dM, int32_t T, int32_t Cin, int32_t Cout, float* dweight, int32_t flags,
                                   void* ws, size_t ws_bytes, void* stream) {
  return wino_bwd_weight(36, V, dM, T, Cin, Cout, dweight, flags, ws, ws_bytes, stream);
}

// ----------------------------------------------------------------------------------------------
// Weight re-layout between the parameter layout of torch (state_dict compatible) and the igemm layout.
//   mode 0: conv   OIHW (O,I,kh,kw) -> [kh][kw][I][O]           (forward)
//   mode 1: conv   OIHW             -> [kh][kw][O][I]           (backward-data; taps NOT flipped: the index map does it)
//   mode 2: convT  IOHW (I,O,kh,kw) -> [kh][kw][I][O]           (forward)
//   mode 3: convT  IOHW             -> [kh][kw][O][I]           (backward-data)
// dst is indexed [kh][kw][R][C]; inverse = 1 scatters a gradient in igemm layout back to the parameter layout
// (modes 0 and 2 only).
__global__ void k_relayout_w(const float* __restrict__ src, float* __restrict__ dst, int O, int I, int KH, int KW, int mode,
                             int inverse, int accumulate) {
  if (!inverse) {   // (shared with the batched refresh: weight_forms.h)
    wf_relayout_body(src, dst, O, I, KH, KW, mode, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
    return;
  }
  const int64_t n = (int64_t)O * I * KH * KW;
  const int R = (mode == 0 || mode == 2) ? I : O, C = (mode == 0 || mode == 2) ? O : I;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    // i indexes the igemm layout [kh][kw][R][C]
    const int c = (int)(i % C);
    int64_t r1 = i / C;
    const int r = (int)(r1 % R);
    r1 /= R;
    const int kw = (int)(r1 % KW), kh = (int)(r1 / KW);
    const int o = (mode == 0 || mode == 2) ? c : r, ii = (mode == 0 || mode == 2) ? r : c;
    const int64_t p = (mode < 2) ? (((int64_t)o * I + ii) * KH + kh) * KW + kw   // OIHW
                                 : (((int64_t)ii * O + o) * KH + kh) * KW + kw;  // IOHW
    dst[p] = (accumulate ? dst[p] : 0.f) + src[i];
  }
}

MOPA_API int mopa_conv2d_relayout_weight(const float* src, float* dst, int32_t O, int32_t I, int32_t KH, int32_t KW,
                                         int32_t mode, int32_t inverse, int32_t accumulate, void* stream) {
  if (O <= 0 || I <= 0 || KH <= 0 || KW <= 0 || mode < 0 || mode > 3 || (inverse && (mode & 1))) return MOPA_ERR_ARG;
  k_relayout_w<<<stream_grid((int64_t)O * I * KH * KW, 256), 256, 0, (hipStream_t)stream>>>(src, dst, O, I, KH, KW, mode,
                                                                                            inverse, accumulate);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// Stem (conv1 7x7, Cin=3): weights OIHW (64,3,7,7) <-> igemm layout [7][2][16][64] where tap (kh, tx) covers
// kw = 4*tx .. 4*tx+3 and channel slot 4 (kw=7 and c=3 are zero).  inverse = 1 maps a gradient back.
__global__ void k_stem_relayout(const float* __restrict__ src, float* __restrict__ dst, int O, int inverse, int accumulate) {
  const int n = 7 * 2 * 16 * O;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int o = i % O;
    int r = i / O;
    const int k16 = r % 16; r /= 16;
    const int tx = r % 2, kh = r / 2;
    const int kw = 4 * tx + k16 / 4, c = k16 % 4;
    const bool real = kw < 7 && c < 3;
    const int p = ((o * 3 + c) * 7 + kh) * 7 + kw;
    if (!inverse) dst[i] = real ? src[p] : 0.f;
    else if (real) dst[p] = (accumulate ? dst[p] : 0.f) + src[i];
  }
}
MOPA_API int mopa_conv2d_stem_relayout(const float* src, float* dst, int32_t O, int32_t inverse, int32_t accumulate, void* stream) {
  if (O <= 0) return MOPA_ERR_ARG;
  k_stem_relayout<<<stream_grid(7 * 2 * 16 * O, 256), 256, 0, (hipStream_t)stream>>>(src, dst, O, inverse, accumulate);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// img NCHW (B,3,H,W) -> zero-padded NHWC4 (B, Hp+6, Wp+8, 4): 3 px of conv padding on top/left, 3 bottom,
// 5 right (3 + 2 so the 16-float tap reads of the stem stay inside the row), channel 3 = 0;
// (Hp,Wp) = (H,W) rounded up to 16 (resnet34_unet.py:133-138).
__global__ void k_img_to_nhwc4(const float* __restrict__ img, int B, int H, int W, int Hp, int Wp, float* __restrict__ out) {
  const int PH = Hp + 6, PW = Wp + 8;
  const int64_t n = (int64_t)B * PH * PW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % PW) - 3;
    const int64_t r = i / PW;
    const int y = (int)(r % PH) - 3, b = (int)(r / PH);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
      const int64_t base = ((int64_t)b * 3 * H + y) * W + x;
      v.x = img[base]; v.y = img[base + (int64_t)H * W]; v.z = img[base + 2 * (int64_t)H * W];
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}
MOPA_API int mopa_img_to_nhwc4(const float* img, int32_t B, int32_t H, int32_t W, int32_t Hp, int32_t Wp, float* out, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Hp < H || Wp < W) return MOPA_ERR_ARG;
  k_img_to_nhwc4<<<stream_grid((int64_t)B * (Hp + 6) * (Wp + 8), 256), 256, 0, (hipStream_t)stream>>>(img, B, H, W, Hp, Wp, out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Gradient w.r.t. the input image (only when the caller passes an image with requires_grad: adversarial / saliency style
// use of the model -- MoPA's training never asks for it, so this is not a hot-path kernel, just the missing edge of the
// drop-in).  Backward-data of conv1 (7x7, stride 1, padding 3, 3 -> 64; resnet34_unet.py:144) restricted to the H x W window
// the image occupies inside the /16-padded frame:
//   dimg[b][c][y][x] = sum_{ky,kx,co} dc1[b][y+3-ky][x+3-kx][co] * w[co][c][ky][kx]        (terms outside the Hp x Wp map dropped)
// One thread per image pixel (16 x 16 tile per block), the filter in LDS as [tap][co] float4 (c0,c1,c2,0), broadcast reads;
// fixed summation order (tap-major, then co).
__global__ __launch_bounds__(256) void k_stem_dgrad_image(const float* __restrict__ dout, int ld, int Hp, int Wp, int H, int W,
                                                           const float* __restrict__ w, float* __restrict__ dimg) {
  __shared__ float4 wl[49 * 64];
  for (int i = threadIdx.x; i < 49 * 64; i += 256) {
    const int tap = i >> 6, co = i & 63;
    const float* p = w + (int64_t)co * 147 + tap;   // OIHW: ((co*3 + c)*7 + ky)*7 + kx
    wl[i] = make_float4(p[0], p[49], p[98], 0.f);
  }
  __syncthreads();
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4), b = blockIdx.z;
  if (x >= W || y >= H) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int ky = 0; ky < 7; ++ky) {
    const int py = y + 3 - ky;
    if ((unsigned)py >= (unsigned)Hp) continue;
    for (int kx = 0; kx < 7; ++kx) {
      const int px = x + 3 - kx;
      if ((unsigned)px >= (unsigned)Wp) continue;
      const float4* d = reinterpret_cast<const float4*>(dout + ((int64_t)(b * Hp + py) * Wp + px) * ld);
      const float4* wt = wl + (ky * 7 + kx) * 64;
#pragma unroll 4
      for (int c4 = 0; c4 < 16; ++c4) {
        const float4 v = d[c4];
        const float vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 ww = wt[c4 * 4 + j];
          a0 = fmaf(vs[j], ww.x, a0); a1 = fmaf(vs[j], ww.y, a1); a2 = fmaf(vs[j], ww.z, a2);
        }
      }
    }
  }
  const int64_t o = ((int64_t)b * 3 * H + y) * W + x;
  dimg[o] = a0; dimg[o + (int64_t)H * W] = a1; dimg[o + 2 * (int64_t)H * W] = a2;
}
// dout: gradient of conv1's output, NHWC [B][Hp][Wp] rows of `ld` floats (64 used); w: conv1.weight OIHW [64][3][7][7];
// dimg: NCHW (B,3,H,W), overwritten.
MOPA_API int mopa_stem_dgrad_image(const float* dout, int32_t ld, int32_t B, int32_t Hp, int32_t Wp, int32_t H, int32_t W,
                                   const float* w, float* dimg, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Hp < H || Wp < W || ld < 64 || (ld & 3) || (((uintptr_t)dout) & 15)) return MOPA_ERR_ARG;
  dim3 grid((unsigned)((W + 15) / 16), (unsigned)((H + 15) / 16), (unsigned)B);
  k_stem_dgrad_image<<<grid, 256, 0, (hipStream_t)stream>>>(dout, ld, Hp, Wp, H, W, w, dimg);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
