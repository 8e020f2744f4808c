// Weight gradient of a stride-1 3x3 convolution through Winograd F(4x4,3x3) in ONE kernel: neither V = B^T x B nor dM = A dY A^T
// reaches HBM (torch autograd's conv2d weight gradient behind /root/reference/mopa/models/resnet34_unet.py:97-110 -- layer1's
// BasicBlocks and the decoder's 3x3 convolutions at 152x240 / 304x480).
//
//   dU[p][ci][co] = sum_t V[p][t][ci] * dM[p][t][co]      p = 36 transform points, t = the T = B * ceil(H/4) * ceil(W/4) tiles
//   dW = G^T dU G                                         (k_wino4_dw, conv2d.hip: slabs summed in split order, deterministic)
//
// Why: on the 64 / 128-channel layers the two-operand form is HBM-bound -- V and dM are 2.25x the activations each, written once
// (mopa_wino4_input / mopa_wino4_dout) and read once by the batched GEMM: 672 MB per launch at 64 -> 64 / 152x240 / 16 images, 16 flop
// per byte -- and it forces the training forward pass to materialise V.  Here a workgroup reads x and dY (298 MB), transforms both in
// registers and multiplies out of LDS.
//
// Workgroup = 8 waves = two TEAMS of four, one workgroup per CU (108 KB of LDS, <= 256 registers).  A team owns a 32 (ci) x 64 (co)
// block of HALF the points -- the three rows a = 3 h .. 3 h + 2 of the 6 x 6 transform: the transforms are separable, so half the points
// cost half the column transforms -- over a contiguous range of tiles (split-K: one slab per range); the two teams of a workgroup take
// the two halves of the same block and range.  Per chunk of 8 tiles a team alternates two phases:
//   transform  thread (tile slot s = t / 32, c = t % 32) holds the 6x6 patch of x[., ci0 + c] and the 4x4 tile of dY[., co0 + 2 c .. + 1]
//              of tile s in registers (loaded one chunk ahead; interior tiles: one address per patch row + immediate column offsets;
//              tiles on an image border or beyond the range: clamped addresses, zeroed here; a deferred BatchNorm + ReLU on the way in
//              with the expression of k_bn_relu_apply, as mopa_wino4_input_bn), transforms both (12 / 8 operations per 6-point / 4-point
//              transform) and writes its 18 + 2 x 18 results into A[p][s][c], Bm[p][s][2 c .. + 1];
//   multiply   wave w = (co half w & 1, points 9 (w / 2) .. + 8 of the block's 18): per tile pair one ds_read_b32 per operand and point
//              (lane = (tile parity, channel): the v_mfma_f32_32x32x2_f32 operand layout as stored) and 9 MFMAs; 144 accumulator
//              registers live over the whole range.
// The teams run in ANTI-PHASE, held there by one workgroup barrier per half period: while one transforms (VALU + LDS stores) the
// other multiplies, so each SIMD always has one wave on the matrix pipe and one on the vector pipe.  (Two independent 4-wave
// workgroups per CU -- the first version -- fall into lockstep: both transform, then both multiply, and the phases add up:
// 216 us instead of the sum's parts 145 / 147 us measured with the probes below, 64 -> 64 at 16 x 152 x 240.)
#include "wino4.h"
#include <stdlib.h>
#include <stdio.h>

typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef float f32x2w __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

#define WG_TC 8          // tiles per chunk
#define WG_CI 32         // input channels per block
#define WG_CO 64         // output channels per block
#define WG_NP 36
#define WG_HP 18         // points per block (three rows of the 6 x 6 transform)

struct WgArgs {
  const float* in; const float* dy; float* slabs; const float* stats;
  int ld_in, ld_dy, B, H, W, Cin, Cout, th, tw, T, tiles_per_split, nsplit, imgs_per_group, bn_c0;
  long long* prof;   // -DWG_PROFILE: in-kernel cycle counters of two waves (one per team) of one workgroup
};

// 6-point input transform t = B^T d in 12 operations (the values of w4_bt6 up to rounding: t1 / t2 = p +- q with p = d4 - 4 d2,
// q = d3 - 4 d1; t3 / t4 = r +- 2 s with r = d4 - d2, s = d3 - d1), and its two halves (rows 0-2 / rows 3-5) in 6 each
__device__ __forceinline__ void wg_bt_lo(float d0, float d1, float d2, float d3, float d4, float& t0, float& t1, float& t2) {
  t0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
  const float p = fmaf(-4.f, d2, d4), q = fmaf(-4.f, d1, d3);
  t1 = p + q;
  t2 = p - q;
}
__device__ __forceinline__ void wg_bt_hi(float d1, float d2, float d3, float d4, float d5, float& t3, float& t4, float& t5) {
  const float r = d4 - d2, s = d3 - d1;
  t3 = fmaf(2.f, s, r);
  t4 = fmaf(-2.f, s, r);
  t5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}
// 4 -> 6 output-gradient transform r = A d in 8 operations (r0 = d0, r5 = d3), halves in 4 each
__device__ __forceinline__ void wg_a_lo(float d0, float d1, float d2, float d3, float& r0, float& r1, float& r2) {
  const float s02 = d0 + d2, s13 = d1 + d3;
  r0 = d0;
  r1 = s02 + s13;
  r2 = s02 - s13;
}
__device__ __forceinline__ void wg_a_hi(float d0, float d1, float d2, float d3, float& r3, float& r4, float& r5) {
  const float u = fmaf(4.f, d2, d0), v = fmaf(4.f, d3, d1);
  r3 = fmaf(2.f, v, u);
  r4 = fmaf(-2.f, v, u);
  r5 = d3;
}

template <bool BN, int LDI, int LDD>   // LDI / LDD: the row strides of x / dY when they are compile-time constants (0: run-time values)
__global__ __launch_bounds__(512, 2) void k_wino4_wgrad(const WgArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[WG_NP * WG_TC * (WG_CI + WG_CO)];
  float* __restrict__ As = lds;                             // [36][8][32]
  float* __restrict__ Bs = lds + WG_NP * WG_TC * WG_CI;     // [36][8][64]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;   // (not readfirstlane: with scalar role branches the register allocation spills 34 instead of 5)
  // blockIdx.x -> (tile range, channel block pair): the `by` blocks of one tile range sit on ONE XCD (ids that are equal modulo 8
  // share an XCD's L2: they read the same x / dY rows), eight ranges side by side
  const int nco = a.Cout / WG_CO;
  const int by = (a.Cin / WG_CI) * nco;
  const int grp = blockIdx.x / (8 * by), r8 = blockIdx.x - grp * 8 * by;
  const int cb = r8 >> 3, split = grp * 8 + (r8 & 7);
  const int ci0 = (cb / nco) * WG_CI, co0 = (cb % nco) * WG_CO;
#ifdef WG_PROFILE
  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
  const long long tstart = tprev;
#define WG_T(K_) { const long long n_ = __builtin_readcyclecounter(); pt[K_] += n_ - tprev; tprev = n_; }
#else
#define WG_T(K_)
#endif
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.T, t_begin + a.tiles_per_split);
  // Roles: waves 0-3 transform x -- thread = (tile slot ts, input channel cq); waves 4-7 transform dY -- thread = (tile slot ts, output
  // channel PAIR cq): the two halves of the workgroup do about the same arithmetic (144 / 160 operations) and every raw byte of the
  // chunk is loaded ONCE per workgroup (the kernel is bound by what its CUs can pull from L2 / memory, ~5.5 TB/s measured over the
  // chip: a first version that split the transform points between two independent workgroups loaded everything twice and took 220 us
  // on 64 -> 64 at 16 x 152 x 240 whatever else was changed).
  const bool xrole = wv < 4;
  const int ts = (t & 255) >> 5, cq = t & 31;
  const int H = a.H, W = a.W, tw = a.tw, th = a.th, thw = a.th * a.tw;
  const uint32_t ldi4 = (LDI ? (uint32_t)LDI : (uint32_t)a.ld_in) * 4u, ldd4 = (LDD ? (uint32_t)LDD : (uint32_t)a.ld_dy) * 4u;

  f32x16w acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

  // this thread's tile of the current chunk: image tb, tile row ty, tile column tx (advanced by 8 tiles per chunk)
  int tile = t_begin + ts;
  int tb = tile / thw, ty, tx;
  { const int rt = tile - tb * thw; ty = rt / tw; tx = rt - ty * tw; }

  float raw[36];   // x role: the 6x6 patch (element (i, j) = raw[6 i + j]); dY role: the 4x4 tile of two channels (raw[2 (4 i + j) + e])
  bool ltv = false, lint = false;   // the loaded tile's validity; lint (wave-uniform): every tile of this wave's load was an interior tile
  int lty = 0, ltx = 0;
  float sc = 1.f, sh = 0.f;
  const char* __restrict__ inb = reinterpret_cast<const char*>(a.in);    // uniform bases + 32-bit per-lane byte offsets (saddr loads)
  const char* __restrict__ dyb = reinterpret_cast<const char*>(a.dy);
  const uint32_t chi4 = (uint32_t)(ci0 + cq) * 4u, chd4 = (uint32_t)(co0 + 2 * cq) * 4u;
  const bool bn_ch = BN && (ci0 + cq >= a.bn_c0);

  // The next chunk's raw operand is loaded at the START of the multiplication (unconditional loads; tiles on an image border or beyond
  // the range read CLAMPED pixels and are zeroed when they are transformed) and lands while the matrix pipe works.
  // The next chunk's raw operand is loaded at the START of the multiplication (unconditional loads; tiles on an image border or beyond
  // the range read CLAMPED pixels and are zeroed when they are transformed) and lands while the matrix pipe works.
  auto issue_loads = [&]() {
    ltv = tile < t_end;
    lty = ty; ltx = tx;
    // interior: the whole 6x6 patch (and with it the 4x4 tile) lies inside the image
    lint = __all(ltv && ty >= 1 && tx >= 1 && 4 * ty + 4 < H && 4 * tx + 4 < W) != 0;
    if (xrole) {
      if (lint) {
        const uint32_t rs = (uint32_t)W * ldi4;
        uint32_t o = (uint32_t)((tb * H + 4 * ty - 1) * W + 4 * tx - 1) * ldi4 + chi4;
#pragma unroll
        for (int i = 0; i < 6; ++i, o += rs)
#pragma unroll
          for (int j = 0; j < 6; ++j) raw[6 * i + j] = *reinterpret_cast<const float*>(inb + (o + (uint32_t)j * ldi4));
      } else {
        const int ab = ltv ? tb : 0, ay = ltv ? ty : 0, ax = ltv ? tx : 0;
        uint32_t ro[6], co[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int y = min(max(4 * ay - 1 + i, 0), H - 1), x = min(max(4 * ax - 1 + i, 0), W - 1);
          ro[i] = (uint32_t)((ab * H + y) * W) * ldi4;
          co[i] = (uint32_t)x * ldi4 + chi4;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) raw[6 * i + j] = *reinterpret_cast<const float*>(inb + (ro[i] + co[j]));
      }
      if (BN) {
        if (bn_ch) {
          const float* __restrict__ sg = a.stats + (int64_t)((ltv ? tb : 0) / a.imgs_per_group) * 4 * (a.Cin - a.bn_c0) + (ci0 + cq - a.bn_c0);
          sc = sg[0];
          sh = sg[a.Cin - a.bn_c0];
        }
      }
    } else {
      if (lint) {
        const uint32_t rsd = (uint32_t)W * ldd4;
        uint32_t od = (uint32_t)((tb * H + 4 * ty) * W + 4 * tx) * ldd4 + chd4;
#pragma unroll
        for (int i = 0; i < 4; ++i, od += rsd)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x2w v = *reinterpret_cast<const f32x2w*>(dyb + (od + (uint32_t)j * ldd4));
            raw[2 * (4 * i + j)] = v[0];
            raw[2 * (4 * i + j) + 1] = v[1];
          }
      } else {
        const int ab = ltv ? tb : 0, ay = ltv ? ty : 0, ax = ltv ? tx : 0;
        uint32_t rd[4], cd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int y = min(4 * ay + i, H - 1), x = min(4 * ax + i, W - 1);
          rd[i] = (uint32_t)((ab * H + y) * W) * ldd4;
          cd[i] = (uint32_t)x * ldd4 + chd4;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x2w v = *reinterpret_cast<const f32x2w*>(dyb + (rd[i] + cd[j]));
            raw[2 * (4 * i + j)] = v[0];
            raw[2 * (4 * i + j) + 1] = v[1];
          }
      }
    }
  };
  auto advance = [&]() {
    tile += WG_TC;
    tx += WG_TC;
    while (tx >= tw) {
      tx -= tw;
      if (++ty == th) { ty = 0; ++tb; }
    }
  };

#ifdef WG_PROBE_NOLOAD   // timing probe: no global loads inside the loop (the first chunk's registers are transformed again and again)
#define WG_ISSUE()
#else
#define WG_ISSUE() issue_loads()
#endif
  issue_loads();
  float* __restrict__ aw = As + (t & 255);                                  // A[p][ts][cq]: lane-linear stores
  f32x2w* __restrict__ bw = reinterpret_cast<f32x2w*>(Bs) + (t & 255);      // Bm[p][ts][2 cq .. + 1]: lane-linear 8-byte stores
  const int l32 = lane & 31, lk = lane >> 5, coh = wv & 1, pg = wv >> 1;
  const float* __restrict__ ar = As + (pg * 9) * (WG_TC * WG_CI) + lane;
  const float* __restrict__ br = Bs + (pg * 9) * (WG_TC * WG_CO) + lk * WG_CO + coh * 32 + l32;

  // Every range has the same number of chunks (tiles beyond its end are zero tiles).
  const int nchunk = a.tiles_per_split / WG_TC;
  for (int kk = 0; kk < nchunk; ++kk) {
    WG_T(0)
#ifdef WG_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_T(1)   // waiting for the prefetched patch / tile
#endif
    // ---- transform the loaded tile: V = B^T d B (x role) or dM = A dY A^T (dY role), 36 points
    if (xrole) {
      if (!lint || BN) {   // (wave-uniform unless BN: border tiles are zeroed, the deferred BatchNorm is applied)
        bool rok[6], cok[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          rok[i] = ltv && (unsigned)(4 * lty - 1 + i) < (unsigned)H;
          cok[i] = (unsigned)(4 * ltx - 1 + i) < (unsigned)W;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            float v = raw[6 * i + j];
            if (BN) {
              const float o = fmaf(v, sc, sh);
              v = bn_ch ? (o > 0.f ? o : o * 0.f) : v;
            }
            raw[6 * i + j] = (rok[i] && cok[j]) ? v : 0.f;
          }
      }
      float m[6][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {   // B^T d, column by column
        wg_bt_lo(raw[j], raw[6 + j], raw[12 + j], raw[18 + j], raw[24 + j], m[0][j], m[1][j], m[2][j]);
        wg_bt_hi(raw[6 + j], raw[12 + j], raw[18 + j], raw[24 + j], raw[30 + j], m[3][j], m[4][j], m[5][j]);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {   // (.) B
        float v[6];
        wg_bt_lo(m[i][0], m[i][1], m[i][2], m[i][3], m[i][4], v[0], v[1], v[2]);
        wg_bt_hi(m[i][1], m[i][2], m[i][3], m[i][4], m[i][5], v[3], v[4], v[5]);
#pragma unroll
        for (int j = 0; j < 6; ++j) aw[(i * 6 + j) * (WG_TC * WG_CI)] = v[j];
      }
    } else {
      if (!lint) {
        bool rok[4], cok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rok[i] = ltv && 4 * lty + i < H;
          cok[i] = 4 * ltx + i < W;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) raw[2 * (4 * i + j) + e] = (rok[i] && cok[j]) ? raw[2 * (4 * i + j) + e] : 0.f;
      }
      float r[2][6][4];   // the thread's two output channels: A dY, column by column
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wg_a_lo(raw[2 * j + e], raw[2 * (4 + j) + e], raw[2 * (8 + j) + e], raw[2 * (12 + j) + e], r[e][0][j], r[e][1][j], r[e][2][j]);
          wg_a_hi(raw[2 * j + e], raw[2 * (4 + j) + e], raw[2 * (8 + j) + e], raw[2 * (12 + j) + e], r[e][3][j], r[e][4][j], r[e][5][j]);
        }
#pragma unroll
      for (int i = 0; i < 6; ++i) {   // (.) A^T; the channel pair goes out as one 8-byte store (lane-linear: conflict-free)
        float m0[6], m1[6];
        wg_a_lo(r[0][i][0], r[0][i][1], r[0][i][2], r[0][i][3], m0[0], m0[1], m0[2]);
        wg_a_hi(r[0][i][0], r[0][i][1], r[0][i][2], r[0][i][3], m0[3], m0[4], m0[5]);
        wg_a_lo(r[1][i][0], r[1][i][1], r[1][i][2], r[1][i][3], m1[0], m1[1], m1[2]);
        wg_a_hi(r[1][i][0], r[1][i][1], r[1][i][2], r[1][i][3], m1[3], m1[4], m1[5]);
#pragma unroll
        for (int j = 0; j < 6; ++j) bw[(i * 6 + j) * (WG_TC * WG_CO / 2)] = (f32x2w){m0[j], m1[j]};
      }
    }
#ifdef WG_PROFILE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    WG_T(2)   // transform + LDS stores
    __syncthreads();   // A, Bm of this chunk are complete
    WG_T(3)
    // ---- the next chunk's patch / tile: in flight during the multiplication
    if (kk + 1 < nchunk) {
      advance();
      WG_ISSUE();
    }
    WG_T(5)
    // ---- multiply: four tile pairs, 9 points (one tile pair's operands at a time: registers)
#pragma unroll
    for (int s = 0; s < WG_TC / 2; ++s) {
      float av[9], bv[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        av[q] = ar[(q * WG_TC + 2 * s) * WG_CI];
        bv[q] = br[(q * WG_TC + 2 * s) * WG_CO];
      }
#ifndef WG_PROBE_NOMFMA
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc[q], 0, 0, 0);
#else   // timing probe (wrong results on purpose): operands read, no matrix instruction
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q][0] += av[q] * bv[q];
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    WG_T(4)   // LDS operand reads + MFMA issue
    __syncthreads();   // everyone is done reading A, Bm
    WG_T(6)   // barrier
  }
#ifdef WG_PROFILE
  if (blockIdx.x == 100 && (threadIdx.x & 255) == 0) {
    const int w_ = threadIdx.x >> 8;   // role
    for (int k = 0; k < 8; ++k) a.prof[w_ * 16 + k] = pt[k];
    a.prof[w_ * 16 + 8] = __builtin_readcyclecounter() - tstart;
  }
#endif
  // ---- slab [split][p][Cin][Cout]: accumulator register e of a 32x32 tile = row (ci) 8 (e / 4) + 4 (lane / 32) + e % 4, column (co) lane % 32
#ifdef WG_PROBE_NOEPI   // timing probe: one accumulator block per wave is stored instead of nine
  if (acc[0][0] != 12345.f) {
#pragma unroll
    for (int q = 1; q < 9; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][e] += acc[q][e];
  }
#define WG_EPI_Q 1
#else
#define WG_EPI_Q 9
#endif
  float* __restrict__ dst = a.slabs + (((int64_t)split * WG_NP + pg * 9) * a.Cin + ci0) * a.Cout + co0 + coh * 32 + l32;
#pragma unroll
  for (int q = 0; q < WG_EPI_Q; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      dst[((int64_t)q * a.Cin + 8 * (e >> 2) + 4 * lk + (e & 3)) * a.Cout] = acc[q][e];
}

static void wg_plan(int64_t T, int Cin, int Cout, int ncu, int* nsplit, int* tiles_per_split) {
  const int per_cu = 1;   // (one 8-wave workgroup per CU and round: two fall into lockstep, profiles/r6_wino4_wgrad.md)
  const int by = (Cin / WG_CI) * (Cout / WG_CO);
  int64_t ns = ((int64_t)ncu * per_cu + by - 1) / by;   // ~ per_cu workgroups per CU in one round
  ns = (ns + 7) / 8 * 8;
  const int64_t maxs = cdiv64(T, 4 * WG_TC) / 8 * 8;   // at least four chunks per range
  if (ns > maxs) ns = maxs;
  if (ns < 8) ns = 8;
  const int64_t tps = cdiv64(cdiv64(T, ns), WG_TC) * WG_TC;
  *nsplit = (int)ns;
  *tiles_per_split = (int)tps;
}

// Shapes the one-kernel weight gradient takes: input channels in blocks of 32, output channels in blocks of 64, at most 128 of either
// (the slabs grow with Cin x Cout, the x / dY re-reads with the number of channel blocks) -- the layers where the two-operand form is
// bandwidth-bound.  Which of them USE it is the caller's table (mopa_amd/dense2d.py).
MOPA_API int mopa_wino4_wgrad_fused_ok(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
  if (B <= 0 || H < 4 || W < 4 || Cin <= 0 || Cout <= 0 || Cin % WG_CI || Cout % WG_CO) return 0;
  if (Cin > 128 || Cout > 128) return 0;
  const int64_t T = (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4);
  return T < (1 << 30) && (int64_t)B * H * W * 256 < (1ll << 30);
}

MOPA_API size_t mopa_wino4_wgrad_fused_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
  if (!mopa_wino4_wgrad_fused_ok(B, H, W, Cin, Cout)) return 0;
  const int ncu = mopa_cu_count();
  int ns, tps;
  wg_plan((int64_t)B * ((H + 3) / 4) * ((W + 3) / 4), Cin, Cout, ncu > 0 ? ncu : 256, &ns, &tps);
  return align_up((size_t)ns * WG_NP * Cin * Cout * sizeof(float), 256);
}

// dweight (+)= the weight gradient of out = conv3x3(in, pad 1) given dy = dL/d(out), both NHWC with row strides ld_in / ld_dy.
// stats != null: `in` is a BatchNorm's input and relu(batchnorm(in)) is what was convolved -- stats = [n_groups][4][Cin - bn_c0]
// (mopa_bn_act_fwd_groups), the B images are n_groups equal consecutive groups, channels below bn_c0 pass through (mopa_wino4_input_bn).
// flags: bit 0 = accumulate into dweight; bit 1 = dweight is torch's OIHW tensor [Cout][Cin][3][3] (else [3][3][Cin][Cout]).
MOPA_API int mopa_wino4_wgrad_fused(const float* in, int32_t ld_in, const float* stats, int32_t n_groups, int32_t bn_c0, const float* dy,
                                    int32_t ld_dy, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* dweight,
                                    int32_t flags, void* ws, size_t ws_bytes, void* stream) {
  if (!mopa_wino4_wgrad_fused_ok(B, H, W, Cin, Cout) || ld_in < Cin || ld_dy < Cout || (ld_dy & 1) || ((uintptr_t)dy & 7) || !in || !dy || !dweight)
    return MOPA_ERR_ARG;
  if ((int64_t)B * H * W * ld_in >= (1ll << 30) || (int64_t)B * H * W * ld_dy >= (1ll << 30)) return MOPA_ERR_ARG;   // 32-bit byte offsets
  if (stats && (n_groups < 1 || B % n_groups || bn_c0 < 0 || bn_c0 >= Cin)) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_wino4_wgrad_fused_workspace_bytes(B, H, W, Cin, Cout)) return MOPA_ERR_WORKSPACE;
  const int ncu = mopa_cu_count();
  if (ncu <= 0) return MOPA_ERR_LAUNCH;
  WgArgs a;
  a.in = in; a.dy = dy; a.slabs = (float*)ws; a.stats = stats;
  a.ld_in = ld_in; a.ld_dy = ld_dy; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.th = (H + 3) / 4; a.tw = (W + 3) / 4; a.T = B * a.th * a.tw;
  wg_plan(a.T, Cin, Cout, ncu, &a.nsplit, &a.tiles_per_split);
  a.imgs_per_group = stats ? B / n_groups : 1;
  a.bn_c0 = stats ? bn_c0 : 0;
  a.prof = nullptr;
#ifdef WG_PROFILE
  static long long* prof = nullptr;
  if (!prof) hipMallocManaged(&prof, 512);
  a.prof = prof;
#endif
  hipStream_t st = (hipStream_t)stream;
  const int by = (Cin / WG_CI) * (Cout / WG_CO);
  const unsigned nblk = (unsigned)(a.nsplit * by);
  // row strides as compile-time constants where the network's buffers have them (immediate column offsets in the loads)
#define WG_GO(LI, LD)                                                   \
  {                                                                     \
    if (stats) k_wino4_wgrad<true, LI, LD><<<nblk, 512, 0, st>>>(a);    \
    else k_wino4_wgrad<false, LI, LD><<<nblk, 512, 0, st>>>(a);         \
  }
  if (ld_in == 64 && ld_dy == 64) WG_GO(64, 64)
  else if (ld_in == 128 && ld_dy == 64) WG_GO(128, 64)
  else if (ld_in == 128 && ld_dy == 128) WG_GO(128, 128)
  else if (ld_in == 64 && ld_dy == 128) WG_GO(64, 128)
  else WG_GO(0, 0)
#undef WG_GO
  MOPA_CHECK_LAUNCH();
#ifdef WG_PROFILE
  hipStreamSynchronize(st);
  for (int tm = 0; tm < 2; ++tm) {
    const double n_ = (double)a.tiles_per_split / WG_TC;
    const long long* p_ = a.prof + tm * 16;
    printf("[wg profile] block 100 %s wave, cycles per chunk (%.0f chunks): loop head %.0f | load wait %.0f | transform + lds stores %.0f | barrier %.0f | "
           "load issue %.0f | operand reads + mfma issue %.0f | barrier %.0f | whole loop %.0f\n",
           tm ? "dY" : "x", n_, p_[0] / n_, p_[1] / n_, p_[2] / n_, p_[3] / n_, p_[5] / n_, p_[4] / n_, p_[6] / n_, p_[8] / n_);
  }
#endif
  return wino4_dw_launch(a.slabs, a.nsplit, Cin, Cout, dweight, flags, st);
}
