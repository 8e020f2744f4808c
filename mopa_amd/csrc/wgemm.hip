// Transform-domain weight gradient of the 128- to 512-channel Winograd layers as a ring-buffered GEMM (the deep half of torch autograd's
// conv2d weight gradients behind /root/reference/mopa/models/resnet34_unet.py:99-110 -- layer2 .. layer4, dec_conv_stage3 / 4):
//
//   dU[p][ci][co] = sum_t V[p][t][ci] * dM[p][t][co]        p = 16 or 36 transform points, t = tiles (the GEMM's K dimension)
//
// Both operands are K-major as they lie in memory (a row = one tile's channels), which is the v_mfma_f32_32x32x2_f32 operand order
// (lane = (k parity, channel)): a K-chunk is a straight copy into LDS -- done by LDS-DMA (global_load_lds_dwordx4: no staging
// registers, 1 KB per instruction) into a ring of three stages, two chunks in flight behind counted vmcnt and ONE raw s_barrier per
// chunk; operand reads are inline asm (the compiler answers every LDS read it can see after a DMA with vmcnt(0), which drains the
// ring).  Workgroup = 4 waves on a 128 x 128 (ci, co) block of one point over a K range (split-K slabs, summed in order by
// k_wino4_dw / k_wino_dw: deterministic); wave = 64 x 64 = 2 x 2 MFMA tiles: per K pair 4 ds_read_b32 + 4 MFMAs, 32 MFMAs per chunk and
// wave between barriers (k_conv2d_wgrad_mfma's 64 x 64 block: 8, with register staging).  Needs Cin % 128 == Cout % 128 == 0 and
// T % 16 == 0 (a K tail would need zero rows: LDS-DMA cannot mask); everything else stays on k_conv2d_wgrad_mfma.
#include "wino4.h"
#include <stdlib.h>

typedef float f32x16g __attribute__((ext_vector_type(16)));

#define WGM_B 128     // block tile (both sides)
#define WGM_K 16      // tiles per chunk
#define WGM_NST 3     // ring stages
#define WGM_STAGE (2 * WGM_K * WGM_B)   // floats per stage: A[16][128] + B[16][128]

__global__ __launch_bounds__(256, 3) void k_wgemm_tn(const float* __restrict__ V, const float* __restrict__ dM, float* __restrict__ slabs,
                                                      int T, int Cin, int Cout, int np, int k_per_split) {
  __shared__ __attribute__((aligned(16))) float lds[WGM_NST * WGM_STAGE];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntn = Cout / WGM_B, ntile = (Cin / WGM_B) * ntn;
  const int p = blockIdx.x / ntile, tl = blockIdx.x - p * ntile;
  const int ci0 = (tl / ntn) * WGM_B, co0 = (tl % ntn) * WGM_B;
  const int split = blockIdx.y;
  const int k_begin = split * k_per_split, k_end = min(T, k_begin + k_per_split);
  const int NU = (k_end - k_begin) / WGM_K;   // whole chunks (T % 16 == 0, k_per_split % 16 == 0)

  f32x16g acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // DMA: wave w copies rows [4 w, 4 w + 4) of both operand chunks: 2 + 2 instructions of 1 KB (lane = (row parity, 16-byte column))
  typedef const __attribute__((address_space(1))) char* gptr_t;
  const int drow = 4 * wv + (lane >> 5), dcol = (lane & 31) * 16;
  gptr_t dma_a = (gptr_t)(V + ((int64_t)p * T + k_begin) * Cin + ci0) + (int64_t)drow * Cin * 4 + dcol;
  gptr_t dma_b = (gptr_t)(dM + ((int64_t)p * T + k_begin) * Cout + co0) + (int64_t)drow * Cout * 4 + dcol;
  const int64_t a_rows2 = (int64_t)2 * Cin * 4, b_rows2 = (int64_t)2 * Cout * 4;          // two rows further
  const int64_t a_step = (int64_t)WGM_K * Cin * 4, b_step = (int64_t)WGM_K * Cout * 4;    // next chunk
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[0];
  const unsigned dma_l = lds0 + (unsigned)wv * 2048u;   // + stage * STAGE bytes (+ 1024 for the second row pair, + 8192 for B)
  unsigned dma_st = 0;
  int dma_left = NU;
#define WGM_DMA()                                                                                                   \
  if (dma_left > 0) {                                                                                               \
    const unsigned l_ = dma_l + dma_st * (WGM_STAGE * 4);                                                           \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_a),                        \
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(l_), 16, 0, 0);           \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_a + a_rows2),              \
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(l_ + 1024u), 16, 0, 0);   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_b),                        \
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(l_ + 8192u), 16, 0, 0);   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_b + b_rows2),              \
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(l_ + 9216u), 16, 0, 0);   \
    dma_a += a_step; dma_b += b_step;                                                                               \
    --dma_left;                                                                                                     \
    dma_st = dma_st == WGM_NST - 1 ? 0u : dma_st + 1u;                                                              \
  }
  WGM_DMA();
  WGM_DMA();

  // operand reads: lane = (k parity lk, column l32): A[2 kk + lk][64 wm + 32 i + l32], B[2 kk + lk][64 wn + 32 j + l32]
  const int l32 = lane & 31, lk = lane >> 5, wm = wv & 1, wn = wv >> 1;
  const unsigned ra = lds0 + (unsigned)(lk * 512 + (64 * wm + l32) * 4), rb = lds0 + 8192u + (unsigned)(lk * 512 + (64 * wn + l32) * 4);
  unsigned st = 0;
  for (int u = 0; u < NU; ++u) {
    if (u + 1 < NU) __builtin_amdgcn_s_waitcnt(0x0F74);   // vmcnt(4): this wave's part of chunk u has landed, chunk u + 1 may be in flight
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();   // everyone's part of chunk u has landed; everyone is done reading chunk u - 1 (the slot chunk u + 2 goes to)
    WGM_DMA();
    const unsigned sa = ra + st * (WGM_STAGE * 4), sb = rb + st * (WGM_STAGE * 4);
    st = st == WGM_NST - 1 ? 0u : st + 1u;
    float a0, a1, b0, b1, c0, c1, d0, d1;
#define WGM_RD(KK, A0, A1, B0, B1)                                                                                                   \
  asm volatile("ds_read_b32 %0, %4 offset:%6\n\tds_read_b32 %1, %4 offset:%7\n\tds_read_b32 %2, %5 offset:%6\n\tds_read_b32 %3, %5 offset:%7" \
               : "=&v"(A0), "=&v"(A1), "=&v"(B0), "=&v"(B1)                                                                            \
               : "v"(sa), "v"(sb), "i"((KK) * 1024), "i"((KK) * 1024 + 128))
#define WGM_WAIT(N, A0, A1, B0, B1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1))
#define WGM_MM(A0, A1, B0, B1)                                                      \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, acc[0][0], 0, 0, 0);     \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B1, acc[0][1], 0, 0, 0);     \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B0, acc[1][0], 0, 0, 0);     \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, acc[1][1], 0, 0, 0);
    WGM_RD(0, a0, a1, b0, b1);
    WGM_RD(1, c0, c1, d0, d1);
    WGM_WAIT(4, a0, a1, b0, b1);
    WGM_MM(a0, a1, b0, b1)
    WGM_RD(2, a0, a1, b0, b1);
    WGM_WAIT(4, c0, c1, d0, d1);
    WGM_MM(c0, c1, d0, d1)
    WGM_RD(3, c0, c1, d0, d1);
    WGM_WAIT(4, a0, a1, b0, b1);
    WGM_MM(a0, a1, b0, b1)
    WGM_RD(4, a0, a1, b0, b1);
    WGM_WAIT(4, c0, c1, d0, d1);
    WGM_MM(c0, c1, d0, d1)
    WGM_RD(5, c0, c1, d0, d1);
    WGM_WAIT(4, a0, a1, b0, b1);
    WGM_MM(a0, a1, b0, b1)
    WGM_RD(6, a0, a1, b0, b1);
    WGM_WAIT(4, c0, c1, d0, d1);
    WGM_MM(c0, c1, d0, d1)
    WGM_RD(7, c0, c1, d0, d1);
    WGM_WAIT(4, a0, a1, b0, b1);
    WGM_MM(a0, a1, b0, b1)
    WGM_WAIT(0, c0, c1, d0, d1);
    WGM_MM(c0, c1, d0, d1)
  }
  // slab [split][p][Cin][Cout]: accumulator register e of a 32x32 tile = row (ci) 8 (e / 4) + 4 lk + e % 4, column (co) l32
  float* __restrict__ dst = slabs + (((int64_t)split * np + p) * Cin + ci0 + 64 * wm) * Cout + co0 + 64 * wn + l32;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        dst[(int64_t)(32 * i + 8 * (e >> 2) + 4 * lk + (e & 3)) * Cout + 32 * j] = acc[i][j][e];
}

// Does the ring-buffered GEMM take this transform-domain weight gradient?  (MOPA_WGEMM=0: never.)
bool wgemm_tn_ok(int np, int64_t T, int Cin, int Cout) {
  static const bool on = [] { const char* e = getenv("MOPA_WGEMM"); return !e || atoi(e) != 0; }();
  return on && (np == 16 || np == 36) && Cin % WGM_B == 0 && Cout % WGM_B == 0 && T % WGM_K == 0 && T >= 4 * WGM_K &&
         T * Cin < (1ll << 30) && T * Cout < (1ll << 30);
}

// K ranges.  Equal blocks on 256 CUs quantise: 576 blocks are 3 on a quarter of the CUs and 2 on the rest, and the launch lasts as long
// as the CUs with 3 (measured: 256 -> 256 at 2400 tiles 87 TF/s with 4 ranges).  Pick the count that minimises
//   ceil(blocks / CUs) * chunks per block * (one chunk of 32 MFMAs per wave) + the slabs written and read once at ~4 TB/s.
void wgemm_tn_split(int np, int64_t T, int Cin, int Cout, int* nsplit, int* k_per_split) {
  const int64_t items = (int64_t)np * (Cin / WGM_B) * (Cout / WGM_B), chunks = T / WGM_K;
  int ncu = mopa_cu_count();
  if (ncu <= 0) ncu = 256;
  const double t_chunk = 0.975e-6, slab = (double)np * Cin * Cout * 4;
  int64_t best = 1;
  double best_t = 1e30;
  for (int64_t ns = 1; ns <= 32 && ns * 4 <= chunks; ++ns) {
    const double tt = (double)cdiv64(items * ns, ncu) * (double)cdiv64(chunks, ns) * t_chunk + 2.0 * ns * slab / 4e12;
    if (tt < best_t * 0.999) { best_t = tt; best = ns; }
  }
  const int64_t kps = cdiv64(chunks, best) * WGM_K;
  *nsplit = (int)cdiv64(T, kps);
  *k_per_split = (int)kps;
}

int wgemm_tn_launch(int np, const float* V, const float* dM, int T, int Cin, int Cout, float* slabs, int nsplit, int k_per_split,
                    hipStream_t st) {
  dim3 grid((unsigned)(np * (Cin / WGM_B) * (Cout / WGM_B)), (unsigned)nsplit);
  k_wgemm_tn<<<grid, 256, 0, st>>>(V, dM, slabs, T, Cin, Cout, np, k_per_split);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}
