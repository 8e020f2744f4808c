// Sparse 3D convolution, offset-major: gather -> per-offset GEMM -> ordered reduce (round 5).
//
// The output-stationary kernels of spconv.hip cut every 64-row tile's rules of one filter offset into 16-rule MFMA groups:
// 49-78 % of a group carries a rule, and every group re-fetches its weight fragments.  For the layers that are bound by the
// matrix pipe (levels >= 2-3 of the UNet, the wide 2C -> C blocks, the short 8-offset tables) this file flips the loop:
//
//   run-major rulebook   per table, the rules of filter offset o are ONE contiguous run of slots (in output-row order), each run
//                        padded to whole 128-slot items: an MFMA row group is full (fill >= 0.95 instead of 0.49-0.78);
//   k_spconv_run         a work item = 128 (or 64) consecutive slots of ONE offset x a column group: the block keeps the column
//                        slice of W[o] in LDS for as long as its items stay on that offset, every wave gathers its 16-rule row
//                        groups straight into the MFMA operand layout, accumulates over the whole Cin in registers, and writes
//                        the products of slot s to row s of a partial slab P[slot][Cout] (whole float4 rows), or -- for tables
//                        with one rule per output row (Deconvolution k2s2 and the backward-data of Convolution k2s2) -- straight
//                        to the output row;
//   k_run_reduce         out[i] = sum over the offsets o (ascending: a fixed order) of P[pos[o][i]]: one deterministic pass, each
//                        output element written once.  No float atomics anywhere.
//
// This is north_star's "gather-GEMM-scatter ... MFMA only for the dense per-rule GEMM"  with the scatter replaced by
// an ordered gather; semantics: mopa/models/scn_unet.py:25-30 (SubmanifoldConvolution, scn.UNet Convolution / Deconvolution),
// SURVEY Appendix A.4 / A.5, oracle: oracle/scn3d.py::sparse_conv.  Products and sums are exact fp32 (v_mfma_f32_16x16x4_f32).
#include "common.h"
#include "sprun_pack.h"
#include <atomic>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x4r __attribute__((ext_vector_type(4)));

#define RUN_PAD 128        // every offset's run is padded to whole items of this many slots
#define RUN_HDR 128        // int32 header words in front of the arrays
#define RUN_CHUNK 1024     // table entries one wave compacts
// header: [0 .. K] first slot of offset o's run ([K] = all slots), [32 .. 32 + K) rules of offset o, [64] slots (= [K])

// int32 layout of one table's run-major rulebook: header | run_in[cap] | run_out[cap] | pos[K][A] | chunk counters [K][nchunk]
__host__ __device__ static inline int64_t run_cap(int K, int64_t A) { return (K * A + (int64_t)K * RUN_PAD + RUN_PAD - 1) / RUN_PAD * RUN_PAD; }
static inline int64_t run_nchunk(int64_t A) { return (A + RUN_CHUNK - 1) / RUN_CHUNK; }
MOPA_API size_t mopa_rulebook_runs_bytes(int32_t K, int32_t num_out) {
  if (K <= 0 || K > 27 || num_out <= 0) return 0;
  const int64_t cap = run_cap(K, num_out);
  return (size_t)(RUN_HDR + 2 * cap + (int64_t)K * num_out + (int64_t)K * run_nchunk(num_out)) * sizeof(int32_t);
}

#define RUNB_MAX 24
struct RunDescs { int64_t nbr[RUNB_MAX], buf[RUNB_MAX]; int32_t K[RUNB_MAX], A[RUNB_MAX], wave0[RUNB_MAX]; };
__device__ __forceinline__ int run_find(const RunDescs& d, int n, int w) {
  int t = 0;
  for (int k = 1; k < n; ++k)
    if (w >= d.wave0[k]) t = k;
  return t;
}

// one wave per (table, offset, chunk of RUN_CHUNK rows): valid entries of the chunk
__global__ __launch_bounds__(64) void k_run_count(const RunDescs d, int n) {
  const int t = run_find(d, n, blockIdx.x);
  const int K = d.K[t], A = d.A[t];
  const int nch = (A + RUN_CHUNK - 1) / RUN_CHUNK;
  const int w = blockIdx.x - d.wave0[t], o = w / nch, c = w - o * nch;
  const int* __restrict__ nbr = reinterpret_cast<const int*>(d.nbr[t]) + (int64_t)o * A;
  int* buf = reinterpret_cast<int*>(d.buf[t]);
  const int64_t cap = run_cap(K, A);
  int* cnt = buf + RUN_HDR + 2 * cap + (int64_t)K * A;
  int nv = 0;
#pragma unroll 4
  for (int i = 0; i < RUN_CHUNK / 64; ++i) {
    const int row = c * RUN_CHUNK + i * 64 + threadIdx.x;
    const int v = row < A ? nbr[row] : -1;
    nv += __popcll(__ballot(v >= 0));
  }
  if (threadIdx.x == 0) cnt[o * nch + c] = nv;
}

// one block per table: exclusive scan of the chunk counters in (offset, chunk) order, every offset's run starting on a RUN_PAD
// boundary; the counters become the chunks' first slots
__global__ __launch_bounds__(256) void k_run_scan(const RunDescs d) {
  __shared__ int part[256];
  __shared__ int base_s;
  const int t = blockIdx.x, tid = threadIdx.x;
  const int K = d.K[t], A = d.A[t];
  const int nch = (A + RUN_CHUNK - 1) / RUN_CHUNK;
  int* buf = reinterpret_cast<int*>(d.buf[t]);
  const int64_t cap = run_cap(K, A);
  int* cnt = buf + RUN_HDR + 2 * cap + (int64_t)K * A;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int o = 0; o < K; ++o) {
    const int base = base_s;
    int run = 0;   // rules of this offset in front of the current batch of 256 chunks
    for (int c0 = 0; c0 < nch; c0 += 256) {
      const int c = c0 + tid;
      const int v = c < nch ? cnt[o * nch + c] : 0;
      part[tid] = v;
      __syncthreads();
      for (int s = 1; s < 256; s <<= 1) {   // Hillis-Steele inclusive scan
        const int add = tid >= s ? part[tid - s] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
      }
      if (c < nch) cnt[o * nch + c] = base + run + part[tid] - v;
      run += part[255];
      __syncthreads();
    }
    if (tid == 0) {
      buf[o] = base;
      buf[32 + o] = run;
      base_s = base + (run + RUN_PAD - 1) / RUN_PAD * RUN_PAD;
    }
    __syncthreads();
  }
  if (tid == 0) { buf[K] = base_s; buf[64] = base_s; }
}

// one wave per (table, offset, chunk): slots of the chunk's rules, pos[o][row], and -- last chunk of the offset -- the run's padding
__global__ __launch_bounds__(64) void k_run_fill(const RunDescs d, int n) {
  const int t = run_find(d, n, blockIdx.x);
  const int K = d.K[t], A = d.A[t];
  const int nch = (A + RUN_CHUNK - 1) / RUN_CHUNK;
  const int w = blockIdx.x - d.wave0[t], o = w / nch, c = w - o * nch;
  const int lane = threadIdx.x;
  const int* __restrict__ nbr = reinterpret_cast<const int*>(d.nbr[t]) + (int64_t)o * A;
  int* buf = reinterpret_cast<int*>(d.buf[t]);
  const int64_t cap = run_cap(K, A);
  int* __restrict__ run_in = buf + RUN_HDR;
  int* __restrict__ run_out = run_in + cap;
  int* __restrict__ pos = run_out + cap + (int64_t)o * A;
  const int* cnt = buf + RUN_HDR + 2 * cap + (int64_t)K * A;
  int slot = cnt[o * nch + c];
#pragma unroll 4
  for (int i = 0; i < RUN_CHUNK / 64; ++i) {
    const int row = c * RUN_CHUNK + i * 64 + lane;
    const int v = row < A ? nbr[row] : -1;
    const unsigned long long bal = __ballot(v >= 0);
    const int p = slot + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    if (v >= 0) { run_in[p] = v; run_out[p] = row; }
    if (row < A) pos[row] = v >= 0 ? p : -1;
    slot += __popcll(bal);
  }
  if (c == nch - 1) {   // slot = end of the run's rules: pad to the next offset's first slot
    const int end = (slot + RUN_PAD - 1) / RUN_PAD * RUN_PAD;
    for (int p = slot + lane; p < end; p += 64) { run_in[p] = -1; run_out[p] = -1; }
  }
}

// Run-major rulebooks of n tables in three launches.  desc_host [n][4] int64: rule table nbr[K][A] (device), K, A, destination
// buffer (device, mopa_rulebook_runs_bytes(K, A) bytes).  No host synchronisation: the buffers are sized from the bound K * A.
MOPA_API int mopa_rulebook_runs_build_batched(const int64_t* desc_host, int32_t n, void* stream) {
  if (!desc_host || n <= 0 || n > RUNB_MAX) return MOPA_ERR_ARG;
  RunDescs d;
  memset(&d, 0, sizeof(d));
  int64_t waves = 0;
  for (int t = 0; t < n; ++t) {
    const int64_t* r = desc_host + (int64_t)t * 4;
    if (!r[0] || !r[3] || r[1] <= 0 || r[1] > 27 || r[2] <= 0 || r[1] * r[2] >= (1ll << 30)) return MOPA_ERR_ARG;
    d.nbr[t] = r[0]; d.K[t] = (int)r[1]; d.A[t] = (int)r[2]; d.buf[t] = r[3];
    d.wave0[t] = (int)waves;
    waves += r[1] * run_nchunk(r[2]);
    if (waves >= (1ll << 30)) return MOPA_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  k_run_count<<<(unsigned)waves, 64, 0, st>>>(d, n);
  k_run_scan<<<n, 256, 0, st>>>(d);
  k_run_fill<<<(unsigned)waves, 64, 0, st>>>(d, n);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ----------------------------------------------------------------------------------------------
__global__ void k_run_pack_w(const float* __restrict__ w, int K, int cin_w, int cout_w, int transpose, int nt, float* __restrict__ wr) {
  const int n = K * cin_w * cout_w;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) run_pack_elem(w, wr, i, K, cin_w, cout_w, transpose, nt);
}
MOPA_API int mopa_spconv_run_pack_weight(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t transpose, float* wr, void* stream) {
  const int cin_c = transpose ? cout : cin, cout_c = transpose ? cin : cout;
  const int nt = run_nt(cin_c, cout_c);
  if (!w || !wr || K <= 0 || K > 27 || nt == 0) return MOPA_ERR_ARG;
  const int64_t n = (int64_t)K * cin * cout;
  k_run_pack_w<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(w, K, cin, cout, transpose, nt, wr);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ----------------------------------------------------------------------------------------------
// The per-offset GEMM.  Block = 4 waves, persistent over a contiguous range of items (an item = 64 RG consecutive slots of one
// offset; runs are padded to RUN_PAD = 128 slots, so an item never straddles two offsets).  Wave w owns slots
// [item + 16 RG w, + 16 RG): RG MFMA row groups x NT column tiles, accumulated in registers over the whole Cin.
//   operands (k permuted consistently so that every load is 16 contiguous bytes per lane): lane l, r = l & 15, q = l >> 4
//     rows   in[run_in[slot r]][16 kc + 4 q + s]       one global float4 per row group and 16-channel chunk, one chunk ahead
//     W      Wc[o][16 kc + 4 q + s][16 t + r]          one ds_read_b128 per tile and chunk out of the block's LDS slice
//   the MFMA takes the weight fragment as its A operand and the rows as B, so the result tile is TRANSPOSED: lane (r, q) holds
//   output channels 16 t + 4 q .. + 3 of slot r -- one float4 store per tile and row group into the slab / the output row.
template <int NT, int RG, bool SCATTER>
__global__ __launch_bounds__(256) void k_spconv_run(const int* __restrict__ hdr, const int* __restrict__ run_in, const int* __restrict__ run_out,
                                                     int K, const float* in, int ld_in, int cin,
                                                     const float* Wr, int ncg, int w_flip,
                                                     float* __restrict__ dst, int ld_dst) {
  constexpr int IT = 64 * RG;
  extern __shared__ float4 run_smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = blockIdx.y;
  const int nkc = cin >> 4;
  const int n_items = hdr[64] / IT;
  const int i0 = (int)((int64_t)blockIdx.x * n_items / gridDim.x), i1 = (int)((int64_t)(blockIdx.x + 1) * n_items / gridDim.x);
  const int slice4 = nkc * NT * 64;   // float4 per (offset, column group) slice
  const char* __restrict__ in_b = reinterpret_cast<const char*>(in);
  const unsigned a_off = (unsigned)(q * 16);
  int cur_o = -1, o = 0, o_end = hdr[1], o_rules_end = hdr[0] + hdr[32];
  for (int it = i0; it < i1; ++it) {
    const int slot0 = it * IT;
    while (slot0 >= o_end) {   // block-uniform: the offset whose run holds this item
      ++o;
      o_end = hdr[o + 1];
      o_rules_end = hdr[o] + hdr[32 + o];
    }
    const bool stage = o != cur_o;
    if (stage) {   // stage the slice of W[o]: a straight copy (the packed form is the LDS image) in 1-KB pieces, wave w takes pieces
      __syncthreads();   // w, w + 4, ...: eight loads in flight per lane, wave-uniform control flow only
      const float4* src = reinterpret_cast<const float4*>(Wr) + ((int64_t)(w_flip ? K - 1 - o : o) * ncg + cg) * slice4 + lane;   // (no __restrict__: see below)
      float4* dstl = run_smem + lane;
      const int np = nkc * NT;
      for (int p = wv; p < np; p += 32) {
#define RUN_LD(U) const float4 v##U = src[min(p + 4 * U, np - 1) * 64];   /* unconditional, in bounds */
#define RUN_ST(U) if (p + 4 * U < np) dstl[(p + 4 * U) * 64] = v##U;
        RUN_LD(0) RUN_LD(1) RUN_LD(2) RUN_LD(3) RUN_LD(4) RUN_LD(5) RUN_LD(6) RUN_LD(7)
        asm volatile("" ::: "memory");   // keeps the loads together (the compiler would sink each one to its store)
        RUN_ST(0) RUN_ST(1) RUN_ST(2) RUN_ST(3) RUN_ST(4) RUN_ST(5) RUN_ST(6) RUN_ST(7)
#undef RUN_LD
#undef RUN_ST
      }
      cur_o = o;
    }
    const int wslot = slot0 + wv * 16 * RG;
    const bool idle = wslot >= o_rules_end;   // a wave of padding slots (wave-uniform)
    unsigned arow[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const int ir = run_in[wslot + g * 16 + r];   // (padding slots hold -1: in bounds either way)
      arow[g] = (unsigned)(ir < 0 ? 0 : ir) * (unsigned)(ld_in * 4) + a_off;
    }
    if (stage) __syncthreads();   // (the rows' indices are on their way meanwhile)
    if (idle) continue;
    f32x4r acc[RG][NT];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[g][t] = (f32x4r){0.f, 0.f, 0.f, 0.f};
    // two row sets alternate, the other set's rows are in flight while one is multiplied (unconditional, in-bounds loads: the last
    // chunk is fetched again instead of a branch).  The clobbers pin the loads where they are written: without them the compiler
    // folds the loop-carried loads into one load at the loop head (phi of loads -> load of phi) and every chunk waits for its rows.
    float4 a0[RG], a1[RG];
#define RUN_LOAD(A, KC)                                                                        \
  {                                                                                            \
    const unsigned ko_ = (unsigned)(min((KC), nkc - 1) * 64);                                  \
    _Pragma("unroll") for (int g = 0; g < RG; ++g) A[g] = *reinterpret_cast<const float4*>(in_b + (arow[g] + ko_)); \
    asm volatile("" ::: "memory");                                                             \
  }
#define RUN_MM(A, KC)                                                                          \
  {                                                                                            \
    const float4* wl_ = run_smem + (KC) * NT * 64 + lane;                                      \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                           \
      const float4 b_ = wl_[t * 64];                                                           \
      const float bv_[4] = {b_.x, b_.y, b_.z, b_.w};                                           \
      _Pragma("unroll") for (int g = 0; g < RG; ++g) {                                         \
        const float av_[4] = {A[g].x, A[g].y, A[g].z, A[g].w};                                 \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                       \
          acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv_[s_], av_[s_], acc[g][t], 0, 0, 0); \
      }                                                                                        \
    }                                                                                          \
  }
    RUN_LOAD(a0, 0);
    for (int kc = 0; kc < nkc; kc += 2) {
      RUN_LOAD(a1, kc + 1);
      RUN_MM(a0, kc);
      RUN_LOAD(a0, kc + 2);
      if (kc + 1 < nkc) RUN_MM(a1, kc + 1);
    }
#undef RUN_LOAD
#undef RUN_MM
    // lane (r, q): slot wslot + 16 g + r, channels cg * 16 NT + 16 t + 4 q .. + 3
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      int64_t row = wslot + g * 16 + r;
      bool ok = true;
      if (SCATTER) {
        const int orow = run_out[wslot + g * 16 + r];
        ok = orow >= 0;
        row = orow;
      }
      if (ok) {
        float* p = dst + row * ld_dst + cg * (16 * NT) + 4 * q;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          *reinterpret_cast<float4*>(p + 16 * t) = make_float4(acc[g][t][0], acc[g][t][1], acc[g][t][2], acc[g][t][3]);
      }
    }
  }
}

// out[i][c] = sum over the offsets with a rule for row i, ascending, of P[pos[o][i]][c].  Block = RED_ROWS output rows: the rows'
// table entries are staged and compacted in LDS, then a thread sums float4 columns with eight slab rows in flight.  Few rows per
// block: the short levels (a few thousand rows) are a chain of dependent round trips per thread, so they want many blocks.
#define RED_ROWS 16
__global__ __launch_bounds__(256) void k_run_reduce(const int* __restrict__ pos, int K, int A, const float* __restrict__ P, int cout,
                                                     float* __restrict__ out, int ld_out) {
  __shared__ int lst[27][RED_ROWS];
  __shared__ int cnt[RED_ROWS];
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * RED_ROWS;
  for (int e = tid; e < K * RED_ROWS; e += 256) {
    const int o = e / RED_ROWS, i = e % RED_ROWS;
    lst[o][i] = row0 + i < A ? pos[(int64_t)o * A + row0 + i] : -1;
  }
  __syncthreads();
  if (tid < RED_ROWS) {
    int v[27];
#pragma unroll
    for (int o = 0; o < 27; ++o) v[o] = o < K ? lst[o][tid] : -1;   // all reads in flight, then the in-place compaction
    int n = 0;
#pragma unroll
    for (int o = 0; o < 27; ++o)
      if (v[o] >= 0) lst[n++][tid] = v[o];
    cnt[tid] = n;
  }
  __syncthreads();
  const int CQ = cout >> 2;
  const float4* __restrict__ P4 = reinterpret_cast<const float4*>(P);
  for (int e = tid; e < RED_ROWS * CQ; e += 256) {
    const int i = e / CQ, c4 = e - i * CQ;
    if (row0 + i >= A) break;
    const int n = cnt[i];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < n; j += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = P4[(int64_t)lst[min(j + u, n - 1)][i] * CQ + c4];   // unconditional (the last row again)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j + u < n) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    *reinterpret_cast<float4*>(out + (int64_t)(row0 + i) * ld_out + c4 * 4) = s;
  }
}

// Shapes the offset-major path takes (the dispatcher's rule; cin / cout are those of the convolution to run).  Measured per layer
// against the output-stationary kernels at 8 and 16 scans (profiles/r5_spconv_run.md):
//   * tables with one rule per output row (no slab, no reduce): from 32 input channels and 8,192 rows;
//   * 27-offset tables: where the matrix pipe bounds the launch and the slab's bytes (2 x rules x cout x 4, written and read once)
//     stay below what the fuller MFMA groups return -- cin >= 64, cout >= 48 and rows x cout <= 110,000 x cin (wins: 103k rows
//     96 -> 48 0.78x, 49k 128 -> 64 0.71x, 98k 64 -> 64 0.97x, 39k 80 -> 160 0.86x; losses: 386k 64 -> 32 1.33x, 98k 64 -> 128 1.21x);
//   * the 8-offset tables with several rules per row (Convolution k2s2 forward, Deconvolution backward-data): within 10 % either
//     way on every level -- they stay on the output-stationary kernels.
// MOPA_SPCONV_RUN=0 switches the path off, =2 forces it wherever the shape is supported (tuning / tests).
MOPA_API int mopa_spconv_run_wanted(int32_t K, int32_t num_out, int32_t cin, int32_t cout, int32_t one_rule_per_row) {
  static const int mode = getenv("MOPA_SPCONV_RUN") ? atoi(getenv("MOPA_SPCONV_RUN")) : 1;
  const int64_t rows_per = 110000;
  if (mode == 0 || K <= 0 || K > 27 || num_out <= 0 || run_nt(cin, cout) == 0) return 0;
  if ((int64_t)num_out * 8 * 224 * 4 >= (1ll << 32)) return 0;   // 32-bit byte offsets into the input rows
  if (mode == 2) return 1;
  if (one_rule_per_row) return cin >= 32 && num_out >= 8192;   // (6,789 rows: 13.5 against 11.0 us; 19,312 rows: 9.0 against 14.9)
  if (K != 27) return 0;
  // the slab is sized from the bound K x rows slots (no host round trip), ~3 x the rules in practice: refuse layers whose bound
  // exceeds 1 GiB (the measured wins all sit below 0.8 GiB) -- they stay on the output-stationary kernels
  if (run_cap(K, num_out) * (int64_t)cout * (int64_t)sizeof(float) > (1ll << 30)) return 0;
  return cin >= 64 && cout >= 48 && (int64_t)num_out * cout <= rows_per * cin;
}

// Column-group width of the run layout for the convolution cin -> cout (0: shape not supported): what the batched weight-form
// refresh (mopa_spconv_pack_weights_batched, flag bit 16) needs in bits 8-15.
MOPA_API int mopa_spconv_run_form(int32_t cin, int32_t cout) { return run_nt(cin, cout); }

MOPA_API size_t mopa_spconv_run_workspace_bytes(int32_t K, int32_t num_out, int32_t cout) {
  if (K <= 0 || K > 27 || num_out <= 0 || cout <= 0) return 0;
  return align_up((size_t)run_cap(K, num_out) * cout * sizeof(float), 256);   // the slab, sized from the bound K * num_out slots
}

template <int NT, int RG>
static int launch_run(const int* hdr, const int* run_in, const int* run_out, int K, int64_t slots_bound, const float* in, int ld_in, int cin,
                      const float* wr, int cout, int w_flip, float* dst, int ld_dst, bool scatter, hipStream_t st) {
  const int ncu = mopa_cu_count();   // of the current device
  if (ncu <= 0) return MOPA_ERR_LAUNCH;
  const size_t lds = (size_t)cin * NT * 16 * sizeof(float);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 4) per_cu = 4;   // (2, 3, 6 per CU and 64-rule row sets per wave measured within 3 %: profiles/r5_spconv_run.md)
  if (per_cu < 1) per_cu = 1;
  const int64_t items_bound = slots_bound / (64 * RG);
  int64_t gx = (int64_t)ncu * per_cu;
  if (gx > items_bound) gx = items_bound;
  if (gx < 1) gx = 1;
  const int ncg = (cout / 16) / NT;
  dim3 grid((unsigned)gx, ncg);
#define RUN_GO(S)                                                                                                            \
  {                                                                                                                          \
    auto kern = k_spconv_run<NT, RG, S>;                                                                                     \
    if (lds > 64 * 1024 &&                                                                                                   \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return MOPA_ERR_LAUNCH;                                                                                                \
    kern<<<grid, 256, lds, st>>>(hdr, run_in, run_out, K, in, ld_in, cin, wr, ncg, w_flip, dst, ld_dst);                    \
  }
  if (scatter) RUN_GO(true) else RUN_GO(false)
#undef RUN_GO
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// Same contract as mopa_spconv_fwd on the table whose run-major rulebook is `runs` (mopa_rulebook_runs_build_batched), with the
// weight in the run layout (mopa_spconv_run_pack_weight; w_flip bit 0 = mirrored filter offsets).  one_rule_per_row: 1 = every
// output row has exactly one rule (Deconvolution k2s2, backward-data of Convolution k2s2: every fine row has one parent), 2 = at
// most one (rows without a rule are zeroed first): products go straight to `out`, no slab.  0: ws = the slab,
// mopa_spconv_run_workspace_bytes.
MOPA_API int mopa_spconv_fwd_run(const int32_t* runs, int32_t K, int32_t num_out, const float* in, int32_t ld_in, int32_t cin,
                                 const float* weight_run, int32_t cout, int32_t w_flip, float* out, int32_t ld_out,
                                 int32_t one_rule_per_row, void* ws, size_t ws_bytes, void* stream) {
  if (!runs || !in || !weight_run || !out || K <= 0 || K > 27 || num_out <= 0 || ld_in < cin || ld_out < cout) return MOPA_ERR_ARG;
  const int nt = run_nt(cin, cout);
  if (nt == 0 || ld_in % 4 || ld_out % 4 || (((uintptr_t)in | (uintptr_t)out | (uintptr_t)weight_run) & 15)) return MOPA_ERR_ARG;
  if ((int64_t)num_out * 8 * ld_in * 4 >= (1ll << 32)) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t cap = run_cap(K, num_out);
  const int* hdr = runs;
  const int* run_in = runs + RUN_HDR;
  const int* run_out = run_in + cap;
  const int* pos = run_out + cap;
  float* dst = out;
  int ld_dst = ld_out;
  if (!one_rule_per_row) {
    if (!ws || ws_bytes < mopa_spconv_run_workspace_bytes(K, num_out, cout)) return MOPA_ERR_WORKSPACE;
    dst = (float*)ws;
    ld_dst = cout;
  } else if (one_rule_per_row != 1 && hipMemset2DAsync(out, (size_t)ld_out * 4, 0, (size_t)cout * 4, (size_t)num_out, st) != hipSuccess) {
    return MOPA_ERR_LAUNCH;
  }
  // items of 128 slots (two row groups per wave) unless the table is so short that they would leave CUs idle: the rule count is
  // not known on the host (no synchronisation), ~9 rules per row on the deep 27-offset tables, rows_in <= 8 rows_out otherwise
  const int64_t rules_est = K == 27 ? (int64_t)9 * num_out : one_rule_per_row ? num_out : (int64_t)5 * num_out / 2;
  const int rg = rules_est / 128 < 1024 ? 1 : 2;   // (fewer than ~4 items of 128 per CU: level 5 at 8 scans 27 -> 17 us with 64-slot items)
  int rc;
#define RUN_L(N, G) rc = launch_run<N, G>(hdr, run_in, run_out, K, cap, in, ld_in, cin, weight_run, cout, w_flip & 1, dst, ld_dst, one_rule_per_row != 0, st)
#define RUN_N(N) if (rg == 1) RUN_L(N, 1); else RUN_L(N, 2);
  switch (nt) {
    case 1: RUN_N(1); break;
    case 2: RUN_N(2); break;
    case 3: RUN_N(3); break;
    case 4: RUN_N(4); break;
    case 5: RUN_N(5); break;
    case 6: RUN_N(6); break;
    default: RUN_N(7); break;
  }
#undef RUN_N
#undef RUN_L
  if (rc) return rc;
  if (!one_rule_per_row) {
    k_run_reduce<<<(unsigned)cdiv64(num_out, RED_ROWS), 256, 0, st>>>(pos, K, num_out, (const float*)ws, cout, out, ld_out);
    MOPA_CHECK_LAUNCH();
  }
  return MOPA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient on the run lists: dW[o] = sum over the slots s of offset o's run of in[a(s)]^T dout[b(s)] -- one GEMM per offset
// with the slots as its K dimension, (a, b) = (run_in, run_out) or, for the stride-2 convolution's gradient taken on its
// deconvolution table, (run_out, run_in).  spconv.hip's k_spconv_wgrad2 walks the dense table (27 x rows entries, one rule per 1-10
// entries), one wave per (offset, row chunk, 64-channel block of Cin) loading both operands straight into the MFMA layout: every
// output-gradient row is gathered once per Cin block and the wave's 16 MFMAs per 4 rules wait on a dependent index -> gather chain.
// Here a block of eight waves owns a PIECE of S consecutive slots of one offset (pieces are equal in rules, not in table rows: the
// centre offset of a submanifold table has 10x the rules of a corner), stages 32 slots x (Cin + Cout) floats at a time into LDS
// -- whole rows, float4 gathers, the next tile's rows in flight in registers, the tile after next's indices behind them -- and
// multiplies the full Cin x Cout tile from there (wave (w, kh): Cin blocks [w MU, w MU + MU) x all Cout blocks over half kh of the
// tile's k-steps; the halves are added through LDS at the end).  Pieces write slabs;
// k_wgrad_run_reduce adds the slabs of an offset in piece order (deterministic) into dW[o].
// Semantics: autograd of scn.SubmanifoldConvolution / Convolution / Deconvolution weights (mopa/models/scn_unet.py:27-28); oracle:
// oracle/scn3d.py::sparse_conv_bwd.
#ifndef WGR_TS
#define WGR_TS 32
#endif
#ifndef WGR_PROBE
#define WGR_PROBE 0   // timing probes (profiles/): 1 = staging only, 2 = no gathers (zero rows), 3 = neither
#endif
#define WGR_PROBE_NOCOMPUTE ((WGR_PROBE & 1) != 0)
#define WGR_PROBE_NOGATHER ((WGR_PROBE & 2) != 0)
// N consecutive floats from LDS as ONE access where the width allows (16-byte / 8-byte aligned by construction: the strides, the wave's
// base and r N are multiples of 4 / 2): scalar reads of a row-of-4 layout collide four ways on the banks, ds_read_b128 does not
template <int N>
__device__ __forceinline__ void wgr_read(const float* __restrict__ p, float (&v)[N]) {
  typedef float f32x2r __attribute__((ext_vector_type(2)));
  if constexpr (N == 4) {
    const f32x4r t = *reinterpret_cast<const f32x4r*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else if constexpr (N == 2) {
    const f32x2r t = *reinterpret_cast<const f32x2r*>(p);
    v[0] = t[0]; v[1] = t[1];
  } else if constexpr (N == 6) {
    const f32x2r t0 = *reinterpret_cast<const f32x2r*>(p), t1 = *reinterpret_cast<const f32x2r*>(p + 2), t2 = *reinterpret_cast<const f32x2r*>(p + 4);
    v[0] = t0[0]; v[1] = t0[1]; v[2] = t1[0]; v[3] = t1[1]; v[4] = t2[0]; v[5] = t2[1];
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[i];
  }
}
template <int MU, int NT, bool FULL>
__device__ __forceinline__ void wgr_tile(f32x4r (&acc)[MU][NT], const float* __restrict__ pa, const float* __restrict__ pb, int sA, int sB, int mu_w) {
#pragma unroll
  for (int kk = 0; kk < WGR_TS / 8; ++kk) {   // (the wave's half of the tile's k-steps: pa / pb point at its first slot)
    float a[MU], b[NT];
    if constexpr (FULL) {
      wgr_read<MU>(pa + kk * 4 * sA, a);
    } else {
#pragma unroll
      for (int u = 0; u < MU; ++u) a[u] = u < mu_w ? pa[kk * 4 * sA + u] : 0.f;
    }
    wgr_read<NT>(pb + kk * 4 * sB, b);
#pragma unroll
    for (int u = 0; u < MU; ++u)
      if (FULL || u < mu_w) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[t], acc[u][t], 0, 0, 0);
      }
  }
}
template <int MU, int NT>
__global__ __launch_bounds__(512) void k_wgrad_run(const int* __restrict__ hdr, const int* __restrict__ idxA, const int* __restrict__ idxB, int K,
                                                   int S, const float* __restrict__ Am, int ldA, int cin, const float* __restrict__ Bm, int ldB,
                                                   int cout, int sA, int sB, float* __restrict__ slabs) {
  constexpr int NA = (MU * WGR_TS + 31) / 32, NB = (NT * WGR_TS + 127) / 128;   // float4 gathers per thread (of 512) and tile
  extern __shared__ float4 wgr_smem4[];
  float* __restrict__ smem = reinterpret_cast<float*>(wgr_smem4);
  __shared__ int s_rng[2];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = w8 & 3, kh = w8 >> 2;   // wave (w, kh): Cin blocks [w MU, w MU + MU), the k-steps of half kh of every tile
  // piece blockIdx.x: pieces are numbered offset by offset (the K counts and starts are loaded side by side: a serial walk over the
  // header costs every block -- the empty ones of the bound-sized grid too -- K dependent L2 round trips)
  __shared__ int s_cnt[32], s_start[32];
  if (tid < 32) { s_cnt[tid] = tid < K ? hdr[32 + tid] : 0; s_start[tid] = tid < K ? hdr[tid] : 0; }
  __syncthreads();
  if (tid == 0) {
    int b = blockIdx.x, beg = 0, end = 0;
    for (int o = 0; o < K; ++o) {
      const int n = s_cnt[o], np = (n + S - 1) / S;
      if (b < np) { beg = s_start[o] + b * S; end = min(s_start[o] + n, beg + S); break; }
      b -= np;
    }
    s_rng[0] = beg; s_rng[1] = end;
  }
  __syncthreads();
  const int beg = s_rng[0], end = s_rng[1];
  if (end <= beg) return;   // (the grid is sized from the bound on the slots)
  float* __restrict__ tA = smem;                        // [2][WGR_TS][sA]
  float* __restrict__ tB = smem + 2 * WGR_TS * sA;      // [2][WGR_TS][sB]
  const int c4A = cin >> 2, c4B = cout >> 2;
  // staging map: float4 f = tid + 256 i of a tile -> (slot, float4 column), the same for every tile.  Every load and store below is
  // UNCONDITIONAL (indices clamped, values selected, the float4s past the tile's end written to a per-thread dummy slot behind the
  // tiles): a per-lane conditional load compiles to a branch around it and an s_waitcnt vmcnt(0) in front of every later use -- the
  // first version of this kernel ran its gathers, its MFMAs and its LDS stores strictly one after the other (121 us = 48 + 45 + 22
  // for the three parts alone, profiles/r5_wgrad_run.md).
  float* __restrict__ dummy = smem + 2 * WGR_TS * (sA + sB) + tid * 4;
  int slotA[NA], colA[NA], offA[NA], slotB[NB], colB[NB], offB[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = tid + 512 * i;
    const bool ok = f < WGR_TS * c4A;
    slotA[i] = ok ? f / c4A : WGR_TS + 1;   // (a slot past every piece's end: its index reads as -1, its row as zeros)
    colA[i] = ok ? (f % c4A) << 2 : 0;
    offA[i] = ok ? slotA[i] * sA + colA[i] : -1;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int f = tid + 512 * i;
    const bool ok = f < WGR_TS * c4B;
    slotB[i] = ok ? f / c4B : WGR_TS + 1;
    colB[i] = ok ? (f % c4B) << 2 : 0;
    offB[i] = ok ? slotB[i] * sB + colB[i] : -1;
  }
  int ia[2][NA], ib[2][NB];      // index registers of two tiles (the loads of one are issued IN FRONT of the row gathers that use the other's: waiting for them then leaves those gathers in flight)
  f32x4r ra[2][NA], rb[2][NB];   // two tiles of rows in flight
  unsigned vmask[2];             // which of them are rules (bit i: ra[i], bit 16 + i: rb[i])
  unsigned imask[2] = {0, 0};    // which of the index registers lie inside the piece
  const f32x4r zero4 = {0.f, 0.f, 0.f, 0.f};
  const int last = end - 1;
#define WGR_IDX(T_, IS_)   /* raw entries + which of them count: combined when the rows are issued (an OR here would wait for the loads) */ \
  {                                                                                                   \
    const int s0_ = beg + (T_) * WGR_TS;                                                              \
    unsigned m_ = 0;                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                  \
      const int s_ = s0_ + slotA[i];                                                                  \
      ia[IS_][i] = idxA[min(s_, last)];                                                               \
      m_ |= (slotA[i] <= WGR_TS && s_ < end) ? (1u << i) : 0u;                                        \
    }                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                  \
      const int s_ = s0_ + slotB[i];                                                                  \
      ib[IS_][i] = idxB[min(s_, last)];                                                               \
      m_ |= (slotB[i] <= WGR_TS && s_ < end) ? (1u << (16 + i)) : 0u;                                 \
    }                                                                                                 \
    imask[IS_] = m_;                                                                                  \
  }
#define WGR_ROWS(SET_, IS_)   /* (a padding slot's row: row 0 is loaded, dropped when the set is stored -- a select here would wait for the load) */ \
  {                                                                                                   \
    unsigned m_ = 0;                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                  \
      ra[SET_][i] = *reinterpret_cast<const f32x4r*>(Am + (int64_t)max(ia[IS_][i], 0) * ldA + colA[i]);  \
      m_ |= (ia[IS_][i] >= 0 && ((imask[IS_] >> i) & 1u) && !WGR_PROBE_NOGATHER) ? (1u << i) : 0u;              \
    }                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                  \
      rb[SET_][i] = *reinterpret_cast<const f32x4r*>(Bm + (int64_t)max(ib[IS_][i], 0) * ldB + colB[i]);  \
      m_ |= (ib[IS_][i] >= 0 && ((imask[IS_] >> (16 + i)) & 1u) && !WGR_PROBE_NOGATHER) ? (1u << (16 + i)) : 0u; \
    }                                                                                                 \
    vmask[SET_] = m_;                                                                                 \
  }
#define WGR_STORE(SET_)   /* rows of register set SET_ -> LDS buffer SET_ */                          \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                    \
      *reinterpret_cast<f32x4r*>(offA[i] >= 0 ? tA + (SET_) * WGR_TS * sA + offA[i] : dummy) = ((vmask[SET_] >> i) & 1u) ? ra[SET_][i] : zero4; \
    _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                    \
      *reinterpret_cast<f32x4r*>(offB[i] >= 0 ? tB + (SET_) * WGR_TS * sB + offB[i] : dummy) = ((vmask[SET_] >> (16 + i)) & 1u) ? rb[SET_][i] : zero4; \
  }
  const int cinb = cin >> 4;
  const int mu_w = min(MU, max(cinb - w * MU, 0));   // Cin blocks of this wave (wave-uniform)
  const int base_w = w * MU * 16;
  f32x4r acc[MU][NT];
#pragma unroll
  for (int u = 0; u < MU; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[u][t] = zero4;
  const int ntile = (end - beg + WGR_TS - 1) / WGR_TS;
  // pipeline: LDS buffer t & 1 holds tile t, register set (t + 1) & 1 the rows of tile t + 1 (in flight), the index registers tile t + 2
  WGR_IDX(0, 0);
  WGR_IDX(1, 1);   // (tiles past the piece's end: every index -1, zero rows -- issued all the same, the loop body has no branch)
  WGR_ROWS(0, 0);
  WGR_IDX(2, 0);
  WGR_ROWS(1, 1);
  WGR_STORE(0);
  __syncthreads();
#define WGR_STEP(T_, CUR_)   /* CUR_ = T_ & 1 (static: the register sets are indexed at compile time) */ \
  {                                                                                                   \
    WGR_IDX((T_) + 3, (CUR_) ^ 1);                  /* in front of the gathers: see ia */                    \
    WGR_ROWS(CUR_, CUR_);                           /* rows of tile T_ + 2 into the set tile T_ left */     \
    if (WGR_PROBE_NOCOMPUTE) {}                                                                       \
    else if (mu_w == MU)                                                                              \
      wgr_tile<MU, NT, true>(acc, tA + ((CUR_) * WGR_TS + kh * (WGR_TS / 2) + q) * sA + base_w + r * MU, tB + ((CUR_) * WGR_TS + kh * (WGR_TS / 2) + q) * sB + r * NT, sA, sB, MU); \
    else if (mu_w > 0)                                                                                \
      wgr_tile<MU, NT, false>(acc, tA + ((CUR_) * WGR_TS + kh * (WGR_TS / 2) + q) * sA + base_w + r * mu_w, tB + ((CUR_) * WGR_TS + kh * (WGR_TS / 2) + q) * sB + r * NT, sA, sB, mu_w); \
    __builtin_amdgcn_sched_barrier(0);              /* (nothing of the next step -- its address arithmetic waits for this step's index loads -- moves up here) */ \
    WGR_STORE((CUR_) ^ 1);                          /* tile T_ + 1: issued a whole tile ago */              \
    __syncthreads();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
  // lane (r, q): A[m = r][k = q] = in-row of slot 4 kk + q, channel base_w + r mu_w + u (block u); B[k = q][n = r] = its NT
  // consecutive output-gradient channels r NT + t.  A wave with all MU blocks takes the branch-free body (every LDS read of the
  // tile can be scheduled ahead of the MFMAs); the last wave of a Cin that is not a multiple of 64 MU / 4 the guarded one.
  for (int tile = 0; tile < ntile; tile += 2) {
    WGR_STEP(tile, 0);
    if (tile + 1 < ntile) WGR_STEP(tile + 1, 1);
  }
#undef WGR_STEP
#undef WGR_IDX
#undef WGR_ROWS
#undef WGR_STORE
  // the two k-halves meet in LDS (the tiles are dead: the last step ended with a barrier): waves (w, 1) write, waves (w, 0) add
  // [block (u, t)][element j][lane] floats per wave w: 16 MU NT lanes-rows of 256 bytes
  {
    float* __restrict__ cmb = smem + w * (MU * NT * 4 * 64) + lane;
    if (kh == 1) {
#pragma unroll
      for (int u = 0; u < MU; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) cmb[((u * NT + t) * 4 + j) * 64] = acc[u][t][j];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int u = 0; u < MU; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[u][t][j] += cmb[((u * NT + t) * 4 + j) * 64];
  }
  // D[m = 4 q + j][n = r] of block (u, t) = dW[channel base_w + (4 q + j) mu_w + u][r NT + t]
  float* __restrict__ sl = slabs + (int64_t)blockIdx.x * cin * cout;
#pragma unroll
  for (int u = 0; u < MU; ++u)
    if (u < mu_w) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = base_w + (4 * q + j) * mu_w + u;
#pragma unroll
        for (int t = 0; t < NT; ++t) sl[(int64_t)ci * cout + r * NT + t] = acc[u][t][j];
      }
    }
}

// dW[o][e] (+)= the slabs of offset o's pieces, in piece order: block = 64 elements (16 float4 lanes: 256-byte runs) x 16 piece-lanes
// that sum strided pieces, then a fixed-order combine (as k_reduce_slabs: deterministic)
__global__ __launch_bounds__(256) void k_wgrad_run_reduce(const int* __restrict__ hdr, int K, int S, const float* __restrict__ slabs, int n_e,
                                                          float* __restrict__ dw, int accumulate) {
  __shared__ f32x4r red[16][17];
  __shared__ int s_p[2];
  __shared__ int s_np[32];
  const int o = blockIdx.y;
  if (threadIdx.x < 32) s_np[threadIdx.x] = (int)threadIdx.x < K ? (hdr[32 + threadIdx.x] + S - 1) / S : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int p0 = 0;
    for (int k = 0; k < o; ++k) p0 += s_np[k];
    s_p[0] = p0; s_p[1] = s_np[o];
  }
  __syncthreads();
  const int p0 = s_p[0], np = s_p[1];
  const int el = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int i = (blockIdx.x * 16 + el) * 4;   // (n_e = cin cout is a multiple of 256)
  f32x4r s = {0.f, 0.f, 0.f, 0.f};
  if (i < n_e) {
    const f32x4r* __restrict__ sp = reinterpret_cast<const f32x4r*>(slabs + (int64_t)p0 * n_e + i);
    const int64_t step = (int64_t)n_e >> 2;
#pragma unroll 4
    for (int c = cl; c < np; c += 16) s += sp[c * step];
  }
  red[cl][el] = s;
  __syncthreads();
  if (cl == 0 && i < n_e) {
    float* __restrict__ d = dw + (int64_t)o * n_e + i;   // (a parameter's gradient inside a flat buffer: 4-byte aligned is all that is promised)
    f32x4r t = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) t = (f32x4r){d[0], d[1], d[2], d[3]};
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][el];
    d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3];
  }
}

static inline int wgr_stride(int c, int n16) {   // smallest row stride >= c with stride % 64 == 16 n16 % 64: the MFMA-layout reads of the four slots of a k-step fall into different banks
  const int want = (16 * n16) & 63;
  int s = (c + 3) & ~3;
  while ((s & 63) != want) s += 4;
  return s;
}
struct WgrPlan { int mu, nt, S, sA, sB; int64_t npieces; size_t lds; };
static inline bool wgr_plan(int K, int64_t num_out, int cin, int cout, int one_rule_per_row, WgrPlan* p) {
  if (K <= 0 || K > 27 || cin < 16 || cin > 256 || cin % 16 || cout < 16 || cout > 112 || cout % 16) return false;
  const int cinb = cin / 16;
  p->mu = (cinb + 3) / 4;
  p->nt = cout / 16;
  p->sA = wgr_stride(cin, p->mu);
  p->sB = wgr_stride(cout, p->nt);
  p->lds = (size_t)2 * WGR_TS * (p->sA + p->sB) * sizeof(float) + 512 * 16;   // two tiles of each operand + the dummy slots
  const size_t cmb = (size_t)4 * p->mu * p->nt * 4 * 64 * sizeof(float);         // (the k-halves' combine buffer lies over the tiles)
  if (p->lds < cmb) p->lds = cmb;
  // piece size S (slots): a piece's time grows with its steps (S / 32, strictly serial) and the kernel's fixed costs with the number of
  // pieces (prologue, slab, reduce, the tail of the last round), so the measured optimum follows sqrt(rules) -- 128 at 21 k expected rules,
  // 512-672 at 125-350 k, 1024 from 900 k on (profiles/r5_wgrad_run.md, "piece size") -- with the rule count estimated on the host (~9 per
  // row on the 27-offset tables); the 8-offset tables (one rule per row, small slabs): three pieces per CU, from 128 slots.  And no
  // more slab than 128 MiB at the BOUND on the slots.
  const int ncu = mopa_cu_count() > 0 ? mopa_cu_count() : 256;
  const int64_t rules_est = one_rule_per_row ? num_out : K == 27 ? 9 * num_out : 5 * num_out / 2;
  int64_t S = one_rule_per_row ? rules_est / (3 * ncu) : (int64_t)(1.2 * sqrt((double)rules_est));
  S = (S + WGR_TS - 1) / WGR_TS * WGR_TS;
  if (S < 128) S = 128;
  if (S > 1024) S = 1024;
  const int64_t bound = one_rule_per_row ? num_out + (int64_t)K * RUN_PAD : run_cap(K, num_out);   // slots that can hold a rule
  const int64_t per = (int64_t)cin * cout * sizeof(float);
  while ((bound / S + K) * per > (128ll << 20)) S += WGR_TS;
  p->S = (int)S;
  p->npieces = bound / S + K;
  return true;
}
// Which weight gradients take the run lists (profiles/r5_wgrad_run.md: us per launch of both kernels at 8 and 16 scans, every layer
// shape): the 27-offset layers from 48 input channels on (below, the old kernel's one-wave-per-(offset, chunk) walk is as fast: the
// GEMM is small against the table scan either way) up to 192 (224: 94 KB of LDS tiles, one block per CU) -- unless more than 30 % of
// the four Cin wave slots are empty (Cin = 80: blocks 2, 2, 1, 0) on a short level; the 8-offset tables (one rule per fine row) from
// 40,000 rows and 48 channels on.  MOPA_SPCONV_WGRAD_RUN=0: never, 2: whenever the shape is supported.
MOPA_API int mopa_spconv_wgrad_run_wanted(int32_t K, int32_t num_out, int32_t cin, int32_t cout, int32_t one_rule_per_row) {
  static const int mode = getenv("MOPA_SPCONV_WGRAD_RUN") ? atoi(getenv("MOPA_SPCONV_WGRAD_RUN")) : 1;
  WgrPlan p;
  if (!mode || !wgr_plan(K, num_out, cin, cout, one_rule_per_row, &p)) return 0;
  if (mode == 2) return 1;
  if (one_rule_per_row) return num_out >= 40000 && (cin >= 48 || cout >= 48);
  if (K != 27 || cin < 48 || cin > 192) return 0;
  const int slots = 4 * p.mu, used = cin / 16;
  return 10 * (slots - used) <= 3 * slots || num_out >= 30000;
}
MOPA_API size_t mopa_spconv_wgrad_run_workspace_bytes(int32_t K, int32_t num_out, int32_t cin, int32_t cout, int32_t one_rule_per_row) {
  WgrPlan p;
  if (!wgr_plan(K, num_out, cin, cout, one_rule_per_row, &p)) return 0;
  return align_up((size_t)p.npieces * cin * cout * sizeof(float), 256);
}

template <int MU>
static int launch_wgrad_run(const WgrPlan& p, const int* hdr, const int* ia, const int* ib, int K, const float* A, int ldA, int cin, const float* B,
                            int ldB, int cout, float* slabs, hipStream_t st) {
  typedef void (*kern_t)(const int*, const int*, const int*, int, int, const float*, int, int, const float*, int, int, int, int, float*);
  static const kern_t kerns[7] = {k_wgrad_run<MU, 1>, k_wgrad_run<MU, 2>, k_wgrad_run<MU, 3>, k_wgrad_run<MU, 4>,
                                  k_wgrad_run<MU, 5>, k_wgrad_run<MU, 6>, k_wgrad_run<MU, 7>};
  static std::atomic<bool> attr[64][7];
  const int dev = mopa_device_index();
  if (dev < 0) return MOPA_ERR_LAUNCH;
  const kern_t k = kerns[p.nt - 1];
  if (p.lds > 64 * 1024 && !attr[dev][p.nt - 1].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
      return MOPA_ERR_LAUNCH;
    attr[dev][p.nt - 1].store(true, std::memory_order_release);
  }
  k<<<(unsigned)p.npieces, 512, p.lds, st>>>(hdr, ia, ib, K, p.S, A, ldA, cin, B, ldB, cout, p.sA, p.sB, slabs);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// dweight[K][cin][cout] (+= if accumulate) from in[*, cin] and dout[*, cout] over the rules of the table the run-major rulebook `runs` was
// built from (num_out = its output rows).  swap == 0: `in` rows are the rules' input rows and `dout` rows their output rows; swap == 1
// (the stride-2 convolution's gradient on its deconvolution table: one_rule_per_row): the other way round.  ws >=
// mopa_spconv_wgrad_run_workspace_bytes.  Rows 16-byte aligned (ld % 4 == 0).
MOPA_API int mopa_spconv_bwd_weight_run(const int32_t* runs, int32_t K, int32_t num_out, int32_t one_rule_per_row, int32_t swap, const float* in,
                                        int32_t ld_in, int32_t cin, const float* dout, int32_t ld_dout, int32_t cout, float* dweight,
                                        int32_t accumulate, void* ws, size_t ws_bytes, void* stream) {
  WgrPlan p;
  if (!runs || !in || !dout || !dweight || num_out <= 0 || !wgr_plan(K, num_out, cin, cout, one_rule_per_row, &p)) return MOPA_ERR_ARG;
  if (ld_in < cin || ld_dout < cout || (ld_in & 3) || (ld_dout & 3) || (((uintptr_t)in | (uintptr_t)dout | (uintptr_t)ws) & 15)) return MOPA_ERR_ARG;
  if (!ws || ws_bytes < mopa_spconv_wgrad_run_workspace_bytes(K, num_out, cin, cout, one_rule_per_row)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t cap = run_cap(K, num_out);
  const int* hdr = runs;
  const int* run_in = runs + RUN_HDR;
  const int* run_out = run_in + cap;
  const int* ia = swap ? run_out : run_in;
  const int* ib = swap ? run_in : run_out;
  float* slabs = (float*)ws;
  int rc;
  switch (p.mu) {
    case 1: rc = launch_wgrad_run<1>(p, hdr, ia, ib, K, in, ld_in, cin, dout, ld_dout, cout, slabs, st); break;
    case 2: rc = launch_wgrad_run<2>(p, hdr, ia, ib, K, in, ld_in, cin, dout, ld_dout, cout, slabs, st); break;
    case 3: rc = launch_wgrad_run<3>(p, hdr, ia, ib, K, in, ld_in, cin, dout, ld_dout, cout, slabs, st); break;
    default: rc = launch_wgrad_run<4>(p, hdr, ia, ib, K, in, ld_in, cin, dout, ld_dout, cout, slabs, st); break;
  }
  if (rc) return rc;
  const int n_e = cin * cout;
  k_wgrad_run_reduce<<<dim3((unsigned)cdiv64(n_e, 64), (unsigned)K), 256, 0, st>>>(hdr, K, p.S, slabs, n_e, dweight, accumulate);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
