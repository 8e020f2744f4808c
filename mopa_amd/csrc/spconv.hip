// Sparse 3D convolution on rule tables: forward / backward-data share one output-stationary kernel,
// backward-weight is a pair-reduction kernel.  f32-in/f32-acc MFMA (v_mfma_f32_16x16x4_f32) for the
// per-rule GEMM -- exact fp32 (k-ordered fmaf chain), no reduced precision anywhere.
//
// Replaces sparseconvnet's per-filter-offset gather-GEMM-scatter launches reached from
// mopa/models/scn_unet.py:27-28 (SubmanifoldConvolution, scn.UNet Convolution/Deconvolution);
// semantics: SURVEY.md Appendix A.4/A.5, oracle: oracle/scn3d.py::sparse_conv.
//
// Rule table: nbr[K][A_out] int32, nbr[o][i] = input row feeding output row i through filter offset o, or -1.
//   out[i] = sum_o in[nbr[o][i]] @ W[o]          (offsets applied in increasing o: fixed summation order)
// SubM 3^3 (K=27), Conv k2s2 (K=8, table ch) and Deconv k2s2 (K=8, table up) are the same call.
// Backward-data is the same call on the transposed table (SubM: nbr itself with W[26-o]^T; Conv<->Deconv swap).
//
// Data layout: features row-major [rows][ld] fp32 (ld >= C lets a layer read/write a channel slice of a wider
// JoinTable buffer); weights [K][Cin][Cout] fp32.
#include "common.h"
#include "sprun_pack.h"
#ifdef MOPA_EXP_RING   // round-4 experiment (profiles/experiments/spconv_ring.hip, build_ring.sh): not in the shipped library
#include "spconv_ring.h"
#endif
#include <stdlib.h>
#include <string.h>
#include <atomic>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------------
// Forward / backward-data.  One wave per (tile of TMR output rows, group of 16*NTW output columns).  The wave first
// stages its slice of the rule table (K x TMR int32) in LDS with all loads in flight at once, then per filter offset:
// ballot-compacts the valid rules, gathers 16 input rows at a time straight into the MFMA A-operand layout with
// 16-byte loads, multiplies by its column slice of W[o] and adds the 16 result rows into the tile's LDS accumulator
// (within one offset every output row occurs at most once, so no atomics).  Each output element is written to HBM
// exactly once.  Splitting the columns over grid.y keeps the deep, short levels (a few thousand rows x 96-192
// channels) from serialising 27 x Cin/4 x Cout/16 MFMAs in a handful of waves; the row gathers are then repeated per
// column group but come from L2.
//
// Operand mapping (k and n are permuted consistently so every global load is contiguous per lane):
//   lane l: r = l&15, q = l>>4.   A[i=r][k] = in[pair r][16*kk + 4*q + s]          (one float4 per kk)
//                                 B[k][j=r] = W[o][16*kk + 4*q + s][c0 + r*NTW + t]  (NTW contiguous floats)
//   D tile t: lane holds rows 4*q+j (j<4), i.e. pairs, column c0 + r*NTW + t.
template <int NTW, int NA, bool ALIGNED>
__global__ __launch_bounds__(64) void k_spconv_fwd(const int* __restrict__ nbr, int K, int A_out,
                                                    const float* __restrict__ in, int ld_in, int cin,
                                                    const float* __restrict__ W, int cout, int w_flip,
                                                    float* __restrict__ out, int ld_out) {
  constexpr int TMR = 64;
  constexpr int CP = NTW * 16;     // columns of this group
  constexpr int LD = CP + 4;       // LDS row stride (floats); +4 keeps 16-B alignment, breaks pow2 strides
  __shared__ __attribute__((aligned(16))) float acc[TMR * LD];
  __shared__ int l_in[2][TMR];     // compacted rules of the current / next offset (double-buffered)
  __shared__ int l_out[2][TMR];
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * TMR;
  const int c0 = blockIdx.y * CP;
  const int cin16 = (cin + 15) >> 4;
  const int row = row0 + lane;
  const int rowc = row < A_out ? row : A_out - 1;

  for (int i = lane; i < TMR * LD; i += 64) acc[i] = 0.f;

  // Software pipeline over the filter offsets.  While offset o is multiplied, (1) the rule-table entries of offset
  // o+2 stream in, (2) offset o+1 is ballot-compacted into the other LDS list, (3) the rows of the next 16-rule group
  // (of this offset, or the first of offset o+1) are gathered into registers -- so each step of the per-wave chain
  // costs MFMA + LDS-accumulate time, not an L2/HBM round trip.  All loads are unconditional (clamped + masked).
  auto gather = [&](int irow, float4* a) {
    const float* ar = in + (int64_t)(irow < 0 ? 0 : irow) * ld_in;
#pragma unroll
    for (int kk = 0; kk < NA; ++kk) {
      const int kb = kk * 16 + q * 4;
      if (ALIGNED) {
        const float4 v = *reinterpret_cast<const float4*>(ar + (kk < cin16 ? kb : q * 4));
        const bool ok = irow >= 0 && kk < cin16;
        a[kk] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
      } else {
        a[kk].x = (irow >= 0 && kb + 0 < cin) ? ar[kb + 0] : 0.f;
        a[kk].y = (irow >= 0 && kb + 1 < cin) ? ar[kb + 1] : 0.f;
        a[kk].z = (irow >= 0 && kb + 2 < cin) ? ar[kb + 2] : 0.f;
        a[kk].w = (irow >= 0 && kb + 3 < cin) ? ar[kb + 3] : 0.f;
      }
    }
  };
  auto compact = [&](int nb, int buf) -> int {
    const unsigned long long bal = __ballot(nb >= 0);
    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    if (nb >= 0) { l_in[buf][pos] = nb; l_out[buf][pos] = lane; }
    return __popcll(bal);
  };
  auto table = [&](int o) -> int {   // entry of offset o for this lane's row (clamped, masked)
    const int oc = o < K ? o : K - 1;
    const int v = nbr[(int64_t)oc * A_out + rowc];
    return (o < K && row < A_out) ? v : -1;
  };

  int nb1 = table(1);                       // offset 1 in flight
  int n_cur = compact(table(0), 0);         // offset 0 compacted
  __syncthreads();
  float4 a_c[NA], a_n[NA];
  gather(n_cur > r ? l_in[0][r] : -1, a_n); // first group of offset 0
  for (int o = 0; o < K; ++o) {
    const int cur = o & 1;
    const int nb2 = table(o + 2);
    const int n_next = compact(nb1, cur ^ 1);   // safe: list cur^1 was last read two offsets ago (barrier below)
    nb1 = nb2;
    __syncthreads();
    const float* __restrict__ wo = W + (int64_t)(w_flip ? K - 1 - o : o) * cin * cout + c0 + r * NTW;
    for (int g0 = 0; g0 < n_cur; g0 += 16) {
#pragma unroll
      for (int kk = 0; kk < NA; ++kk) a_c[kk] = a_n[kk];
      // prefetch: next group of this offset, or the first group of the next offset
      {
        const int p = g0 + 16 + r;
        int irow;
        if (g0 + 16 < n_cur) irow = p < n_cur ? l_in[cur][p] : -1;
        else irow = r < n_next ? l_in[cur ^ 1][r] : -1;
        gather(irow, a_n);
      }
      f32x4 d[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NA; ++kk) {
        if (NA <= 4 || kk < cin16) {
          const float av[4] = {a_c[kk].x, a_c[kk].y, a_c[kk].z, a_c[kk].w};
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const int k = kk * 16 + q * 4 + s2;
            float bw[NTW];
            const float* __restrict__ wr = wo + (int64_t)(ALIGNED ? k : (k < cin ? k : 0)) * cout;
#pragma unroll
            for (int t = 0; t < NTW; ++t) bw[t] = (ALIGNED || (k < cin && c0 + r * NTW + t < cout)) ? wr[t] : 0.f;
#pragma unroll
            for (int t = 0; t < NTW; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bw[t], d[t], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pr = g0 + q * 4 + j;
        if (pr < n_cur) {
          float* ap = acc + l_out[cur][pr] * LD + r * NTW;
#pragma unroll
          for (int t = 0; t < NTW; ++t) ap[t] += d[t][j];
        }
      }
    }
    if (n_cur == 0) {  // nothing consumed a_n's slot: (re)issue the first group of the next offset
      gather(r < n_next ? l_in[cur ^ 1][r] : -1, a_n);
    }
    n_cur = n_next;
  }
  __syncthreads();
  // write the tile's column group: each output element exactly once
  if (ALIGNED) {
    constexpr int V = CP / 4;  // float4 per row of this group
    for (int i = lane; i < TMR * V; i += 64) {
      const int rr = i / V, c4 = i - rr * V;
      if (row0 + rr < A_out)
        *reinterpret_cast<float4*>(out + (int64_t)(row0 + rr) * ld_out + c0 + c4 * 4) =
            *reinterpret_cast<const float4*>(acc + rr * LD + c4 * 4);
    }
  } else {
    for (int i = lane; i < TMR * CP; i += 64) {
      const int rr = i / CP, c = i - rr * CP;
      if (row0 + rr < A_out && c0 + c < cout) out[(int64_t)(row0 + rr) * ld_out + c0 + c] = acc[rr * LD + c];
    }
  }
}

// Column-group width (in 16-column tiles) and row-tile height.  NTW must divide NT (vector B loads); columns are
// split further whenever a launch would otherwise hold fewer than ~2k waves.  128-row tiles only where a 64-row tile
// sees < 16 rules per offset on average (the sparse shallow levels), so MFMA row groups are better filled.
static void fwd_plan(int K, int A_out, int cout, int* ntw, int* tmr) {
  const int NT = (cout + 15) / 16;
  // measured (profiles/r1_*): tiles >= ~1500 already fill the chip with one wave per tile and all columns (the row
  // gathers are not repeated); below that the per-wave serial chain over the offsets dominates and the columns are
  // split across waves.
  const int64_t tiles = cdiv64(A_out, 64);
  int best = 1;
  for (int c = 4; c >= 1; --c)
    if (NT % c == 0 && (tiles >= 1500 || tiles * (NT / c) >= 1024)) { best = c; break; }
  *ntw = best;
  *tmr = 64;
}

template <int NTW, int NA>
static int launch_fwd(const int* nbr, int K, int A_out, const float* in, int ld_in, int cin, const float* W,
                      int cout, int w_flip, float* out, int ld_out, bool aligned, hipStream_t st) {
  const int NT = (cout + 15) / 16;
  dim3 grid((A_out + 63) / 64, (NT + NTW - 1) / NTW);
  if (aligned)
    k_spconv_fwd<NTW, NA, true><<<grid, 64, 0, st>>>(nbr, K, A_out, in, ld_in, cin, W, cout, w_flip, out, ld_out);
  else
    k_spconv_fwd<NTW, NA, false><<<grid, 64, 0, st>>>(nbr, K, A_out, in, ld_in, cin, W, cout, w_flip, out, ld_out);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

template <int NTW>
static int dispatch_fwd(const int* nbr, int K, int A_out, const float* in, int ld_in, int cin, const float* W, int cout,
                        int w_flip, float* out, int ld_out, bool aligned, hipStream_t st) {
#define FW(C) return launch_fwd<NTW, C>(nbr, K, A_out, in, ld_in, cin, W, cout, w_flip, out, ld_out, aligned, st)
  switch ((cin + 15) / 16) {
    case 1: FW(1);
    case 2: FW(2);
    case 3: FW(3);
    case 4: FW(4);
    default: break;
  }
  if (cin <= 128) FW(8);
  FW(12);
#undef FW
}

// ----------------------------------------------------------------------------------------------
// The stem: SubmanifoldConvolution(in_channels -> m) on the raw voxel features (mopa/models/scn_unet.py:27: 1 -> 16).  With one to
// four input channels there is no GEMM: out[i][c] = sum_o sum_ci in[nbr[o][i]][ci] * W[o][ci][c] is 27 scalar gathers per row.  One
// thread per output row keeps the row's COUT accumulators in registers, the table entries are coalesced loads, the weights are
// wave-uniform (scalar loads).  Same order as the table kernel (offsets ascending, input channels ascending): the same bits for
// one input channel.  The MFMA kernels spend 50 us per launch here (16-wide K padding, 16 columns per wave) on 3 MB of work.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_spconv_stem(const int* __restrict__ nbr, int K, int A_out, const float* __restrict__ in, int ld_in,
                                                      const float* __restrict__ W, int w_flip, float* __restrict__ out, int ld_out) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= A_out) return;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
  for (int o0 = 0; o0 < K; o0 += 9) {   // nine table entries (and their gathers) in flight at a time
    int idx[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) idx[u] = o0 + u < K ? nbr[(int64_t)(o0 + u) * A_out + row] : -1;
    float x[9][CIN];
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) x[u][ci] = idx[u] >= 0 ? in[(int64_t)idx[u] * ld_in + ci] : 0.f;
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      if (o0 + u >= K) break;
      const float* __restrict__ wo = W + (int64_t)(w_flip ? K - 1 - (o0 + u) : o0 + u) * CIN * COUT;   // wave-uniform
      if (idx[u] >= 0) {   // the arithmetic of the table kernel: the offset's product (rounded: an MFMA chain from zero), then one add
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
          float d = __fmul_rn(x[u][0], wo[c]);
#pragma unroll
          for (int ci = 1; ci < CIN; ++ci) d = fmaf(x[u][ci], wo[ci * COUT + c], d);
          acc[c] = __fadd_rn(acc[c], d);
        }
      }
    }
  }
  float* __restrict__ op = out + (int64_t)row * ld_out;
#pragma unroll
  for (int c = 0; c < COUT; c += 4) *reinterpret_cast<float4*>(op + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
}

template <int CIN>
static int launch_stem(const int* nbr, int K, int A_out, const float* in, int ld_in, const float* W, int cout, int w_flip, float* out,
                       int ld_out, hipStream_t st) {
  const unsigned nb = (unsigned)cdiv64(A_out, 256);
  if (cout == 16) k_spconv_stem<CIN, 16><<<nb, 256, 0, st>>>(nbr, K, A_out, in, ld_in, W, w_flip, out, ld_out);
  else k_spconv_stem<CIN, 32><<<nb, 256, 0, st>>>(nbr, K, A_out, in, ld_in, W, w_flip, out, ld_out);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// out[A_out][cout] (row stride ld_out) = sum_o in[nbr[o][.]] @ W[w_flip ? K-1-o : o]
MOPA_API int mopa_spconv_fwd(const int32_t* nbr, int32_t K, int32_t num_out, const float* in, int32_t ld_in,
                             int32_t cin, const float* weight, int32_t cout, int32_t w_flip, float* out,
                             int32_t ld_out, void* stream) {
  if (K <= 0 || K > 27 || num_out <= 0 || cin <= 0 || cin > 192 || cout <= 0 || ld_in < cin || ld_out < cout) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  static const int stem_on = getenv("MOPA_SPCONV_STEM") ? atoi(getenv("MOPA_SPCONV_STEM")) : 1;   // A/B switch
  if (stem_on && cin <= 4 && (cout == 16 || cout == 32) && ld_out % 4 == 0 && ((uintptr_t)out & 15) == 0) {   // the stem: scalar gathers
    switch (cin) {
      case 1: return launch_stem<1>(nbr, K, num_out, in, ld_in, weight, cout, w_flip, out, ld_out, st);
      case 2: return launch_stem<2>(nbr, K, num_out, in, ld_in, weight, cout, w_flip, out, ld_out, st);
      case 3: return launch_stem<3>(nbr, K, num_out, in, ld_in, weight, cout, w_flip, out, ld_out, st);
      default: return launch_stem<4>(nbr, K, num_out, in, ld_in, weight, cout, w_flip, out, ld_out, st);
    }
  }
  const bool aligned = (cin % 16 == 0) && (cout % 16 == 0) && (ld_in % 4 == 0) && (ld_out % 4 == 0) &&
                       (((uintptr_t)in | (uintptr_t)out | (uintptr_t)weight) % 16 == 0);
  int ntw, tmr;
  fwd_plan(K, num_out, cout, &ntw, &tmr);
  if (!aligned) ntw = 1;
  switch (ntw) {
    case 1: return dispatch_fwd<1>(nbr, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, aligned, st);
    case 2: return dispatch_fwd<2>(nbr, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, aligned, st);
    case 3: return dispatch_fwd<3>(nbr, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, aligned, st);
    default: return dispatch_fwd<4>(nbr, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, aligned, st);
  }
}

// ----------------------------------------------------------------------------------------------
// Grouped rulebook: the rule table compacted ONCE per geometry into MFMA-ready groups, so that the convolution
// kernel has no ballot / LDS-list / barrier chain per offset and can prefetch its gathers.
//   tile t (64 consecutive output rows) owns groups grp_start[t] .. grp_start[t+1]-1, ordered by filter offset;
//   group g: grp_o[g] = filter offset, grp_in[g][16] = input rows (-1 = padding), grp_out[g][16] = output row within
//   the tile (0..63, -1 = padding).  Every (offset, tile) with n valid rules contributes ceil(n/16) groups.
__global__ __launch_bounds__(64) void k_rb_count(const int* __restrict__ nbr, int K, int A_out, int* __restrict__ tile_groups) {
  const int row = blockIdx.x * 64 + threadIdx.x;
  int ng = 0;
  for (int o = 0; o < K; ++o) {
    const int nb = (row < A_out) ? nbr[(int64_t)o * A_out + row] : -1;
    ng += (__popcll(__ballot(nb >= 0)) + 15) >> 4;
  }
  if (threadIdx.x == 0) tile_groups[blockIdx.x] = ng;
}

__global__ __launch_bounds__(64) void k_rb_fill(const int* __restrict__ nbr, int K, int A_out, const int* __restrict__ grp_start,
                                                 int* __restrict__ grp_o, int* __restrict__ grp_in, int* __restrict__ grp_out) {
  const int lane = threadIdx.x, row = blockIdx.x * 64 + lane;
  int g = grp_start[blockIdx.x];
  for (int o = 0; o < K; ++o) {
    const int nb = (row < A_out) ? nbr[(int64_t)o * A_out + row] : -1;
    const unsigned long long bal = __ballot(nb >= 0);
    const int n = __popcll(bal);
    if (n == 0) continue;
    const int ng = (n + 15) >> 4;
    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    if (nb >= 0) { grp_in[(int64_t)g * 16 + pos] = nb; grp_out[(int64_t)g * 16 + pos] = lane; }
    if (lane < ng * 16 - n) { grp_in[(int64_t)g * 16 + n + lane] = -1; grp_out[(int64_t)g * 16 + n + lane] = -1; }
    if (lane < ng) grp_o[g + lane] = o;
    g += ng;
  }
}

MOPA_API int mopa_rulebook_groups_count(const int32_t* nbr, int32_t K, int32_t num_out, int32_t* tile_groups, void* stream) {
  if (K <= 0 || num_out <= 0) return MOPA_ERR_ARG;
  k_rb_count<<<(num_out + 63) / 64, 64, 0, (hipStream_t)stream>>>(nbr, K, num_out, tile_groups);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_rulebook_groups_fill(const int32_t* nbr, int32_t K, int32_t num_out, const int32_t* grp_start, int32_t* grp_o,
                                       int32_t* grp_in, int32_t* grp_out, void* stream) {
  if (K <= 0 || num_out <= 0) return MOPA_ERR_ARG;
  k_rb_fill<<<(num_out + 63) / 64, 64, 0, (hipStream_t)stream>>>(nbr, K, num_out, grp_start, grp_o, grp_in, grp_out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ----------------------------------------------------------------------------------------------
// Weight packing for the pipelined kernels, in column groups of NTW 16-column tiles:
//   wp[cg][o][kc][lane][s2][t] = Wc[o][16*kc + 4*(lane>>4) + s2][cg*16*NTW + (lane&15)*NTW + t]  where Wc
// is the [K][cin_c][cout_c] weight of the convolution to run: w itself, or (transpose) w[o]^T for backward-data.
__global__ void k_pack_w(const float* __restrict__ w, int K, int cin_w, int cout_w, int transpose, int ntw, float* __restrict__ wp) {
  const int cin_c = transpose ? cout_w : cin_w, cout_c = transpose ? cin_w : cout_w;
  const int nkc = cin_c >> 4, cp = ntw * 16;
  const int n = K * cin_c * cout_c;   // < 2^31
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int rem = i;
    const int t = rem % ntw; rem /= ntw;
    const int s2 = rem & 3; rem >>= 2;
    const int lane = rem & 63; rem >>= 6;
    const int kc = rem % nkc; rem /= nkc;
    const int o = rem % K;
    const int cg = rem / K;
    const int k = kc * 16 + (lane >> 4) * 4 + s2, c = cg * cp + (lane & 15) * ntw + t;
    wp[i] = transpose ? w[((int64_t)o * cin_w + c) * cout_w + k] : w[((int64_t)o * cin_w + k) * cout_w + c];
  }
}

// All rule tables of one geometry in ONE launch each (20 tables for a 7-level UNet: 27-offset tables of every level and
// the down / up tables between levels): desc[t] = {table pointer, K, rows, first global tile} (int64 x 4, a HOST array that
// travels as a kernel argument), tiles and groups are numbered globally, so every table's grp_start is a slice of one scan and its group arrays are
// the shared ones.  Saves ~55 launches per geometry build (the 3D-only step is host-bound).
#define RB_MAX_TABLES 32
struct RbDescs { int64_t v[RB_MAX_TABLES * 4]; };
__device__ __forceinline__ int rb_find_table(const RbDescs& d, int ntables, int tile) {
  int t = 0;
  for (int k = 1; k < ntables; ++k)
    if (tile >= (int)d.v[k * 4 + 3]) t = k;
  return t;
}
__global__ __launch_bounds__(64) void k_rb_count_batched(const RbDescs desc, int ntables, int* __restrict__ tile_groups) {
  const int t = rb_find_table(desc, ntables, blockIdx.x);
  const int* __restrict__ nbr = reinterpret_cast<const int*>(desc.v[t * 4]);
  const int K = (int)desc.v[t * 4 + 1], A_out = (int)desc.v[t * 4 + 2];
  const int row = (blockIdx.x - (int)desc.v[t * 4 + 3]) * 64 + threadIdx.x;
  int ng = 0;
  for (int o = 0; o < K; ++o) {
    const int nb = (row < A_out) ? nbr[(int64_t)o * A_out + row] : -1;
    ng += (__popcll(__ballot(nb >= 0)) + 15) >> 4;
  }
  if (threadIdx.x == 0) tile_groups[blockIdx.x] = ng;
}
__global__ __launch_bounds__(64) void k_rb_fill_batched(const RbDescs desc, int ntables, const int* __restrict__ grp_start,
                                                         int* __restrict__ grp_o, int* __restrict__ grp_in, int* __restrict__ grp_out) {
  const int t = rb_find_table(desc, ntables, blockIdx.x);
  const int* __restrict__ nbr = reinterpret_cast<const int*>(desc.v[t * 4]);
  const int K = (int)desc.v[t * 4 + 1], A_out = (int)desc.v[t * 4 + 2];
  const int lane = threadIdx.x, row = (blockIdx.x - (int)desc.v[t * 4 + 3]) * 64 + lane;
  int g = grp_start[blockIdx.x];
  for (int o = 0; o < K; ++o) {
    const int nb = (row < A_out) ? nbr[(int64_t)o * A_out + row] : -1;
    const unsigned long long bal = __ballot(nb >= 0);
    const int n = __popcll(bal);
    if (n == 0) continue;
    const int ng = (n + 15) >> 4;
    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    if (nb >= 0) { grp_in[(int64_t)g * 16 + pos] = nb; grp_out[(int64_t)g * 16 + pos] = lane; }
    if (lane < ng * 16 - n) { grp_in[(int64_t)g * 16 + n + lane] = -1; grp_out[(int64_t)g * 16 + n + lane] = -1; }
    if (lane < ng) grp_o[g + lane] = o;
    g += ng;
  }
}
MOPA_API int mopa_rulebook_groups_count_batched(const int64_t* desc_host, int32_t ntables, int32_t total_tiles, int32_t* tile_groups,
                                                void* stream) {
  if (ntables <= 0 || ntables > RB_MAX_TABLES || total_tiles <= 0) return MOPA_ERR_ARG;
  RbDescs desc;
  memcpy(desc.v, desc_host, (size_t)ntables * 4 * sizeof(int64_t));
  k_rb_count_batched<<<total_tiles, 64, 0, (hipStream_t)stream>>>(desc, ntables, tile_groups);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
// grp_start: exclusive scan of tile_groups over ALL tiles (global group numbers); grp_o / grp_in / grp_out: shared arrays.
MOPA_API int mopa_rulebook_groups_fill_batched(const int64_t* desc_host, int32_t ntables, int32_t total_tiles, const int32_t* grp_start,
                                               int32_t* grp_o, int32_t* grp_in, int32_t* grp_out, void* stream) {
  if (ntables <= 0 || ntables > RB_MAX_TABLES || total_tiles <= 0) return MOPA_ERR_ARG;
  RbDescs desc;
  memcpy(desc.v, desc_host, (size_t)ntables * 4 * sizeof(int64_t));
  k_rb_fill_batched<<<total_tiles, 64, 0, (hipStream_t)stream>>>(desc, ntables, grp_start, grp_o, grp_in, grp_out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// Which kernel mopa_spconv_fwd_grouped runs for a shape (cin / cout are those of the convolution to run, i.e. swapped
// for backward-data) and the column-group width NTW its packed weights need (w_flip bit 1).  0 = unpacked weights.
enum { SP_BLK = 0, SP_PIPE = 1, SP_T4 = 2, SP_RING = 3 };
static int packed_plan(int K, int64_t num_out, int cin, int cout, int* ntw) {
  static const int force_path = getenv("MOPA_SPCONV_PATH") ? atoi(getenv("MOPA_SPCONV_PATH")) : 0;  // tuning only
  *ntw = 0;
  if (cin % 16 || cout % 16 || cin > 224 || cout > 224 || force_path == 2) return SP_BLK;
  const int64_t tiles = cdiv64(num_out, 64);
  const int nt = cout / 16;
#ifdef MOPA_EXP_RING
  if (force_path == 0) {   // persistent ring kernel, MOPA_SPCONV_RING=1|2
    const int rw = mopa_ring_plan(K, num_out, cin, cout);
    if (rw > 0) { *ntw = rw; return SP_RING; }
  }
#endif
  // 8-offset down/up tables: few groups per tile -> one wave per tile on the long levels (dispatch below), four on the short
  if (K != 27 && force_path != 4 && force_path != 0) return SP_BLK;
  if (force_path == 1) {
    if (cout > 64) return SP_BLK;
    *ntw = nt;
    return SP_PIPE;
  }
  // measured (profiles/bench_spconv.py, us per launch, dense-table / block / pipe / t4):  L0 16->16 46/47/30/45,
  // L1 32->32 80/100/75/66, L2 96->48 216/240/168/117, L3 64->64 93/110/108/64, L3 128->64 229/199/207/116,
  // L4 160->80 632/169/-/122, L5 96->96 115/52/-/37.
  if (force_path == 0 && cout == 16 && tiles >= 1500) {  // 16-channel units are too thin to split four ways
    *ntw = 1;
    return SP_PIPE;
  }
  // 4 waves per (tile, column group): column groups of 2 tiles when they divide Cout, else 3, else 1 -- and 1 on the
  // shortest levels, where the wider groups would leave fewer than ~3 blocks per CU
  int w = nt % 2 == 0 ? 2 : nt % 3 == 0 ? 3 : 1;
  if (tiles * (nt / w) < 768) w = 1;   // (level 5 with 32- / 48-column groups instead: 35.3 -> 34.5 / 51.8 us -- not kept)
  if (force_path == 4 || tiles * (nt / w) >= 256) {
    *ntw = w;
    return SP_T4;
  }
  return SP_BLK;
}
MOPA_API int mopa_spconv_grouped_wants_packed(int32_t K, int32_t num_out, int32_t cin, int32_t cout) {
  int ntw;
  packed_plan(K, num_out, cin, cout, &ntw);
  return ntw;
}

// wp (K*cin*cout floats) = packed form, for column groups of `ntw` 16-column tiles (mopa_spconv_grouped_wants_packed),
// of w [K][cin][cout] (transpose = 0) or of its per-offset transpose (1).
MOPA_API int mopa_spconv_pack_weight(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t transpose, int32_t ntw,
                                     float* wp, void* stream) {
  const int cin_c = transpose ? cout : cin, cout_c = transpose ? cin : cout;
  if (K <= 0 || cin_c <= 0 || cout_c <= 0 || cin_c % 16 || cout_c % 16 || ntw < 1 || ntw > 4 || (cout_c / 16) % ntw) return MOPA_ERR_ARG;
  const int64_t n = (int64_t)K * cin * cout;
  if (n >= (1ll << 31)) return MOPA_ERR_ARG;
  k_pack_w<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(w, K, cin, cout, transpose, ntw, wp);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// Every packed / transposed weight form of a network in ONE launch (the 3D step re-lays out ~50 conv weights per optimizer step:
// 50 launches at the ~5 us floor each, on a host-paced step).  desc_host [n][6] int64: source, destination, K, cin, cout (of the
// layer weight [K][cin][cout]), flags: bit 0 = the convolution to run is the per-offset transpose (backward-data), bits 8-15 =
// ntw (column groups of ntw 16-column tiles, as mopa_spconv_pack_weight) or 0 = plain per-offset transpose
// (mopa_spconv_transpose_weight); bit 16 = the run layout of the offset-major kernel (mopa_spconv_run_pack_weight; bits 8-15 then
// hold ITS column-group width, mopa_spconv_run_form).  n <= 64 per call; the array travels as a kernel argument.
#define PW_MAX 64
struct PackDescs { int64_t src[PW_MAX], dst[PW_MAX]; int32_t K[PW_MAX], cin[PW_MAX], cout[PW_MAX], flags[PW_MAX]; };
__global__ void k_pack_w_batched(const PackDescs d) {
  const int e = blockIdx.y;
  const float* __restrict__ w = reinterpret_cast<const float*>(d.src[e]);
  float* __restrict__ wp = reinterpret_cast<float*>(d.dst[e]);
  const int K = d.K[e], cin_w = d.cin[e], cout_w = d.cout[e], transpose = d.flags[e] & 1, ntw = (d.flags[e] >> 8) & 0xff;
  const int n = K * cin_w * cout_w;
  if (d.flags[e] & 0x10000) {   // the run layout of sprun.hip (bits 8-15: its column-group width)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) run_pack_elem(w, wp, i, K, cin_w, cout_w, transpose, ntw);
    return;
  }
  if (ntw == 0) {   // wt[o][co][ci] = w[o][ci][co]
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const int ci = i % cin_w, t = i / cin_w;
      const int co = t % cout_w, o = t / cout_w;
      wp[i] = w[((int64_t)o * cin_w + ci) * cout_w + co];
    }
    return;
  }
  const int cin_c = transpose ? cout_w : cin_w;
  const int nkc = cin_c >> 4, cp = ntw * 16;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {   // same map as k_pack_w
    int rem = i;
    const int t = rem % ntw; rem /= ntw;
    const int s2 = rem & 3; rem >>= 2;
    const int lane = rem & 63; rem >>= 6;
    const int kc = rem % nkc; rem /= nkc;
    const int o = rem % K;
    const int cg = rem / K;
    const int k = kc * 16 + (lane >> 4) * 4 + s2, c = cg * cp + (lane & 15) * ntw + t;
    wp[i] = transpose ? w[((int64_t)o * cin_w + c) * cout_w + k] : w[((int64_t)o * cin_w + k) * cout_w + c];
  }
}
MOPA_API int mopa_spconv_pack_weights_batched(const int64_t* desc_host, int32_t n, void* stream) {
  if (!desc_host || n <= 0 || n > PW_MAX) return MOPA_ERR_ARG;
  PackDescs d;
  memset(&d, 0, sizeof(d));
  int64_t nmax = 0;
  for (int e = 0; e < n; ++e) {
    const int64_t* r = desc_host + (int64_t)e * 6;
    const int K = (int)r[2], cin = (int)r[3], cout = (int)r[4], flags = (int)r[5];
    const int transpose = flags & 1, ntw = (flags >> 8) & 0xff;
    const int cin_c = transpose ? cout : cin, cout_c = transpose ? cin : cout;
    if (!r[0] || !r[1] || K <= 0 || cin <= 0 || cout <= 0 || (int64_t)K * cin * cout >= (1ll << 31)) return MOPA_ERR_ARG;
    const bool run = (flags & 0x10000) != 0;
    if (run && (ntw == 0 || ntw != run_nt(cin_c, cout_c))) return MOPA_ERR_ARG;
    if (!run && ntw && (cin_c % 16 || cout_c % 16 || ntw > 4 || (cout_c / 16) % ntw)) return MOPA_ERR_ARG;
    if (!ntw && !transpose) return MOPA_ERR_ARG;
    d.src[e] = r[0]; d.dst[e] = r[1]; d.K[e] = K; d.cin[e] = cin; d.cout[e] = cout; d.flags[e] = flags;
    const int64_t ne = (int64_t)K * cin * cout;
    if (ne > nmax) nmax = ne;
  }
  int bx = (int)cdiv64(nmax, 256 * 4);
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  k_pack_w_batched<<<dim3(bx, n), 256, 0, (hipStream_t)stream>>>(d);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ----------------------------------------------------------------------------------------------
// Block kernel on the grouped rulebook: 4 waves = 4 consecutive 64-row tiles walk the filter offsets in lockstep;
// the block stages the column slice W[o][:, c0:c0+16*NTW] of each offset ONCE in LDS (double-buffered, next offset's
// global loads in flight during the current offset's MFMAs), so the MFMA B operand is a conflict-free ds_read instead
// of a dependent L2 load per MFMA, and the L1 path only carries the row gathers.  Each wave still owns its tile's LDS
// accumulator (no atomics); one barrier per offset.  Rows of the next group are gathered while the current group
// is multiplied (C16 = Cin/16 in 1..4: whole rows in registers; C16 = 0: any Cin, chunk-pipelined).
#define SPB_WAVES 4
// NA = capacity of the per-group row registers in 16-channel chunks (1,2,3,4 exact fits; 8 and 12 cover any Cin up to
// 128 / 192 with a guard), so the whole next group is always in flight during the current group's MFMAs.
template <int NTW, int NA, bool ALIGNED>
__global__ __launch_bounds__(64 * SPB_WAVES) void k_spconv_blk(const int* __restrict__ grp_start, const int* __restrict__ grp_o,
                                                                const int* __restrict__ grp_in, const int* __restrict__ grp_out,
                                                                int K, int A_out, const float* __restrict__ in, int ld_in, int cin,
                                                                const float* __restrict__ W, int cout, int w_flip,
                                                                float* __restrict__ out, int ld_out, int osplit) {
  constexpr int CP = NTW * 16;
  constexpr int LD = CP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cinp = (cin + 15) & ~15;
  float* wbuf = smem;                         // [2][cinp][CP]
  float* accs = smem + 2 * cinp * CP;         // [SPB_WAVES][64][LD]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  const int ntiles = (A_out + 63) >> 6;
  const int tile = blockIdx.x * SPB_WAVES + wv;
  const int row0 = tile * 64;
  const int c0 = blockIdx.y * CP;
  const int cin16 = cinp >> 4;
  float* acc = accs + wv * 64 * LD;
  for (int i = lane; i < 64 * LD; i += 64) acc[i] = 0.f;

  // blockIdx.z owns the filter offsets [o_beg, o_end): short levels have too few tiles to fill the chip, so the 27-step
  // chain is cut into `osplit` independent pieces that write partial outputs (summed in order by k_sum_partials).
  const int o_beg = (int)((int64_t)blockIdx.z * K / osplit), o_end = (int)((int64_t)(blockIdx.z + 1) * K / osplit);
  if (osplit > 1) out += (int64_t)blockIdx.z * A_out * ld_out;
  int g = 0, gend = 0;
  if (tile < ntiles) {
    const int gb = grp_start[tile], ge = grp_start[tile + 1];
    g = gb; gend = ge;
    if (osplit > 1) {  // groups are sorted by offset: first group with o >= o_beg / o >= o_end (ballot scan, <= 2 rounds)
      int first = ge, last = ge;
      for (int b0 = gb; b0 < ge; b0 += 64) {
        const int oo = (b0 + lane < ge) ? grp_o[b0 + lane] : K;
        const unsigned long long m1 = __ballot(oo >= o_beg), m2 = __ballot(oo >= o_end);
        if (first == ge && m1) first = b0 + __builtin_ctzll(m1);
        if (last == ge && m2) last = b0 + __builtin_ctzll(m2);
      }
      g = first < last ? first : last;
      gend = last;
    }
  }
  // record pipeline: (offset, input row, output rows) of group g ("c") and g+1 ("n"); g+2 is loaded in the loop
  const int4 none = make_int4(-1, -1, -1, -1);
  int o_c = K, o_n = K, irow_c = -1, irow_n = -1;
  int4 ol_c = none, ol_n = none;
  if (g < gend) {
    o_c = grp_o[g]; irow_c = grp_in[(int64_t)g * 16 + r];
    ol_c = *reinterpret_cast<const int4*>(grp_out + (int64_t)g * 16 + q * 4);
  }
  if (g + 1 < gend) {
    o_n = grp_o[g + 1]; irow_n = grp_in[(int64_t)(g + 1) * 16 + r];
    ol_n = *reinterpret_cast<const int4*>(grp_out + (int64_t)(g + 1) * 16 + q * 4);
  }
  float4 a_c[NA], a_n[NA];
  auto gather = [&](int irow, float4* a) {
#pragma unroll
    for (int kk = 0; kk < NA; ++kk) {
      const int kb = kk * 16 + q * 4;
      if (ALIGNED) {
        a[kk] = (irow >= 0 && kk < cin16) ? *reinterpret_cast<const float4*>(in + (int64_t)irow * ld_in + kb)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const float* ar = in + (int64_t)(irow < 0 ? 0 : irow) * ld_in;
        a[kk].x = (irow >= 0 && kb + 0 < cin) ? ar[kb + 0] : 0.f;
        a[kk].y = (irow >= 0 && kb + 1 < cin) ? ar[kb + 1] : 0.f;
        a[kk].z = (irow >= 0 && kb + 2 < cin) ? ar[kb + 2] : 0.f;
        a[kk].w = (irow >= 0 && kb + 3 < cin) ? ar[kb + 3] : 0.f;
      }
    }
  };
  gather(irow_c, a_c);

  // weight staging (issue-early / write-late): element e = tid + 256*i of the [cinp][CP] slice.  The global loads of
  // offset o+1 are issued before the MFMAs of offset o and land in registers; they are written to the other LDS
  // buffer only after the compute loop, just before the barrier, so their L2 latency hides under the MFMAs.
  constexpr int WREG = (NA * 16 * CP + 64 * SPB_WAVES - 1) / (64 * SPB_WAVES);
  const int wel = cinp * CP;
  float wreg[WREG];
  auto stage_load = [&](int o) {
    const float* src = W + (int64_t)(w_flip ? K - 1 - o : o) * cin * cout + c0;
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int e = tid + 64 * SPB_WAVES * i;
      const int k = e / CP, c = e - k * CP;
      wreg[i] = (e < wel && k < cin && c0 + c < cout) ? src[(int64_t)k * cout + c] : 0.f;
    }
  };
  auto stage_write = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int e = tid + 64 * SPB_WAVES * i;
      if (e < wel) dst[e] = wreg[i];
    }
  };
  stage_load(o_beg);
  stage_write(wbuf + (o_beg & 1) * wel);
  __syncthreads();

  for (int o = o_beg; o < o_end; ++o) {
    const int cur = o & 1;
    if (o + 1 < o_end) stage_load(o + 1);
    const float* __restrict__ wl = wbuf + cur * wel + r * NTW;
    while (__builtin_amdgcn_readfirstlane(o_c) == o) {  // this wave's groups of offset o (wave-uniform)
      int o_nn = K, irow_nn = -1;
      int4 ol_nn = none;
      if (g + 2 < gend) {
        o_nn = grp_o[g + 2]; irow_nn = grp_in[(int64_t)(g + 2) * 16 + r];
        ol_nn = *reinterpret_cast<const int4*>(grp_out + (int64_t)(g + 2) * 16 + q * 4);
      }
      gather(irow_n, a_n);  // next group's rows in flight during this group's MFMAs
      f32x4 d[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NA; ++kk) {
        if (kk < cin16) {
          const float av[4] = {a_c[kk].x, a_c[kk].y, a_c[kk].z, a_c[kk].w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float* bp = wl + (kk * 16 + q * 4 + s) * CP;
#pragma unroll
            for (int t = 0; t < NTW; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bp[t], d[t], 0, 0, 0);
          }
        }
      }
      const int ol[4] = {ol_c.x, ol_c.y, ol_c.z, ol_c.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ol[j] >= 0) {
          float* ap = acc + ol[j] * LD + r * NTW;
#pragma unroll
          for (int t = 0; t < NTW; ++t) ap[t] += d[t][j];
        }
      ++g;
      o_c = o_n; o_n = o_nn;
      irow_c = irow_n; irow_n = irow_nn;
      ol_c = ol_n; ol_n = ol_nn;
#pragma unroll
      for (int kk = 0; kk < NA; ++kk) a_c[kk] = a_n[kk];
    }
    if (o + 1 < o_end) stage_write(wbuf + (cur ^ 1) * wel);
    __syncthreads();
  }
  if (tile >= ntiles) return;
  if (ALIGNED) {
    constexpr int V = CP / 4;
    for (int i = lane; i < 64 * V; i += 64) {
      const int rr = i / V, c4 = i - rr * V;
      if (row0 + rr < A_out)
        *reinterpret_cast<float4*>(out + (int64_t)(row0 + rr) * ld_out + c0 + c4 * 4) =
            *reinterpret_cast<const float4*>(acc + rr * LD + c4 * 4);
    }
  } else {
    for (int i = lane; i < 64 * CP; i += 64) {
      const int rr = i / CP, c = i - rr * CP;
      if (row0 + rr < A_out && c0 + c < cout) out[(int64_t)(row0 + rr) * ld_out + c0 + c] = acc[rr * LD + c];
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Pipelined wave kernel on the grouped rulebook, for the long shallow levels (>= ~1500 tiles, Cout <= 64) where the
// bytes are.  One wave = one 64-row tile, all output columns (Cout == 16*NTW).  The work is a flat stream of units
// (group g, 16-channel chunk kc); every unit is 1 row-gather float4 + 4 weight loads per lane, 4*NTW MFMAs and one
// read-add-write of the 16 result rows into the tile's LDS accumulator.
//  * The weights come pre-packed (mopa_spconv_pack_weight) so that a unit's 16 x Cout chunk of W[o] is one contiguous
//    block in which every lane's MFMA B operands are 16*NTW consecutive bytes: 1 + NTW fully coalesced 16-byte loads
//    per lane and unit (the [K][Cin][Cout] layout needs 4 strided loads; they were 35% of the kernel time).
//  * D units are kept in flight in a statically named register ring (the loop is unrolled by D, every load is
//    unconditional and in-bounds), so the compiler's vmcnt counting stays exact and a wave waits for HBM/L2 about once
//    per D units instead of three times per filter offset.
//  * Per-wave serial latency is what bounds these levels (rocprofv3 SQ counters in profiles/: issue stalls, not memory
//    waits), so a unit has ONE exposed LDS round trip: the group metadata of the next unit is already in registers,
//    and the accumulator rows are read before the unit's MFMAs and written after them.
//  * Metadata is staged through LDS as ready-made offsets (input row -> float4 offset, output row -> byte offset of
//    its accumulator row, filter offset -> byte offset of W[o]); padding rules and the ring's look-ahead past the
//    tile's last group are rewritten to (input row 0 -> accumulator row 64), a sink row that is never written out,
//    which removes every data-dependent branch and select (ds_add_f32 for the accumulation was 5x slower).
//  * LDS per wave is sized so that a whole level is resident at once (16 / 12 / 8 waves per CU at 16 / 32 / 48-64
//    columns): with one equal-length wave per tile a second, partial round would double the kernel time.
// Summation order: filter offsets ascending, 16-channel chunks ascending, k ascending within a chunk (a chunk's partial
// product is added to the accumulator row before the next chunk's; k_spconv_fwd / k_spconv_blk add a whole group's
// product at once, so results agree to rounding, not bit for bit, when Cin > 16).
template <int N> struct FVec;
template <> struct FVec<1> { typedef float T; };
template <> struct FVec<2> { typedef float2 T; };
template <> struct FVec<3> { struct T { float x, y, z; }; };
template <> struct FVec<4> { typedef float4 T; };

template <int NTW, int D, int MU>  // MU: groups consumed per metadata chunk
__global__ __launch_bounds__(64, (NTW == 1 ? 4 : NTW == 2 ? 3 : 2)) void k_spconv_pipe(const int* __restrict__ grp_start, const int* __restrict__ grp_o,
                                                     const int* __restrict__ grp_in, const int* __restrict__ grp_out,
                                                     int K, int A_out, const float* __restrict__ in, int ld_in, int cin,
                                                     const float* __restrict__ W, int w_flip,
                                                     float* __restrict__ out, int ld_out) {
  constexpr int CP = NTW * 16;  // == Cout
  constexpr int LD = CP + 4;
  constexpr int MS = MU + 4;  // groups staged per chunk; the ring's look-ahead (up to 2D-1 units) is clamped to the last one
  static_assert(65 * LD * 4 < 65536 && MS % 4 == 0, "metadata packing");
  typedef typename FVec<NTW>::T BT;
  __shared__ __attribute__((aligned(16))) float acc[65 * LD];  // row 64: sink of padding / look-ahead rules
  __shared__ __attribute__((aligned(16))) unsigned m_in[MS * 16];         // input row * (ld_in / 4): float4 offset
  __shared__ __attribute__((aligned(16))) unsigned short m_out[MS * 16];  // accumulator row * LD * 4: byte offset
  __shared__ unsigned m_w[MS];                                            // byte offset of the group's W[o]
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x, ntiles = gridDim.x;
  const int row0 = tile * 64;
  const int gb = grp_start[tile], ge = grp_start[tile + 1], G = grp_start[ntiles];
  const int nkc = cin >> 4;
  const unsigned ld4 = (unsigned)ld_in >> 2;
  for (int i = lane; i < 65 * LD / 4; i += 64) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned b_lane = (unsigned)(lane * NTW * 16);  // this lane's 4 x NTW floats inside a packed 16 x CP weight chunk
  const float4* __restrict__ a_lane = reinterpret_cast<const float4*>(in) + q;
  char* acc_lane = reinterpret_cast<char*>(acc) + r * NTW * 4;

  float4 A[D];
  float4 B[D][NTW];  // packed chunk: [s2][t] -> 4 * NTW floats per lane, NTW 16-byte loads

  for (int cb = gb; cb < ge; cb += MU) {
    // stage the metadata of groups [cb, cb + MS); reads past this tile's groups stay inside the arrays
    for (int e = lane * 4; e < MS * 16; e += 256) {
      const int src = min(cb * 16 + e, G * 16 - 4);
      const int4 vi = *reinterpret_cast<const int4*>(grp_in + src);
      const int4 vo = *reinterpret_cast<const int4*>(grp_out + src);
      const bool dead = cb + (e >> 4) >= ge || (e >> 4) >= MU;  // look-ahead groups: loaded, never accumulated
      uint4 wi;
      wi.x = (unsigned)max(vi.x, 0) * ld4; wi.y = (unsigned)max(vi.y, 0) * ld4;
      wi.z = (unsigned)max(vi.z, 0) * ld4; wi.w = (unsigned)max(vi.w, 0) * ld4;
      const unsigned o0 = (unsigned)((dead || vo.x < 0) ? 64 : vo.x) * (LD * 4), o1 = (unsigned)((dead || vo.y < 0) ? 64 : vo.y) * (LD * 4);
      const unsigned o2 = (unsigned)((dead || vo.z < 0) ? 64 : vo.z) * (LD * 4), o3 = (unsigned)((dead || vo.w < 0) ? 64 : vo.w) * (LD * 4);
      *reinterpret_cast<uint4*>(m_in + e) = wi;
      *reinterpret_cast<uint2*>(m_out + e) = make_uint2(o0 | (o1 << 16), o2 | (o3 << 16));
    }
    if (lane < MS) {
      const int o = grp_o[min(cb + lane, G - 1)];
      m_w[lane] = (unsigned)((w_flip ? K - 1 - o : o) * cin * CP * 4);
    }
    __syncthreads();
    const int ng = min(MU, ge - cb);
    const int U = ng * nkc;

    int ig = 0, ikc = 0, cg = 0, ckc = 0;
    unsigned io_n = m_in[r], wo_n = m_w[0];          // metadata of the next unit to issue ...
    uint2 mo_n = *reinterpret_cast<const uint2*>(m_out + q * 4);  // ... and of the next unit to consume
#define SPP_ISSUE(S)                                                                                              \
  {                                                                                                               \
    const unsigned woff_ = __builtin_amdgcn_readfirstlane(wo_n) + (unsigned)(ikc * 16 * CP * 4);                  \
    const char* wp_ = reinterpret_cast<const char*>(W) + woff_;                                                   \
    A[S] = a_lane[(uint64_t)io_n + (unsigned)(ikc * 4)];                                                          \
    _Pragma("unroll") for (int v_ = 0; v_ < NTW; ++v_)                                                            \
      B[S][v_] = *reinterpret_cast<const float4*>(wp_ + b_lane + v_ * 16);                                        \
    if (++ikc == nkc) { ikc = 0; ig = min(ig + 1, MS - 1); }                                                      \
    io_n = m_in[ig * 16 + r];                                                                                     \
    wo_n = m_w[ig];                                                                                               \
  }
#pragma unroll
    for (int s = 0; s < D; ++s) {
      SPP_ISSUE(s);
      __builtin_amdgcn_sched_barrier(0);  // keep the ring in issue order: the loop's counted vmcnt relies on it
    }
    // U is rounded up to whole rings: the extra units are look-ahead groups (sink row)
    for (int u = 0; u < U; u += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const unsigned ol[4] = {mo_n.x & 0xffffu, mo_n.x >> 16, mo_n.y & 0xffffu, mo_n.y >> 16};
        BT v[4];  // the 4 rows of a lane are distinct output rows (or the sink): read all, then write all
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const BT*>(acc_lane + ol[j]);
        __builtin_amdgcn_sched_barrier(0);  // keep the 4 accumulator reads in flight under the MFMAs
        f32x4 d[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
          const float av[4] = {A[s].x, A[s].y, A[s].z, A[s].w};
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const float* bw = reinterpret_cast<const float*>(&B[s][0]) + s2 * NTW;
#pragma unroll
            for (int t = 0; t < NTW; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bw[t], d[t], 0, 0, 0);
          }
        }
        if (++ckc == nkc) { ckc = 0; ++cg; }
        mo_n = *reinterpret_cast<const uint2*>(m_out + cg * 16 + q * 4);
        SPP_ISSUE(s);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* vf = reinterpret_cast<float*>(&v[j]);
#pragma unroll
          for (int t = 0; t < NTW; ++t) vf[t] += d[t][j];
          *reinterpret_cast<BT*>(acc_lane + ol[j]) = v[j];
        }
      }
    }
#undef SPP_ISSUE
    __syncthreads();
  }
  constexpr int V = CP / 4;
  for (int i = lane; i < 64 * V; i += 64) {
    const int rr = i / V, c4 = i - rr * V;
    if (row0 + rr < A_out)
      *reinterpret_cast<float4*>(out + (int64_t)(row0 + rr) * ld_out + c4 * 4) =
          *reinterpret_cast<const float4*>(acc + rr * LD + c4 * 4);
  }
}

template <int NTW, int D, int MU>
static int launch_pipe(const int* gs, const int* go, const int* gi, const int* gout, int K, int A_out, const float* in, int ld_in,
                       int cin, const float* W, int w_flip, float* out, int ld_out, hipStream_t st) {
  k_spconv_pipe<NTW, D, MU><<<(unsigned)cdiv64(A_out, 64), 64, 0, st>>>(gs, go, gi, gout, K, A_out, in, ld_in, cin, W, w_flip, out, ld_out);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// ----------------------------------------------------------------------------------------------
// Four waves per (64-row tile, column group of 16*NTW columns): the same pipelined unit stream as k_spconv_pipe, cut
// four ways.  What the measurements of k_spconv_pipe said (profiles/bench_spconv.py and DESIGN.md section 3): a wave
// issues in order, so a tile's ~30-65 groups are one serial chain of ~700 cycles per unit no matter how deep the load
// ring is; the heaviest tile (2x the mean) sets the kernel's duration once every tile is resident; and thin units
// (one 16-channel chunk) are dominated by their fixed per-unit work.  Hence:
//   * wave w of the block takes groups w, w+4, ... of the tile, with a private LDS accumulator; the four accumulators
//     are summed in wave order at the end (deterministic) -- chains are 4x shorter and a level is thousands of blocks
//     over several rounds, so the hardware dispatcher balances heavy and light tiles;
//   * a unit is a group times NKU 16-channel chunks (the whole Cin where registers allow): one metadata lookup, one
//     accumulator read-add-write and 4*NKU*NTW MFMAs per unit;
//   * wide outputs are cut into column groups (grid.y) so four accumulators fit 2-4 blocks per CU.
// The block stages the tile's metadata once (shared); everything else is as in k_spconv_pipe.
// In-kernel cycle counters (-DT4_PROFILE; not in the shipped library): wave 0 of the block at the middle of the grid accumulates the
// cycles of the phases of its pass into g_t4_prof (the launcher synchronises and prints them) -- prologue (zeroing, metadata,
// first loads), per unit: accumulator reads + MFMA issue | issue of the next unit's loads | accumulator write-back, epilogue.
#ifdef T4_PROFILE
__device__ long long g_t4_prof[8];
#define T4_CLK() (prof_on ? (long long)__builtin_readcyclecounter() : 0ll)
#else
#define T4_CLK() 0ll
#endif
#define T4_PAD 4  // accumulator row padding (floats): 0 fits a 4th block per CU at 32 columns (level 1 -6 %) but costs 3-4 % on the MFMA-heavy levels
template <int NTW, int NKU, int D, bool PART, int NWV>  // NWV waves per tile: 4, or 1 for the 16-column layers of the long levels
__global__ __launch_bounds__(64 * NWV) void k_spconv_t4(const int* __restrict__ grp_start, const int* __restrict__ grp_o,
                                                    const int* __restrict__ grp_in, const int* __restrict__ grp_out,
                                                    int K, int A_out, const float* __restrict__ in, int ld_in, int cin,
                                                    const float* __restrict__ Wp, int w_flip,
                                                    float* __restrict__ out, int ld_out) {
  constexpr int CP = NTW * 16;
  constexpr int LD = CP + T4_PAD;
  constexpr int MU = NWV == 1 ? 40 : 48;  // groups staged per chunk (p99 of the bench geometry: 38-57; more = another chunk)
  constexpr int MS = MU + 4;              // + dead groups for the ring's look-ahead
  constexpr int ACCB = 65 * LD * 4;   // bytes of one accumulator (row 64 = sink)
  static_assert(ACCB < 65536 && ACCB % 16 == 0, "metadata packing");
  typedef typename FVec<NTW>::T BT;
  extern __shared__ float4 smem4[];
  char* smem = reinterpret_cast<char*>(smem4);
  unsigned* m_in = reinterpret_cast<unsigned*>(smem + NWV * ACCB);                          // [MS][16] input row * (ld_in/4)
  unsigned short* m_out = reinterpret_cast<unsigned short*>(smem + NWV * ACCB + MS * 64);   // [MS][16] accumulator row byte offset
  unsigned* m_w = reinterpret_cast<unsigned*>(smem + NWV * ACCB + MS * 96);                 // [MS] byte offset of the group's W[o]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x, ntiles = gridDim.x, cg = blockIdx.y;
  const int row0 = tile * 64;
  const int gb = grp_start[tile], ge = grp_start[tile + 1], G = grp_start[ntiles];
  const int nkc = cin >> 4;
  const int NU = (nkc + NKU - 1) / NKU;  // units per group
  const unsigned ld4 = (unsigned)ld_in >> 2;
  float* acc = reinterpret_cast<float*>(smem + wv * ACCB);
#ifdef T4_PROFILE
  const bool prof_on = blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && wv == 0;
  long long p_t0 = T4_CLK(), p_mma = 0, p_iss = 0, p_wb = 0, p_units = 0, p_first = 0;
#endif
  for (int i = lane; i < 65 * LD / 4; i += 64) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // uniform bases (SGPR) + 32-bit per-lane byte offsets: the loads use the saddr + voffset form, no 64-bit VALU adds.
  // Byte offsets of input rows stay below 2^32 (checked by the launcher: 8 * num_out * ld_in * 4 < 2^32).
  const char* __restrict__ wcg = reinterpret_cast<const char*>(Wp) + (size_t)cg * K * cin * CP * 4;
  #ifdef T4_DUMMYA   // timing probe only (wrong results): every gather reads input row 0 -> no gather traffic beyond one row
#define T4_AROW(x) (a_off)
#else
#define T4_AROW(x) (x)
#endif
#ifdef T4_NOMFMA   // timing probe only (wrong results): the operands are consumed by one VALU add each instead of an MFMA
#define T4_MFMA(a, b, c) ((c) + (a) * (b))
#else
#define T4_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#endif
#ifdef T4_LDSB     // timing probe only (wrong results): the weight fragments come out of LDS (whatever is there) instead of global memory
#define T4_BLOAD(gp, lo) (*reinterpret_cast<const float4*>(smem + ((lo) & 0x3ff0)))
#else
#define T4_BLOAD(gp, lo) (*reinterpret_cast<const float4*>(gp))
#endif
#ifdef T4_DUMMYB   // timing probe only (wrong results): every lane of a weight load reads the same 16 bytes -> no weight traffic through L1
  const unsigned b_off = 0u;
#else
  const unsigned b_off = (unsigned)(lane * NTW * 16);
#endif
  const char* __restrict__ in_b = reinterpret_cast<const char*>(in);
  const unsigned a_off = (unsigned)(q * 16);
  char* acc_lane = reinterpret_cast<char*>(acc) + r * NTW * 4;

  float4 A[D][NKU];
  float4 B[D][NKU][NTW];

  for (int cb = gb; cb < ge; cb += MU) {
    __syncthreads();  // every wave is done with the previous chunk's metadata (and the accumulators are zeroed)
    for (int e = tid * 4; e < MS * 16; e += 256 * NWV) {
      const int src = min(cb * 16 + e, G * 16 - 4);
      const int4 vi = *reinterpret_cast<const int4*>(grp_in + src);
      const int4 vo = *reinterpret_cast<const int4*>(grp_out + src);
      const bool dead = cb + (e >> 4) >= ge || (e >> 4) >= MU;
      uint4 wi;
      wi.x = dead ? 0u : (unsigned)max(vi.x, 0) * ld4; wi.y = dead ? 0u : (unsigned)max(vi.y, 0) * ld4;
      wi.z = dead ? 0u : (unsigned)max(vi.z, 0) * ld4; wi.w = dead ? 0u : (unsigned)max(vi.w, 0) * ld4;
      const unsigned o0 = (unsigned)((dead || vo.x < 0) ? 64 : vo.x) * (LD * 4), o1 = (unsigned)((dead || vo.y < 0) ? 64 : vo.y) * (LD * 4);
      const unsigned o2 = (unsigned)((dead || vo.z < 0) ? 64 : vo.z) * (LD * 4), o3 = (unsigned)((dead || vo.w < 0) ? 64 : vo.w) * (LD * 4);
      *reinterpret_cast<uint4*>(m_in + e) = wi;
      *reinterpret_cast<uint2*>(m_out + e) = make_uint2(o0 | (o1 << 16), o2 | (o3 << 16));
    }
    if (tid < MS) {
      const int o = grp_o[min(cb + tid, G - 1)];
      const bool dead = cb + tid >= ge || tid >= MU;
      m_w[tid] = dead ? 0u : (unsigned)((w_flip ? K - 1 - o : o) * cin * CP * 4);
    }
    __syncthreads();
    const int ng = min(MU, ge - cb);
    const int nmine = ng > wv ? (ng - wv + NWV - 1) / NWV : 0;  // this wave: groups wv, wv+NWV, ...
    const int U = nmine * NU;
    const int gdead = MU + wv;

    int ig = wv < ng ? wv : gdead, iku = 0, cgp = ig, cku = 0;
    unsigned io_n = m_in[ig * 16 + r], wo_n = m_w[ig];
    uint2 mo_n = *reinterpret_cast<const uint2*>(m_out + cgp * 16 + q * 4);
#define T4_ISSUE(S)                                                                                        \
  {                                                                                                        \
    const char* wp_ = wcg + __builtin_amdgcn_readfirstlane(wo_n);                                          \
    const unsigned arow_ = T4_AROW(io_n * 16u + a_off);                                                    \
    _Pragma("unroll") for (int j = 0; j < NKU; ++j) {                                                      \
      const int kc_ = PART ? min(iku * NKU + j, nkc - 1) : iku * NKU + j;                                  \
      A[S][j] = *reinterpret_cast<const float4*>(in_b + (arow_ + (unsigned)(kc_ * 64)));                   \
      _Pragma("unroll") for (int v_ = 0; v_ < NTW; ++v_)                                                   \
        B[S][j][v_] = T4_BLOAD(wp_ + (size_t)kc_ * (16 * CP * 4) + (b_off + v_ * 16), (kc_ * NTW + v_) * 1024 + lane * 16); \
    }                                                                                                      \
    if (++iku == NU) { iku = 0; ig = ig + NWV < ng ? ig + NWV : gdead; }                                       \
    io_n = m_in[ig * 16 + r];                                                                              \
    wo_n = m_w[ig];                                                                                        \
  }
#pragma unroll
    for (int s = 0; s < D; ++s) {
      T4_ISSUE(s);
      __builtin_amdgcn_sched_barrier(0);  // keep the ring in issue order: the loop's counted vmcnt relies on it
    }
#ifdef T4_PROFILE
    if (!p_first) p_first = T4_CLK();
#endif
    for (int u = 0; u < U; u += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
#ifdef T4_PROFILE
        const long long c0 = T4_CLK();
#endif
        const unsigned ol[4] = {mo_n.x & 0xffffu, mo_n.x >> 16, mo_n.y & 0xffffu, mo_n.y >> 16};
        BT v[4];  // the 4 rows of a lane are distinct output rows (or the sink): read all, then write all
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const BT*>(acc_lane + ol[j]);
        __builtin_amdgcn_sched_barrier(0);  // keep the accumulator reads in flight under the MFMAs
        f32x4 d[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NKU; ++j) {
          const bool live = !PART || cku * NKU + j < nkc;
          const float av[4] = {live ? A[s][j].x : 0.f, live ? A[s][j].y : 0.f, live ? A[s][j].z : 0.f, live ? A[s][j].w : 0.f};
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const float* bw = reinterpret_cast<const float*>(&B[s][j][0]) + s2 * NTW;
#pragma unroll
            for (int t = 0; t < NTW; ++t) d[t] = T4_MFMA(av[s2], bw[t], d[t]);
          }
        }
        if (++cku == NU) { cku = 0; cgp = cgp + NWV < ng ? cgp + NWV : gdead; }
        mo_n = *reinterpret_cast<const uint2*>(m_out + cgp * 16 + q * 4);
#ifdef T4_PROFILE
        const long long c1 = T4_CLK();
#endif
        T4_ISSUE(s);
#ifdef T4_PROFILE
        const long long c2 = T4_CLK();
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* vf = reinterpret_cast<float*>(&v[j]);
#pragma unroll
          for (int t = 0; t < NTW; ++t) vf[t] += d[t][j];
          *reinterpret_cast<BT*>(acc_lane + ol[j]) = v[j];
        }
#ifdef T4_PROFILE
        const long long c3 = T4_CLK();
        p_mma += c1 - c0; p_iss += c2 - c1; p_wb += c3 - c2; p_units += 1;
#endif
      }
    }
#undef T4_ISSUE
  }
#ifdef T4_PROFILE
  const long long p_t1 = T4_CLK();
#endif
  __syncthreads();
  // ordered sum of the four partial accumulators; each output element is written exactly once
  constexpr int V = CP / 4;
  const float* a0 = reinterpret_cast<const float*>(smem);
  for (int i = tid; i < 64 * V; i += 64 * NWV) {
    const int rr = i / V, c4 = i - rr * V;
    if (row0 + rr < A_out) {
      float4 sum = *reinterpret_cast<const float4*>(a0 + rr * LD + c4 * 4);
#pragma unroll
      for (int w2 = 1; w2 < NWV; ++w2) {
        const float4 p = *reinterpret_cast<const float4*>(a0 + w2 * (65 * LD) + rr * LD + c4 * 4);
        sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
      }
      *reinterpret_cast<float4*>(out + (int64_t)(row0 + rr) * ld_out + cg * CP + c4 * 4) = sum;
    }
  }
#ifdef T4_PROFILE
  if (prof_on && lane == 0) {
    const long long p_t2 = T4_CLK();
    g_t4_prof[0] = p_first - p_t0; g_t4_prof[1] = p_mma; g_t4_prof[2] = p_iss; g_t4_prof[3] = p_wb; g_t4_prof[4] = p_t2 - p_t1;
    g_t4_prof[5] = p_units; g_t4_prof[6] = p_t2 - p_t0; g_t4_prof[7] = ge - gb;
  }
#endif
}

template <int NTW, int NKU, int D, int NWV = 4>
static int launch_t4(const int* gs, const int* go, const int* gi, const int* gout, int K, int A_out, const float* in, int ld_in,
                     int cin, const float* Wp, int cout, int w_flip, float* out, int ld_out, hipStream_t st) {
  constexpr int CP = NTW * 16, LD = CP + T4_PAD, MS = (NWV == 1 ? 40 : 48) + 4;
  const size_t lds = NWV * 65 * LD * 4 + MS * 100;
  const bool part = ((cin >> 4) % NKU) != 0;
  dim3 grid((unsigned)cdiv64(A_out, 64), cout / CP);
#define T4_GO(P)                                                                                                          \
  {                                                                                                                       \
    auto kern = k_spconv_t4<NTW, NKU, D, P, NWV>;                                                                         \
    /* > 64 KB of dynamic LDS needs the attribute once per kernel; the flag only caches that idempotent call (atomic: any   \
       host thread may be the first) -- it carries no state that a result depends on */                                   \
    static std::atomic<bool> attr_set{false};                                                                             \
    if (lds > 64 * 1024 && !attr_set.load(std::memory_order_acquire)) {                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return MOPA_ERR_LAUNCH;                                                                                           \
      attr_set.store(true, std::memory_order_release);                                                                    \
    }                                                                                                                     \
    kern<<<grid, 64 * NWV, lds, st>>>(gs, go, gi, gout, K, A_out, in, ld_in, cin, Wp, w_flip, out, ld_out);               \
  }
  if (part) T4_GO(true) else T4_GO(false)
#undef T4_GO
#ifdef T4_PROFILE
  {
    long long h[8];
    hipStreamSynchronize(st);
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t4_prof), sizeof(h));
    static int printed = 0;
    if (printed++ % 33 == 0)   // (the bench repeats each launch 33 times)
      printf("[t4 profile] K %d rows %d cin %d cout %d NTW %d NKU %d D %d NWV %d grid %u x %u | middle block wave 0: groups %lld units %lld | cycles: prologue %lld, "
             "per unit: acc read + MFMA issue %.0f | next loads issue %.0f | write-back %.0f, epilogue %lld, total %lld\n",
             K, A_out, cin, cout, NTW, NKU, D, NWV, grid.x, grid.y, h[7], h[5], h[0], h[5] ? (double)h[1] / h[5] : 0.0, h[5] ? (double)h[2] / h[5] : 0.0,
             h[5] ? (double)h[3] / h[5] : 0.0, h[4], h[6]);
  }
#endif
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// out[row][c] = sum_z part[z][row][c] (fixed order: deterministic)
__global__ void k_sum_partials(const float* __restrict__ part, int nsplit, int A_out, int cout, int ld, float* __restrict__ out,
                               int ld_out) {
  const int CQ = cout >> 2;
  const int64_t total = (int64_t)A_out * CQ;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / CQ), cq = (int)(i - (int64_t)row * CQ);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nsplit; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)z * A_out + row) * ld + cq * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (int64_t)row * ld_out + cq * 4) = s;
  }
}

static inline size_t blk_lds_bytes(int ntw, int cin) {
  const int cp = ntw * 16, cinp = (cin + 15) & ~15;
  return (size_t)(2 * cinp * cp + SPB_WAVES * 64 * (cp + 4)) * sizeof(float);
}

// Columns per block: the widest of 64/32/16 that divides Cout's 16-column tiles, fits 64 KB of LDS and still leaves
// >= 512 blocks (2 per CU) when the layer allows it.
static int blk_plan(int A_out, int cin, int cout) {
  const int NT = (cout + 15) / 16;
  const int64_t tiles4 = cdiv64(cdiv64(A_out, 64), SPB_WAVES);
  int best = 1;
  const int cand[3] = {4, 2, 1};
  for (int i = 0; i < 3; ++i) {
    const int c = cand[i];
    if (NT % c) continue;
    if (blk_lds_bytes(c, cin) > 64 * 1024) continue;
    best = c;
    if (tiles4 * (NT / c) >= 512) break;
  }
  return best;
}

// Offsets are split over grid.z when the launch would otherwise hold < ~1000 blocks (and a workspace is given).
static int blk_osplit(int K, int A_out, int cout, int ntw) {
  const int NT = (cout + 15) / 16;
  const int64_t blocks = cdiv64(cdiv64(A_out, 64), SPB_WAVES) * ((NT + ntw - 1) / ntw);
  int64_t s = cdiv64(3072, blocks);
  if (s > 9) s = 9;
  const int64_t cap_bytes = (48ll << 20) / ((int64_t)A_out * cout * 4 + 1);  // partial copies stay below ~48 MB
  if (s > cap_bytes) s = cap_bytes;
  if (s > K / 2) s = K / 2;
  if (s < 1) s = 1;
  return (int)s;
}

template <int NTW, int NA>
static int launch_blk(const int* gs, const int* go, const int* gi, const int* gout, int K, int A_out, const float* in, int ld_in,
                      int cin, const float* W, int cout, int w_flip, float* out, int ld_out, bool aligned, float* part,
                      int osplit, hipStream_t st) {
  const int NT = (cout + 15) / 16;
  dim3 grid((unsigned)cdiv64(cdiv64(A_out, 64), SPB_WAVES), (NT + NTW - 1) / NTW, osplit);
  const size_t lds = blk_lds_bytes(NTW, cin);
  float* dst = osplit > 1 ? part : out;
  const int ldd = osplit > 1 ? cout : ld_out;
  if (aligned)
    k_spconv_blk<NTW, NA, true><<<grid, 64 * SPB_WAVES, lds, st>>>(gs, go, gi, gout, K, A_out, in, ld_in, cin, W, cout, w_flip, dst, ldd, osplit);
  else
    k_spconv_blk<NTW, NA, false><<<grid, 64 * SPB_WAVES, lds, st>>>(gs, go, gi, gout, K, A_out, in, ld_in, cin, W, cout, w_flip, dst, ldd, osplit);
  if (osplit > 1)
    k_sum_partials<<<stream_grid((int64_t)A_out * (cout >> 2), 256), 256, 0, st>>>(part, osplit, A_out, cout, cout, out, ld_out);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

template <int NTW>
static int dispatch_blk_c(const int* gs, const int* go, const int* gi, const int* gout, int K, int A_out, const float* in,
                          int ld_in, int cin, const float* W, int cout, int w_flip, float* out, int ld_out, bool aligned,
                          float* part, int osplit, hipStream_t st) {
#define RB(C) return launch_blk<NTW, C>(gs, go, gi, gout, K, A_out, in, ld_in, cin, W, cout, w_flip, out, ld_out, aligned, part, osplit, st)
  const int c16 = (cin + 15) / 16;
  if (c16 <= 4) {
    switch (c16) {
      case 1: RB(1);
      case 2: RB(2);
      case 3: RB(3);
      default: RB(4);
    }
  }
  if (c16 <= 8) RB(8);
  if (c16 <= 12) RB(12);
  return MOPA_ERR_ARG;
#undef RB
}

MOPA_API size_t mopa_spconv_grouped_workspace_bytes(int32_t K, int32_t num_out, int32_t cout) {
  return align_up((size_t)9 * num_out * cout * sizeof(float), 256);  // upper bound: 9 partial copies of the output
}

// Same contract as mopa_spconv_fwd, on the grouped rulebook of the table (mopa_rulebook_groups_{count,fill}).
// ws (optional, mopa_spconv_grouped_workspace_bytes) lets short levels split the filter offsets over more blocks.
MOPA_API int mopa_spconv_fwd_grouped(const int32_t* grp_start, const int32_t* grp_o, const int32_t* grp_in,
                                     const int32_t* grp_out, int32_t K, int32_t num_out, const float* in, int32_t ld_in,
                                     int32_t cin, const float* weight, int32_t cout, int32_t w_flip, float* out,
                                     int32_t ld_out, void* ws, size_t ws_bytes, void* stream) {
  if (K <= 0 || K > 27 || num_out <= 0 || cin <= 0 || cout <= 0 || ld_in < cin || ld_out < cout) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const bool aligned = (cin % 16 == 0) && (cout % 16 == 0) && (ld_in % 4 == 0) && (ld_out % 4 == 0) &&
                       (((uintptr_t)in | (uintptr_t)out | (uintptr_t)weight) % 16 == 0);
  if (cin > ((w_flip & 2) ? 224 : 192)) return MOPA_ERR_ARG;
  // long levels: one pipelined wave per tile (k_spconv_pipe, packed weights); short levels: 4-wave blocks with
  // LDS-staged weights.  w_flip bit 0 = mirrored filter offsets, bit 1 = weight is packed (mopa_spconv_pack_weight).
  const int packed = (w_flip >> 1) & 1;
  w_flip &= 1;
  if (packed) {
    int pntw;
    const int path = packed_plan(K, num_out, cin, cout, &pntw);
    if (!aligned || path == SP_BLK) return MOPA_ERR_ARG;
    // the pipelined kernels address input rows with 32-bit byte offsets (an input has at most 8x the output's rows)
    if ((int64_t)num_out * 8 * ld_in * 4 >= (1ll << 32)) return MOPA_ERR_ARG;
    const int nkc = cin / 16;
#ifdef MOPA_EXP_RING
    if (path == SP_RING)
      return mopa_ring_launch(grp_start, grp_o, grp_in, grp_out, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, pntw, st);
#endif
    if (path == SP_PIPE && cout == 16 && nkc == 2)   // 32 -> 16: one wave per tile with whole-Cin units (40 vs 42 us)
      return launch_t4<1, 2, 2, 1>(grp_start, grp_o, grp_in, grp_out, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, st);
    if (path == SP_PIPE) {
#define PP(N, DD, MU) return launch_pipe<N, DD, MU>(grp_start, grp_o, grp_in, grp_out, K, num_out, in, ld_in, cin, weight, w_flip, out, ld_out, st)
      switch (cout / 16) {
        case 1: PP(1, 2, 40);   // LDS 9.6 KB -> 16 waves / CU
        case 2: PP(2, 6, 32);   // 13.0 KB -> 12
        case 3: PP(3, 6, 40);   // 17.9 KB ->  8
        default: PP(4, 4, 28);  // 20.9 KB ->  7
      }
#undef PP
    }
    if (K != 27 && cdiv64(num_out, 64) > 800) {   // long down/up tables: 17-28 us vs 20-34 on the dense-table kernel
#define T1(N, KU) return launch_t4<N, KU, 2, 1>(grp_start, grp_o, grp_in, grp_out, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, st)
      if (pntw == 1) { if (nkc == 1) T1(1, 1); if (nkc == 2) T1(1, 2); if (nkc == 3) T1(1, 3); T1(1, 4); }
      if (pntw == 2) { if (nkc == 1) T1(2, 1); if (nkc == 2) T1(2, 2); if (nkc == 3) T1(2, 3); T1(2, 4); }
      if (nkc == 1) T1(3, 1); if (nkc == 2) T1(3, 2); T1(3, 3);
#undef T1
    }
#define T4(N, KU, DD) return launch_t4<N, KU, DD>(grp_start, grp_o, grp_in, grp_out, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, st)
    // unit = group x NKU chunks: the whole Cin up to 64 channels, else 3-4 chunks per unit (registers: NKU*4*(1+NTW) per ring slot)
    if (pntw == 1) {
      if (nkc == 1) T4(1, 1, 4);
      if (nkc == 2) T4(1, 2, 2);
      if (nkc % 5 == 0) T4(1, 5, 2);   // 80 / 160 channels: whole units, no padded chunk
      if (nkc % 7 == 0) T4(1, 7, 2);   // 112 / 224
      if (nkc % 3 == 0) T4(1, 3, 2);
      T4(1, 4, 2);
    }
    if (pntw == 2) {
      if (nkc == 1) T4(2, 1, 2);
      if (nkc == 2) T4(2, 2, 2);
      if (nkc % 5 == 0) T4(2, 5, 2);
      if (nkc % 3 == 0) T4(2, 3, 2);
      T4(2, 4, 2);
    }
    if (nkc == 1) T4(3, 1, 2);
    if (nkc == 2 || nkc == 4) T4(3, 2, 2);
    T4(3, 3, 2);
#undef T4
  }
  const int ntw = aligned ? blk_plan(num_out, cin, cout) : 1;
  if (blk_lds_bytes(ntw, cin) > 64 * 1024) return MOPA_ERR_ARG;
  int osplit = (aligned && ws) ? blk_osplit(K, num_out, cout, ntw) : 1;
  if (osplit > 1 && ws_bytes < (size_t)osplit * num_out * cout * sizeof(float)) osplit = 1;
  float* part = (float*)ws;
#define GO(N) return dispatch_blk_c<N>(grp_start, grp_o, grp_in, grp_out, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, aligned, part, osplit, st)
  switch (ntw) {
    case 1: GO(1);
    case 2: GO(2);
    default: GO(4);
  }
#undef GO
}

// ----------------------------------------------------------------------------------------------
// Weight transpose for backward-data: wt[o][co][ci] = w[o][ci][co].
__global__ void k_transpose_w(const float* __restrict__ w, int K, int cin, int cout, float* __restrict__ wt) {
  const int64_t n = (int64_t)K * cin * cout;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin);
    const int64_t t = i / cin;
    const int co = (int)(t % cout), o = (int)(t / cout);
    wt[i] = w[((int64_t)o * cin + ci) * cout + co];
  }
}

MOPA_API int mopa_spconv_transpose_weight(const float* weight, int32_t K, int32_t cin, int32_t cout, float* wt,
                                          void* stream) {
  if (K <= 0 || cin <= 0 || cout <= 0) return MOPA_ERR_ARG;
  k_transpose_w<<<stream_grid((int64_t)K * cin * cout, 256), 256, 0, (hipStream_t)stream>>>(weight, K, cin, cout, wt);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ----------------------------------------------------------------------------------------------
// Backward-weight: dW[o][ci][co] = sum over rules (in_row -> out_row) of offset o of in[in_row][ci] * dout[out_row][co].
// One wave per (offset, row chunk, block of MU*16 input channels); the rule index is the MFMA K dimension
// (4 rules per instruction).  Partial sums go to slabs [chunk][K][cin][cout] and a second kernel adds the
// chunks in order (deterministic; no float atomics).
//   lane l: r = l&15, q = l>>4:  A[m=r][k=q] = in[rule q][mb*16*MU + r*MU + u]   (MU contiguous floats)
//                                B[k=q][n=r] = dout[rule q][r*NT + t]             (NT contiguous floats)
//   D(u,t): lane holds rows 4*q+j -> ci = mb*16*MU + (4*q+j)*MU + u, column co = r*NT + t.
template <int MU, int NT>
__global__ __launch_bounds__(64) void k_spconv_wgrad(const int* __restrict__ nbr, int A_out, int rows_per_chunk,
                                                      const float* __restrict__ in, int ld_in, int cin,
                                                      const float* __restrict__ dout, int ld_do, int cout,
                                                      float* __restrict__ slabs, int K) {
  __shared__ int l_in[64];
  __shared__ int l_out[64];
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int o = blockIdx.x, chunk = blockIdx.y, mb = blockIdx.z;
  const int cbase = mb * 16 * MU;
  f32x4 acc[MU][NT];
#pragma unroll
  for (int u = 0; u < MU; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int rbeg = chunk * rows_per_chunk;
  const int rend = min(A_out, rbeg + rows_per_chunk);
  const int* __restrict__ nrow = nbr + (int64_t)o * A_out;
  int nb_next = nrow[min(rbeg + lane, A_out - 1)];
  for (int base = rbeg; base < rend; base += 64) {
    const int row = base + lane;
    const int nb = (row < rend) ? nb_next : -1;
    nb_next = nrow[min(row + 64, A_out - 1)];  // next block's entries in flight during this block's MFMAs
    const unsigned long long bal = __ballot(nb >= 0);
    if (bal == 0) continue;
    const int n = __popcll(bal);
    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    __syncthreads();
    if (nb >= 0) { l_in[pos] = nb; l_out[pos] = row; }
    __syncthreads();
    for (int s0 = 0; s0 < n; s0 += 4) {
      const int p = s0 + q;
      const bool ok = p < n;
      const float* __restrict__ ar = in + (int64_t)(ok ? l_in[p] : 0) * ld_in + cbase + r * MU;
      const float* __restrict__ br = dout + (int64_t)(ok ? l_out[p] : 0) * ld_do + r * NT;
      float a[MU], b[NT];
#pragma unroll
      for (int u = 0; u < MU; ++u) a[u] = (ok && cbase + r * MU + u < cin) ? ar[u] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = (ok && r * NT + t < cout) ? br[t] : 0.f;
#pragma unroll
      for (int u = 0; u < MU; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[t], acc[u][t], 0, 0, 0);
    }
  }
  float* __restrict__ sl = slabs + ((int64_t)chunk * K + o) * cin * cout;
#pragma unroll
  for (int u = 0; u < MU; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = cbase + (4 * q + j) * MU + u;
      if (ci < cin) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (r * NT + t < cout) sl[(int64_t)ci * cout + r * NT + t] = acc[u][t][j];
      }
    }
}

// Pipelined form of the same computation for 16-aligned channel counts.  The wave alternates two phases:
//   1. scan: batches of 8 x 64 entries of its offset's table row (all loads in flight at once), ballot-compacted into an
//      LDS list of (input row, output row) element offsets until the list holds ~1000 rules or the chunk ends;
//   2. multiply: a flat stream of k-steps (4 rules each) over the list with D k-steps of row loads in flight in a
//      statically named register ring (unconditional, in-bounds loads; rules past the end are masked on the A operand),
// so the wave waits for memory about once per D k-steps instead of once per k-step.
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int N> struct UVec;
template <> struct UVec<1> { typedef float T; };
template <> struct UVec<2> { typedef f2u T; };
template <> struct UVec<3> { typedef f3u T; };
template <> struct UVec<4> { typedef f4u T; };
#define WG2_CAP 1024
#define WG2_TB 8
// WG2_NWV waves per (offset, row chunk): 1, or 4 on the short levels (each takes a quarter of the rows, ordered combine in LDS)
template <int MU, int NT, int D, int WG2_NWV>
__global__ __launch_bounds__(64 * WG2_NWV) void k_spconv_wgrad2(const int* __restrict__ nbr, int A_out, int rows_per_chunk,
                                                                 const float* __restrict__ in, int ld_in, int cin,
                                                                 const float* __restrict__ dout, int ld_do, int cout,
                                                                 float* __restrict__ slabs, int K) {
  constexpr int NT0 = NT > 4 ? 4 : NT, NT1 = NT > 4 ? NT - 4 : 1;
  // The 27 offsets differ 10x in rule count (the submanifold centre has a rule for every row, a corner offset ~0.1 per
  // row), so with one wave per (offset, chunk) and the whole grid resident -- the short levels, where the slab budget
  // caps the chunk count -- the few dense waves set the kernel's duration (3-4x the mean wave).  There, four waves per
  // block quarter every wave and make the grid several rounds deep, so the dispatcher balances dense and sparse offsets;
  // the slab count per offset stays even (cheap ordered reduction).  Long levels already have >10k waves: one per block.
  extern __shared__ float4 wg_smem4[];
  char* wg_smem = reinterpret_cast<char*>(wg_smem4);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  unsigned* l_in = reinterpret_cast<unsigned*>(wg_smem) + wv * 2 * (WG2_CAP + 64);  // element offset of the rule's input row
  unsigned* l_out = l_in + (WG2_CAP + 64);                                           // ... and of its output-gradient row
  float* comb = reinterpret_cast<float*>(wg_smem + WG2_NWV * 2 * (WG2_CAP + 64) * 4);  // [MU*NT*4][64] combine buffer
  const int o = blockIdx.x, chunk = blockIdx.y, mb = blockIdx.z;
  f32x4 acc[MU][NT];
#pragma unroll
  for (int u = 0; u < MU; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int cbeg = chunk * rows_per_chunk;
  const int cend = min(A_out, cbeg + rows_per_chunk);
  const int sub = WG2_NWV == 1 ? rows_per_chunk : ((rows_per_chunk / WG2_NWV) + 63) & ~63;
  const int rbeg = cbeg + wv * sub;
  const int rend = min(cend, rbeg + sub);
  const int* __restrict__ nrow = nbr + (int64_t)o * A_out;
  const float* __restrict__ a_lane = in + mb * 16 * MU + r * MU;
  const float* __restrict__ b_lane = dout + r * NT;

  typename UVec<MU>::T A[D];
  typename UVec<NT0>::T B0[D];
  typename UVec<NT1>::T B1[D];

  int row = rbeg;
  while (row < rend) {
    // ---- phase 1: compact rules into the list
    int n = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // phase-2 reads of the previous list are done
    while (row < rend && n <= WG2_CAP - 64 * WG2_TB) {
      int nb[WG2_TB];
#pragma unroll
      for (int i = 0; i < WG2_TB; ++i) nb[i] = nrow[min(row + 64 * i + lane, A_out - 1)];
#pragma unroll
      for (int i = 0; i < WG2_TB; ++i) {
        const int rr = row + 64 * i + lane;
        const bool ok = rr < rend && nb[i] >= 0;
        const unsigned long long bal = __ballot(ok);
        const int pos = n + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
        if (ok) { l_in[pos] = (unsigned)nb[i] * (unsigned)ld_in; l_out[pos] = (unsigned)rr * (unsigned)ld_do; }
        n += __popcll(bal);
      }
      row += 64 * WG2_TB;
    }
    l_in[n + lane] = 0u;   // padding read by the last k-step and by the ring's look-ahead: row 0, masked below
    l_out[n + lane] = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: its LDS writes are ordered before its later reads
    const int U = (n + 3) >> 2;
    const int plast = n + 63;

    // ---- phase 2: ring over the k-steps
    int ip = q;  // list position of this lane's rule in the next k-step to issue
    unsigned ai = l_in[ip], bi = l_out[ip];
#define WG2_ISSUE(S)                                                                     \
  {                                                                                      \
    A[S] = *reinterpret_cast<const typename UVec<MU>::T*>(a_lane + ai);                  \
    B0[S] = *reinterpret_cast<const typename UVec<NT0>::T*>(b_lane + bi);                \
    if (NT > 4) B1[S] = *reinterpret_cast<const typename UVec<NT1>::T*>(b_lane + bi + 4); \
    ip = min(ip + 4, plast);                                                             \
    ai = l_in[ip];                                                                       \
    bi = l_out[ip];                                                                      \
  }
    // no separate prologue (the compiler would order its loads differently from the loop's and the counted waits would
    // have to cover both): the ring starts zeroed and the first D k-steps of the loop multiply zeros while they fill it
#pragma unroll
    for (int s = 0; s < D; ++s) {
      A[s] = typename UVec<MU>::T(0.f);
      B0[s] = typename UVec<NT0>::T(0.f);
      B1[s] = typename UVec<NT1>::T(0.f);
    }
    int cp = q - 4 * D;
    for (int u = -D; u < U; u += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const bool ok = cp >= 0 && cp < n;
        cp += 4;
        float a[MU], b[NT];
        {
          const float* af = reinterpret_cast<const float*>(&A[s]);
          const float* b0 = reinterpret_cast<const float*>(&B0[s]);
          const float* b1 = reinterpret_cast<const float*>(&B1[s]);
#pragma unroll
          for (int m = 0; m < MU; ++m) a[m] = ok ? af[m] : 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t) b[t] = t < 4 ? b0[t < 4 ? t : 0] : b1[t >= 4 ? t - 4 : 0];
        }
#pragma unroll
        for (int m = 0; m < MU; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[t], acc[m][t], 0, 0, 0);
        WG2_ISSUE(s);
        __builtin_amdgcn_sched_barrier(0);  // refill the slot right after its k-step (the scheduler would batch all refills at the end)
      }
    }
#undef WG2_ISSUE
  }
  // ordered combine: waves 1..3 hand their accumulators to wave 0 through LDS, one after the other (deterministic)
  for (int w = 1; w < WG2_NWV; ++w) {
    __syncthreads();
    if (wv == w) {
#pragma unroll
      for (int u = 0; u < MU; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) comb[((u * NT + t) * 4 + j) * 64 + lane] = acc[u][t][j];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int u = 0; u < MU; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[u][t][j] += comb[((u * NT + t) * 4 + j) * 64 + lane];
    }
  }
  if (wv != 0) return;
  float* __restrict__ sl = slabs + ((int64_t)chunk * K + o) * cin * cout;
  const int cbase = mb * 16 * MU;
#pragma unroll
  for (int u = 0; u < MU; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = cbase + (4 * q + j) * MU + u;
#pragma unroll
      for (int t = 0; t < NT; ++t) sl[(int64_t)ci * cout + r * NT + t] = acc[u][t][j];
    }
}

// dw[i] (+)= sum_c slabs[c][i].  16 chunk-lanes per element sum strided chunks, then a fixed-order LDS reduction:
// deterministic, and short even with ~1000 chunks.
__global__ __launch_bounds__(256) void k_reduce_slabs(const float* __restrict__ slabs, int nchunks, int64_t n, float* __restrict__ dw,
                                                       int accumulate) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + el;
  float s = 0.f;
  if (i < n)
#pragma unroll 8
    for (int c = cl; c < nchunks; c += 16) s += slabs[(int64_t)c * n + i];
  red[cl][el] = s;
  __syncthreads();
  if (cl == 0 && i < n) {
    float t = accumulate ? dw[i] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][el];
    dw[i] = t;
  }
}

// The stem's weight gradient (one input channel): dW[o][c] = sum_i in[nbr[o][i]] * dout[i][c] -- a reduction over the rows, no GEMM.
// Block = 4 waves over a row range; a lane owns one row per pass (its 16 output-gradient channels in registers, the table entries are
// coalesced 256-byte loads) and wave w owns the offsets o = w, w + 4, ... (7 of 27): 7 x 16 accumulators per lane, summed over the 64
// lanes once at the end; the block's result goes to a slab [block][K][cout] (summed in order by k_reduce_slabs: deterministic).
template <int COUT>
__global__ __launch_bounds__(256) void k_spconv_stem_wgrad(const int* __restrict__ nbr, int K, int A_out, int rows_per_block,
                                                            const float* __restrict__ in, int ld_in, const float* __restrict__ dout, int ld_do,
                                                            float* __restrict__ slabs) {
  constexpr int NO = 7;                      // offsets per wave (4 waves x 7 >= 27)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float acc[NO][COUT];
#pragma unroll
  for (int j = 0; j < NO; ++j)
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[j][c] = 0.f;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(A_out, r0 + rows_per_block);
  for (int base = r0; base < r1; base += 64) {
    const int row = base + lane;
    const bool ok = row < r1;
    int idx[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      const int o = wv + 4 * j;
      idx[j] = (ok && o < K) ? nbr[(int64_t)(o < K ? o : 0) * A_out + row] : -1;
    }
    float dy[COUT];
#pragma unroll
    for (int c = 0; c < COUT; c += 4) {
      const float4 v = ok ? *reinterpret_cast<const float4*>(dout + (int64_t)row * ld_do + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      dy[c] = v.x; dy[c + 1] = v.y; dy[c + 2] = v.z; dy[c + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      const float x = idx[j] >= 0 ? in[(int64_t)idx[j] * ld_in] : 0.f;
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[j][c] = fmaf(x, dy[c], acc[j][c]);
    }
  }
  float* __restrict__ sl = slabs + (int64_t)blockIdx.x * K * COUT;
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int o = wv + 4 * j;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      const float v = wave_sum(acc[j][c]);
      if (lane == 0 && o < K) sl[o * COUT + c] = v;
    }
  }
}

static void wgrad_plan(int K, int A_out, int cin, int cout, int* mu, int* mblocks, int* nchunks, int* rows_per_chunk) {
  const int mt = (cin + 15) / 16;
  int m = 1;
  for (int c = 4; c >= 1; --c)
    if (mt % c == 0) { m = c; break; }
  *mu = m;
  *mblocks = mt / m;
  // Many short chunks: each wave walks its rows 64 at a time with a dependent ballot/compact/gather chain, so the
  // latency is hidden by wave count, not by the loop.  Bounded by the slab traffic (<= 32 MB) and >= 256 rows/chunk.
  const int64_t per = (int64_t)K * (*mblocks);
  // Waves to aim for (profiles/r3_wgrad_chunks.md, us per launch over 16384 / 8192 / ... / 512): the pipelined kernel wants FEW, fat
  // waves -- every wave pays a table scan, a ring fill and a K*Cin*Cout-element slab, so the 8-offset tables (one rule per 8
  // scanned entries) were 1.4-2x slower at the old 16384 than at 512, the 27-offset tables 5-15 % -- while the simple kernel of
  // the unaligned shapes hides its dependent ballot / gather chain by wave count only (16384).
  const bool pipelined = cin % 16 == 0 && cout % 16 == 0;
  const int waves = !pipelined ? 16384 : K != 27 ? 512 : cout <= 16 ? 4096 : A_out < 4096 ? 512 : 2048;
  int64_t nc = cdiv64(waves, per);
  const int64_t max_rows = cdiv64(A_out, 256);
  const int64_t max_slab = (32ll << 20) / ((int64_t)K * cin * cout * 4) + 1;
  if (nc > max_rows) nc = max_rows;
  if (nc > max_slab) nc = max_slab;
  if (nc > 2048) nc = 2048;
  if (nc < 1) nc = 1;
  int rpc = (int)cdiv64(cdiv64(A_out, nc), 64) * 64;
  *nchunks = (int)cdiv64(A_out, rpc);
  *rows_per_chunk = rpc;
}

static inline bool stem_wgrad_shape(int K, int cin, int cout) { return K <= 27 && cin == 1 && cout == 16; }   // (7 x 16 accumulators per lane)
static inline int stem_wgrad_blocks(int A_out, int* rows_per_block) {
  int nb = (int)cdiv64(A_out, 256);
  if (nb > 512) nb = 512;
  *rows_per_block = (int)cdiv64(cdiv64(A_out, nb), 64) * 64;
  return (int)cdiv64(A_out, *rows_per_block);
}
MOPA_API size_t mopa_spconv_wgrad_workspace_bytes(int32_t K, int32_t num_out, int32_t cin, int32_t cout) {
  size_t stem = 0;
  if (stem_wgrad_shape(K, cin, cout)) {   // (the generic plan's size as well: an unaligned output gradient takes that path)
    int rpb;
    stem = align_up((size_t)stem_wgrad_blocks(num_out, &rpb) * K * cin * cout * sizeof(float), 256);
  }
  int mu, mb, nc, rpc;
  wgrad_plan(K, num_out, cin, cout, &mu, &mb, &nc, &rpc);
  const size_t gen = align_up((size_t)nc * K * cin * cout * sizeof(float), 256);
  return gen > stem ? gen : stem;
}

template <int MU>
static int launch_wgrad(int nt, dim3 grid, hipStream_t st, const int* nbr, int A_out, int rpc, const float* in,
                        int ld_in, int cin, const float* dout, int ld_do, int cout, float* slabs, int K, bool aligned) {
  constexpr int DD = MU <= 2 ? 8 : 6;
  if (aligned) {
    // (almost) the whole grid resident at once: split every wave four ways -- unless the combine buffer would leave
    // only two blocks per CU (measured: 128->64 at level 3 190 -> 134 us, 96->48 154 -> 125; 64->96 80 -> 97 without the cap)
    const bool four = (int64_t)grid.x * grid.y * grid.z <= 4200 && MU * nt <= 16;
    switch (nt) {
#define CASE(N) case N: if (four) k_spconv_wgrad2<MU, N, (MU + N <= 6 ? 8 : DD), 4><<<grid, 256, 4 * 2 * (WG2_CAP + 64) * 4 + MU * N * 4 * 64 * 4, st>>>(nbr, A_out, rpc, in, ld_in, cin, dout, ld_do, cout, slabs, K); \
                else k_spconv_wgrad2<MU, N, (MU + N <= 6 ? 8 : DD), 1><<<grid, 64, 2 * (WG2_CAP + 64) * 4 + MU * N * 4 * 64 * 4, st>>>(nbr, A_out, rpc, in, ld_in, cin, dout, ld_do, cout, slabs, K); break
      CASE(1); CASE(2); CASE(3); CASE(4); CASE(5); CASE(6); CASE(7);
#undef CASE
      default: return MOPA_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
  }
  switch (nt) {
#define CASE(N) case N: k_spconv_wgrad<MU, N><<<grid, 64, 0, st>>>(nbr, A_out, rpc, in, ld_in, cin, dout, ld_do, cout, slabs, K); break
    CASE(1); CASE(2); CASE(3); CASE(4); CASE(5); CASE(6); CASE(7);
#undef CASE
    default: return MOPA_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// dW[K][cin][cout] (+= if accumulate) from in[*, cin] and dout[num_out, cout] over the rules of nbr[K][num_out].
MOPA_API int mopa_spconv_bwd_weight(const int32_t* nbr, int32_t K, int32_t num_out, const float* in, int32_t ld_in,
                                    int32_t cin, const float* dout, int32_t ld_dout, int32_t cout, float* dweight,
                                    int32_t accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (K <= 0 || num_out <= 0 || cin <= 0 || cout <= 0 || cout > 112 || ld_in < cin || ld_dout < cout) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_spconv_wgrad_workspace_bytes(K, num_out, cin, cout)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  static const int stem_on = getenv("MOPA_SPCONV_STEM") ? atoi(getenv("MOPA_SPCONV_STEM")) : 1;   // A/B switch (must match the workspace query)
  if (stem_wgrad_shape(K, cin, cout) && stem_on && ld_dout % 4 == 0 && ((uintptr_t)dout & 15) == 0) {
    int rpb;
    const int nb = stem_wgrad_blocks(num_out, &rpb);
    float* sl = (float*)ws;
#define STEM_WG() k_spconv_stem_wgrad<16><<<nb, 256, 0, st>>>(nbr, K, num_out, rpb, in, ld_in, dout, ld_dout, sl)
    STEM_WG();
#undef STEM_WG
    MOPA_CHECK_LAUNCH();
    const int64_t n = (int64_t)K * cin * cout;
    k_reduce_slabs<<<(unsigned)cdiv64(n, 16), 256, 0, st>>>(sl, nb, n, dweight, accumulate);
    MOPA_CHECK_LAUNCH();
    return MOPA_OK;
  }
  int mu, mb, nc, rpc;
  wgrad_plan(K, num_out, cin, cout, &mu, &mb, &nc, &rpc);
  dim3 grid(K, nc, mb);
  float* slabs = (float*)ws;
  const int nt = (cout + 15) / 16;
  // pipelined kernel: whole 16-channel tiles, 32-bit element offsets (row * ld < 2^32) and 4-byte aligned rows
  const bool aligned = cin % 16 == 0 && cout % 16 == 0 && (int64_t)num_out * ld_dout < (1ll << 32) &&
                       (int64_t)num_out * 8 * ld_in < (1ll << 32) && (((uintptr_t)in | (uintptr_t)dout) & 3) == 0;
  int rc;
  switch (mu) {
    case 1: rc = launch_wgrad<1>(nt, grid, st, nbr, num_out, rpc, in, ld_in, cin, dout, ld_dout, cout, slabs, K, aligned); break;
    case 2: rc = launch_wgrad<2>(nt, grid, st, nbr, num_out, rpc, in, ld_in, cin, dout, ld_dout, cout, slabs, K, aligned); break;
    case 3: rc = launch_wgrad<3>(nt, grid, st, nbr, num_out, rpc, in, ld_in, cin, dout, ld_dout, cout, slabs, K, aligned); break;
    default: rc = launch_wgrad<4>(nt, grid, st, nbr, num_out, rpc, in, ld_in, cin, dout, ld_dout, cout, slabs, K, aligned); break;
  }
  if (rc) return rc;
  const int64_t n = (int64_t)K * cin * cout;
  k_reduce_slabs<<<(unsigned)cdiv64(n, 16), 256, 0, st>>>(slabs, nc, n, dweight, accumulate);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
