// Sparse 3D convolution, "gather-tile" block kernel for wide outputs (64..128 channels): forward / backward-data on
// the grouped rulebook.  Same contract and reference call sites as mopa_spconv_fwd_grouped (spconv.hip):
// sparseconvnet's gather-GEMM-scatter behind mopa/models/scn_unet.py:27-28; semantics SURVEY.md A.4/A.5; oracle
// oracle/scn3d.py::sparse_conv.
//
// Why (round 2, profiles/r2_spconv_layers.md): the deep UNet levels carry few bytes but most of the family's time.
// The wave-private kernels of spconv.hip cut a tile's columns over blocks, so every column group gathers the same
// input rows again and loads a weight chunk per 16-rule group: at 160 -> 80 channels 1.4 GB move from L2 to the CUs for
// 109 MB of algorithmic bytes (10 vector loads per 20 MFMAs -- the texture-address path saturates at half the MFMA rate).
// Here one block owns a 64-row tile and ALL output columns:
//   * a step = (super-unit, k-slice): the rows of up to MB 16-rule groups of one filter offset, KS 16-channel chunks wide,
//     are gathered ONCE by the whole block into LDS (coalesced 16-byte pieces, next step's loads in flight under this
//     step's MFMAs, double-buffered, one barrier per step);
//   * wave w owns column tile w: its MFMA operands are the staged rows (ds_read_b128, shared by all waves) and its own
//     16 columns of W[o] (one float4 per chunk from the packed weights, loaded one step ahead, reused by the MB groups);
//     global loads per MFMA drop from 1/2 to 1/(4 MB) + the gather's 1/(4 nt);
//   * ONE accumulator [65][Cout + 4] in LDS (row 64 = sink of padding rules): waves write disjoint columns, so there
//     are no private copies, no final reduction, and each output element is written once;
//   * short levels: grid.z splits the filter offsets, partial outputs are summed in order by k_gt_sum (deterministic).
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, exact fp32), lane l: r = l & 15, q = l >> 4:
//   A operand = W[o][16 kc + 4 q + s][16 w + r]        packed [w][o][kc][lane][s]     (k_pack_w_gt)
//   B operand = in[rule r of group m][16 kc + 4 q + s]  from the LDS row tile
//   D: lane holds out channels 16 w + 4 q + (0..3) of rule r -> one 16-byte LDS read-add-write per group.
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GT_MAXG 112   // groups of a 64-row tile: <= 27 offsets x 4

// wp[w][o][kc][lane][s] = Wc[o][16 kc + 4 (lane >> 4) + s][16 w + (lane & 15)], Wc = weight ([K][cin][cout]) or, for
// backward-data (transpose), its per-offset transpose.
__global__ void k_pack_w_gt(const float* __restrict__ w, int K, int cin_w, int cout_w, int transpose, float* __restrict__ wp) {
  const int cin_c = transpose ? cout_w : cin_w;
  const int nkc = cin_c >> 4;
  const int n = K * cin_w * cout_w;   // < 2^31
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int rem = i;
    const int s = rem & 3; rem >>= 2;
    const int lane = rem & 63; rem >>= 6;
    const int kc = rem % nkc; rem /= nkc;
    const int o = rem % K;
    const int wt = rem / K;
    const int k = kc * 16 + (lane >> 4) * 4 + s, c = wt * 16 + (lane & 15);
    wp[i] = transpose ? w[((int64_t)o * cin_w + c) * cout_w + k] : w[((int64_t)o * cin_w + k) * cout_w + c];
  }
}

MOPA_API int mopa_spconv_pack_weight_gt(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t transpose, float* wp,
                                        void* stream) {
  if (K <= 0 || cin <= 0 || cout <= 0 || cin % 16 || cout % 16) return MOPA_ERR_ARG;
  const int64_t n = (int64_t)K * cin * cout;
  if (n >= (1ll << 31)) return MOPA_ERR_ARG;
  k_pack_w_gt<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(w, K, cin, cout, transpose, wp);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// MB: groups per super-unit; KS: 16-channel chunks per k-slice; GREG: float4 gather registers per thread
// (>= MB * 16 * KS * 4 / blockDim).
template <int MB, int KS, int GREG>
__global__ __launch_bounds__(512) void k_spconv_gt(const int* __restrict__ grp_start, const int* __restrict__ grp_o,
                                                    const int* __restrict__ grp_in, const int* __restrict__ grp_out,
                                                    int K, int A_out, int ntiles, const float* __restrict__ in, int ld_in, int cin,
                                                    const float* __restrict__ Wp, int w_flip, int cout,
                                                    float* __restrict__ out, int ld_out, int osplit) {
  constexpr int ALD = KS * 16 + 4;             // row stride of the staged row tile (floats)
  constexpr int ABUF = MB * 16 * ALD;          // floats of one row-tile buffer
  extern __shared__ float4 gt_smem4[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  const int NT = blockDim.x;
  const int LD = cout + 4;
  float* acc = reinterpret_cast<float*>(gt_smem4);                 // [65][LD]
  float* abuf = acc + 65 * LD;                                     // [2][ABUF]
  unsigned* m_in = reinterpret_cast<unsigned*>(abuf + 2 * ABUF);   // [GT_MAXG * 16] byte offset of the rule's input row
  unsigned short* m_out = reinterpret_cast<unsigned short*>(m_in + GT_MAXG * 16);  // [GT_MAXG * 16] accumulator row
  unsigned char* m_o = reinterpret_cast<unsigned char*>(m_out + GT_MAXG * 16);     // [GT_MAXG] filter offset
  // XCD-aware block order (blocks b, b + 8, ... share an L2): contiguous tile ranges per XCD
  int tile;
  {
    const int nb = gridDim.x, per = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    tile = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int nkc = cin >> 4, nks = nkc / KS;   // KS divides nkc (gt_plan)
  const int gb = grp_start[tile], ge0 = grp_start[tile + 1];
  const int ng = min(ge0 - gb, GT_MAXG);

  for (int i = tid; i < 65 * LD / 4; i += NT) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < ng * 16; i += NT) {
    const int ir = grp_in[(int64_t)gb * 16 + i], orow = grp_out[(int64_t)gb * 16 + i];
    m_in[i] = (unsigned)(ir >= 0 ? ir : 0) * (unsigned)(ld_in * 4);
    m_out[i] = (unsigned short)((ir >= 0 && orow >= 0) ? orow : 64);
  }
  for (int i = tid; i < ng; i += NT) m_o[i] = (unsigned char)grp_o[gb + i];
  __syncthreads();

  // this block's filter offsets [o_beg, o_end) -> group range [g, gend) (groups are sorted by offset)
  int g = 0, gend = ng;
  if (osplit > 1) {
    const int o_beg = (int)((int64_t)blockIdx.z * K / osplit), o_end = (int)((int64_t)(blockIdx.z + 1) * K / osplit);
    int first = ng, last = ng;
    for (int b0 = 0; b0 < ng; b0 += 64) {
      const int oo = (b0 + lane < ng) ? (int)m_o[b0 + lane] : K;
      const unsigned long long m1 = __ballot(oo >= o_beg), m2 = __ballot(oo >= o_end);
      if (first == ng && m1) first = b0 + __builtin_ctzll(m1);
      if (last == ng && m2) last = b0 + __builtin_ctzll(m2);
    }
    g = first < last ? first : last;
    gend = last;
    out += (int64_t)blockIdx.z * A_out * ld_out;   // partial outputs [z][A_out][ld_out]
  }

  // ---- step state: super-unit (g, mb) x k-slice ks; "n" = the step being staged
  auto su_len = [&](int g0) -> int {
    int mb = 1;
    const int o0 = m_o[g0];
#pragma unroll
    for (int m = 1; m < MB; ++m)
      if (mb == m && g0 + m < gend && (int)m_o[g0 + m] == o0) mb = m + 1;
    return mb;
  };
  const char* __restrict__ in_b = reinterpret_cast<const char*>(in);
  const f32x4* __restrict__ w4 = reinterpret_cast<const f32x4*>(Wp) + (size_t)wv * K * nkc * 64 + lane;
  f32x4 greg[GREG];   // (ext_vector_type, not HIP's float4 struct: arrays of the struct type were not promoted to registers)
  f32x4 wcur[KS], wnxt[KS];
  // (macros, not lambdas: arrays captured by reference inside conditionally executed lambdas ended up in scratch memory)
#define GT_GATHER_LOAD1(I, G0, MBV, KSV)                                                                 \
  if (I < GREG) {                                                                                        \
    const int e_ = tid + NT * I;                                                                         \
    const int row_ = e_ / (KS * 4), p_ = e_ - row_ * (KS * 4);                                           \
    const bool ok_ = e_ < (MBV) * 16 * KS * 4;                                                           \
    const unsigned off_ = m_in[ok_ ? (G0) * 16 + row_ : 0] + (unsigned)((ok_ ? ((KSV) * KS * 16 + p_ * 4) : 0) * 4); \
    greg[I < GREG ? I : 0] = *reinterpret_cast<const f32x4*>(in_b + off_);                              \
  }
#define GT_GATHER_LOAD(G0, MBV, KSV) \
  { GT_GATHER_LOAD1(0, G0, MBV, KSV) GT_GATHER_LOAD1(1, G0, MBV, KSV) GT_GATHER_LOAD1(2, G0, MBV, KSV) GT_GATHER_LOAD1(3, G0, MBV, KSV) }
#define GT_GATHER_WRITE1(I, DST, MBV)                                                                    \
  if (I < GREG) {                                                                                        \
    const int e_ = tid + NT * I;                                                                         \
    const int row_ = e_ / (KS * 4), p_ = e_ - row_ * (KS * 4);                                           \
    if (e_ < (MBV) * 16 * KS * 4) *reinterpret_cast<f32x4*>((DST) + row_ * ALD + p_ * 4) = greg[I < GREG ? I : 0]; \
  }
#define GT_GATHER_WRITE(DST, MBV) \
  { GT_GATHER_WRITE1(0, DST, MBV) GT_GATHER_WRITE1(1, DST, MBV) GT_GATHER_WRITE1(2, DST, MBV) GT_GATHER_WRITE1(3, DST, MBV) }
#define GT_W_LOAD(G0, KSV, WR)                                                                           \
  {                                                                                                      \
    const int o_ = m_o[G0];                                                                              \
    const f32x4* __restrict__ s_ = w4 + (size_t)((w_flip ? K - 1 - o_ : o_) * nkc) * 64;                \
    _Pragma("unroll") for (int j_ = 0; j_ < KS; ++j_) {                                                  \
      WR[j_] = s_[(size_t)((KSV) * KS + j_) * 64];                                                       \
    }                                                                                                    \
  }

  f32x4 d[MB];
  if (g < gend) {
    int mb = su_len(g), ks = 0, buf = 0;
    GT_GATHER_LOAD(g, mb, 0);
    GT_W_LOAD(g, 0, wcur);
    GT_GATHER_WRITE(abuf, mb);
    __syncthreads();
    while (true) {
      // next step
      int g_n = g, mb_n = mb, ks_n = ks + 1;
      if (ks_n == nks) { g_n = g + mb; ks_n = 0; mb_n = g_n < gend ? su_len(g_n) : 0; }
      const bool more = g_n < gend;
      if (more) {
        GT_GATHER_LOAD(g_n, mb_n, ks_n);
        GT_W_LOAD(g_n, ks_n, wnxt);
      }
      if (ks == 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m) d[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const float* __restrict__ ab = abuf + buf * ABUF + r * ALD + q * 4;
      // straight-line code per super-unit size (uniform switch): no branch between an LDS read and its MFMAs
#define GT_STEP(MBV)                                                                                    \
  {                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < KS; ++j) {                                                    \
      f32x4 af[MBV];                                                                                    \
      _Pragma("unroll") for (int m = 0; m < MBV; ++m)                                                   \
        af[m] = *reinterpret_cast<const f32x4*>(ab + m * 16 * ALD + j * 16);                            \
      _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                     \
        _Pragma("unroll") for (int m = 0; m < MBV; ++m)                                                 \
          d[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[j][s], af[m][s], d[m], 0, 0, 0);             \
    }                                                                                                   \
    if (ks == nks - 1) { /* super-unit complete: add its result rows into the accumulator (own columns only) */ \
      _Pragma("unroll") for (int m = 0; m < MBV; ++m) {                                                 \
        f32x4* p = reinterpret_cast<f32x4*>(acc + (int)m_out[(g + m) * 16 + r] * LD + wv * 16 + q * 4); \
        *p = *p + d[m];                                                                                 \
      }                                                                                                 \
    }                                                                                                   \
  }
      if (MB >= 3 && mb == 3) GT_STEP((MB >= 3 ? 3 : 1))
      else if (MB >= 2 && mb == 2) GT_STEP((MB >= 2 ? 2 : 1))
      else GT_STEP(1)
#undef GT_STEP
      if (!more) break;
      GT_GATHER_WRITE(abuf + (buf ^ 1) * ABUF, mb_n);
#pragma unroll
      for (int j = 0; j < KS; ++j) wcur[j] = wnxt[j];
      g = g_n; mb = mb_n; ks = ks_n; buf ^= 1;
      __syncthreads();
    }
  }
#undef GT_GATHER_LOAD
#undef GT_GATHER_WRITE
#undef GT_GATHER_LOAD1
#undef GT_GATHER_WRITE1
#undef GT_W_LOAD
  __syncthreads();
  const int row0 = tile * 64, V = cout >> 2;
  for (int i = tid; i < 64 * V; i += NT) {
    const int rr = i / V, c4 = i - rr * V;
    if (row0 + rr < A_out)
      *reinterpret_cast<float4*>(out + (int64_t)(row0 + rr) * ld_out + c4 * 4) = *reinterpret_cast<const float4*>(acc + rr * LD + c4 * 4);
  }
}

// out[row][c] = sum_z part[z][row][c] in fixed order (deterministic)
__global__ void k_gt_sum(const float* __restrict__ part, int nsplit, int A_out, int cout, float* __restrict__ out, int ld_out) {
  const int CQ = cout >> 2;
  const int64_t total = (int64_t)A_out * CQ;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / CQ), cq = (int)(i - (int64_t)row * CQ);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nsplit; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)z * A_out + row) * cout + cq * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (int64_t)row * ld_out + cq * 4) = s;
  }
}

// Plan: 0 = shape not handled; else the number of offset splits (1 = none).  MOPA_GT_PLAN="mb,ks,osplit" overrides (tuning).
static int gt_plan(int K, int64_t num_out, int cin, int cout, int* mb, int* ks) {
  if (cin % 16 || cout % 16 || cout < 64 || cout > 128 || cin < 16 || cin > 224 || K < 1 || K > 27) return 0;
  const int64_t tiles = cdiv64(num_out, 64);
  *mb = K == 27 ? 2 : 1;
  const int nkc = cin / 16;   // k-slice: the largest divisor of nkc that is <= 7 and keeps the row tile small
  *ks = nkc <= 7 ? nkc : (nkc % 4 == 0 ? 4 : nkc % 5 == 0 ? 5 : nkc % 6 == 0 ? 6 : nkc % 7 == 0 ? 7 : nkc % 3 == 0 ? 3 : nkc % 2 == 0 ? 2 : 1);
  // enough blocks for ~3 rounds of 256 CUs; partial outputs stay small (short levels only)
  int os = 1;
  if (tiles < 768) os = (int)cdiv64(768, tiles);
  if (os > K / 2) os = K / 2;
  if (os > 9) os = 9;
  if (os < 1) os = 1;
  static const char* env = getenv("MOPA_GT_PLAN");
  if (env) {
    int a = 0, b = 0, c = 0;
    if (sscanf(env, "%d,%d,%d", &a, &b, &c) == 3) {
      if (a >= 1 && a <= 3) *mb = a;
      if (b >= 1 && b <= 7 && nkc % b == 0) *ks = b;
      if (c >= 1 && c <= 9 && c <= K) os = c;
    }
  }
  return os;
}

MOPA_API int mopa_spconv_gt_handles(int32_t K, int32_t num_out, int32_t cin, int32_t cout) {
  int mb, ks;
  return gt_plan(K, num_out, cin, cout, &mb, &ks) > 0 ? 1 : 0;
}

MOPA_API size_t mopa_spconv_gt_workspace_bytes(int32_t K, int32_t num_out, int32_t cin, int32_t cout) {
  int mb, ks;
  const int os = gt_plan(K, num_out, cin, cout, &mb, &ks);
  return os > 1 ? align_up((size_t)os * num_out * cout * sizeof(float), 256) : 256;
}

template <int MB, int KS>
static int launch_gt(const int* gs, const int* go, const int* gi, const int* gout, int K, int A_out, const float* in, int ld_in,
                     int cin, const float* Wp, int cout, int w_flip, float* out, int ld_out, int osplit, float* part, hipStream_t st) {
  const int nt = cout / 16, NT = 64 * nt;
  const int ntiles = (int)cdiv64(A_out, 64);
  constexpr int ALD = KS * 16 + 4;
  const size_t lds = (size_t)65 * (cout + 4) * 4 + (size_t)2 * MB * 16 * ALD * 4 + (size_t)GT_MAXG * (64 + 32 + 1) + 16;
  const int need = (int)cdiv64(MB * 16 * KS * 4, NT);
  dim3 grid(ntiles, 1, osplit);
  float* dst = osplit > 1 ? part : out;
  const int ldd = osplit > 1 ? cout : ld_out;
#define GT_GO(GR)                                                                                                        \
  {                                                                                                                      \
    auto kern = k_spconv_gt<MB, KS, GR>;                                                                                 \
    if (lds > 64 * 1024 &&                                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return MOPA_ERR_LAUNCH;                                                                                            \
    kern<<<grid, NT, lds, st>>>(gs, go, gi, gout, K, A_out, ntiles, in, ld_in, cin, Wp, w_flip, cout, dst, ldd, osplit); \
  }
  if (need <= 1) GT_GO(1) else if (need <= 2) GT_GO(2) else if (need <= 3) GT_GO(3) else if (need <= 4) GT_GO(4) else return MOPA_ERR_ARG;
#undef GT_GO
  if (osplit > 1)
    k_gt_sum<<<stream_grid((int64_t)A_out * (cout >> 2), 256), 256, 0, st>>>(part, osplit, A_out, cout, out, ld_out);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// Same contract as mopa_spconv_fwd_grouped; weight_gt packed by mopa_spconv_pack_weight_gt; ws from mopa_spconv_gt_workspace_bytes.
MOPA_API int mopa_spconv_fwd_gt(const int32_t* grp_start, const int32_t* grp_o, const int32_t* grp_in, const int32_t* grp_out,
                                int32_t K, int32_t num_out, const float* in, int32_t ld_in, int32_t cin, const float* weight_gt,
                                int32_t cout, int32_t w_flip, float* out, int32_t ld_out, void* ws, size_t ws_bytes, void* stream) {
  if (num_out <= 0 || ld_in < cin || ld_out < cout || ld_in % 4 || ld_out % 4) return MOPA_ERR_ARG;
  if ((((uintptr_t)in | (uintptr_t)out | (uintptr_t)weight_gt) & 15) != 0) return MOPA_ERR_ARG;
  if ((int64_t)num_out * 8 * ld_in * 4 >= (1ll << 32)) return MOPA_ERR_ARG;   // 32-bit byte offsets into the input rows
  int mb, ks;
  const int os = gt_plan(K, num_out, cin, cout, &mb, &ks);
  if (os <= 0) return MOPA_ERR_ARG;
  if (os > 1 && (ws == nullptr || ws_bytes < (size_t)os * num_out * cout * sizeof(float))) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
#define GT(M, S) return launch_gt<M, S>(grp_start, grp_o, grp_in, grp_out, K, num_out, in, ld_in, cin, weight_gt, cout, w_flip & 1, out, ld_out, os, (float*)ws, st)
#define GT_M(M)                   \
  switch (ks) {                   \
    case 1: GT(M, 1);             \
    case 2: GT(M, 2);             \
    case 3: GT(M, 3);             \
    case 4: GT(M, 4);             \
    case 5: GT(M, 5);             \
    case 6: GT(M, 6);             \
    default: GT(M, 7);            \
  }
  if (mb == 1) GT_M(1)
  if (mb == 2) GT_M(2)
  GT_M(3)
#undef GT_M
#undef GT
}
