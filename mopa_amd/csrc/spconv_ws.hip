// Sparse 3D convolution, weight-shared block kernel ("ws"): forward / backward-data on the grouped rulebook.
//
// Why this kernel exists (round-2 measurement): the wave-private kernels of spconv.hip re-read a
// 16 x Cout chunk of W[o] from L2 for every 16-rule group, so 2/3 of the L2 -> CU traffic of a launch is weights
// (level 1, 32 -> 32: 412 MB of weights + 206 MB of gathered rows for 157 MB of algorithmic bytes) and the family sits
// at the L2 gather ceiling (~10 of ~18 TB/s), not at the HBM or MFMA one.  Here a block of NW waves walks the filter
// offsets in lockstep and stages the column slice W[o][:, c0:c0+CP] ONCE per block in LDS (double-buffered; the next
// offset's global loads are issued before the current offset's MFMAs and written after them), every wave reads its
// MFMA operand from there with conflict-free ds_read_b128, and only the row gathers travel through the vector L1.
//
// Same contract and rulebook as mopa_spconv_fwd_grouped (spconv.hip): replaces sparseconvnet's per-offset
// gather-GEMM-scatter launches reached from mopa/models/scn_unet.py:27-28; semantics SURVEY.md A.4/A.5; oracle
// oracle/scn3d.py::sparse_conv.
//
// Block = TPB tiles of 64 output rows x NWT waves per tile (NW = TPB * NWT waves).  Wave (tile, sub) takes groups
// sub, sub + NWT, ... of its tile (groups are sorted by filter offset) and adds into a private LDS accumulator
// [65][CP + 4] (row 64 = sink for padding rules), the NWT accumulators of a tile are summed in wave order at the end and
// every output element is written once.  Per filter offset: one barrier.
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, exact fp32), transposed with respect to spconv.hip so that a lane ends up with
// four CONSECUTIVE output channels of ONE rule -- the accumulate is one 16-byte LDS read-add-write per 16-column tile
// and lane, and a lane needs the metadata of one rule only:
//   lane l: r = l & 15, q = l >> 4
//   A operand (16 x 4) = W[o][16 kc + 4 q + s][c0 + 16 t + r]      from LDS, packed [kc][t][lane][s]  (ds_read_b128)
//   B operand (4 x 16) = in[rule r][16 kc + 4 q + s]                one global float4 per chunk kc
//   D (16 x 16): lane holds out channels c0 + 16 t + 4 q + (0..3) of rule r.
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// wp[cg][o][kc][t][lane][s] = Wc[o][16 kc + 4 (lane >> 4) + s][cg * 16 ntw + 16 t + (lane & 15)], Wc = w ([K][cin][cout]) or,
// for backward-data (transpose), its per-offset transpose.  One (cg, o) slice = cin_c x 16 ntw floats, contiguous.
__global__ void k_pack_w_ws(const float* __restrict__ w, int K, int cin_w, int cout_w, int transpose, int ntw, float* __restrict__ wp) {
  const int cin_c = transpose ? cout_w : cin_w, cout_c = transpose ? cin_w : cout_w;
  const int nkc = cin_c >> 4;
  const int n = K * cin_c * cout_c;   // < 2^31
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int rem = i;
    const int s = rem & 3; rem >>= 2;
    const int lane = rem & 63; rem >>= 6;
    const int t = rem % ntw; rem /= ntw;
    const int kc = rem % nkc; rem /= nkc;
    const int o = rem % K;
    const int cg = rem / K;
    const int k = kc * 16 + (lane >> 4) * 4 + s, c = (cg * ntw + t) * 16 + (lane & 15);
    wp[i] = transpose ? w[((int64_t)o * cin_w + c) * cout_w + k] : w[((int64_t)o * cin_w + k) * cout_w + c];
  }
}

MOPA_API int mopa_spconv_pack_weight_ws(const float* w, int32_t K, int32_t cin, int32_t cout, int32_t transpose, int32_t ntw,
                                        float* wp, void* stream) {
  const int cin_c = transpose ? cout : cin, cout_c = transpose ? cin : cout;
  if (K <= 0 || cin_c <= 0 || cout_c <= 0 || cin_c % 16 || cout_c % 16 || ntw < 1 || ntw > 3 || (cout_c / 16) % ntw) return MOPA_ERR_ARG;
  const int64_t n = (int64_t)K * cin * cout;
  if (n >= (1ll << 31)) return MOPA_ERR_ARG;
  k_pack_w_ws<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(w, K, cin, cout, transpose, ntw, wp);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// NTW: 16-column tiles per block (CP = 16 NTW columns); NKU: 16-channel chunks per unit (a group is NU = nkc / NKU
// units); WREG: float4 staging registers per thread for one W slice (>= cin * CP / 4 / blockDim).
template <int NTW, int NKU, int WREG>
__global__ __launch_bounds__(512) void k_spconv_ws(const int* __restrict__ grp_start, const int* __restrict__ grp_o,
                                                    const int* __restrict__ grp_in, const int* __restrict__ grp_out,
                                                    int K, int A_out, int ntiles, int nwt,
                                                    const float* __restrict__ in, int ld_in, int cin,
                                                    const float* __restrict__ Wp, int w_flip,
                                                    float* __restrict__ out, int ld_out) {
  constexpr int CP = NTW * 16, LD = CP + 4, ACCB = 65 * LD * 4;
  extern __shared__ float4 ws_smem4[];
  char* smem = reinterpret_cast<char*>(ws_smem4);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  const int NT = blockDim.x, NW = NT >> 6, tpb = NW / nwt;
  // XCD-aware block order: hardware deals blocks round-robin over the 8 XCDs, so blocks b, b + 8, ... share an L2;
  // give each XCD a contiguous range of tiles (their gathers touch neighbouring rows).
  int bx;
  {
    const int nb = gridDim.x, per = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bx = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int tile = bx * tpb + wv / nwt, sub = wv % nwt, cg = blockIdx.y;
  const int nkc = cin >> 4, NU = nkc / NKU;
  const int slice = cin * CP;                      // floats of one (cg, o) weight slice
  float* acc = reinterpret_cast<float*>(smem + wv * ACCB);
  float* wbuf = reinterpret_cast<float*>(smem + NW * ACCB);   // [2][slice]
  for (int i = lane; i < 65 * LD / 4; i += 64) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- weight staging: thread tid owns float4 elements tid + NT * i of a slice
  const float4* __restrict__ wsrc = reinterpret_cast<const float4*>(Wp) + (size_t)cg * K * (slice >> 2);
  const int n4 = slice >> 2;
  float4 wreg[WREG];
  auto stage_load = [&](int o) {
    const float4* __restrict__ s = wsrc + (size_t)(w_flip ? K - 1 - o : o) * n4;
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int e = tid + NT * i;
      wreg[i] = s[e < n4 ? e : 0];
    }
  };
  auto stage_write = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int e = tid + NT * i;
      if (e < n4) reinterpret_cast<float4*>(dst)[e] = wreg[i];
    }
  };
  stage_load(0);

  // ---- this wave's groups: gb + sub, gb + sub + nwt, ... < ge
  int gb = 0, ge = 0;
  if (tile < ntiles) { gb = grp_start[tile]; ge = grp_start[tile + 1]; }
  const int G = grp_start[ntiles];
  // metadata pipeline: "c" = group being multiplied, "n" = next (its rows are being gathered), "nn" = in flight
  auto meta = [&](int g, int& o, unsigned& ioff, unsigned& ooff) {
    const bool live = g < ge;
    const int gc = live ? g : (G > 0 ? G - 1 : 0);
    const int oo = grp_o[gc], ir = grp_in[(int64_t)gc * 16 + r], orow = grp_out[(int64_t)gc * 16 + r];
    o = live ? oo : K;                                             // K = sentinel: never the current offset
    ioff = (unsigned)((live && ir >= 0) ? ir : 0) * (unsigned)(ld_in * 4) + (unsigned)(q * 16);
    ooff = (unsigned)((live && orow >= 0) ? orow : 64) * (unsigned)(LD * 4) + (unsigned)(q * 16);
  };
  const char* __restrict__ in_b = reinterpret_cast<const char*>(in);
  auto gather = [&](unsigned ioff, int ku, float4* a) {
#pragma unroll
    for (int j = 0; j < NKU; ++j) a[j] = *reinterpret_cast<const float4*>(in_b + (size_t)ioff + (unsigned)((ku * NKU + j) * 64));
  };
  int o_c, o_n, o_nn;
  unsigned io_c, io_n, io_nn, oo_c, oo_n, oo_nn;
  int g = gb + sub;
  meta(g, o_c, io_c, oo_c);
  meta(g + nwt, o_n, io_n, oo_n);
  float4 a_c[NKU], a_n[NKU];
  gather(io_c, 0, a_c);

  stage_write(wbuf);
  __syncthreads();

  char* acc_b = reinterpret_cast<char*>(acc);
  for (int o = 0; o < K; ++o) {
    const int cur = o & 1;
    if (o + 1 < K) stage_load(o + 1);
    const char* __restrict__ wl = reinterpret_cast<const char*>(wbuf + cur * slice) + lane * 16;
    while (__builtin_amdgcn_readfirstlane(o_c) == o) {
      meta(g + 2 * nwt, o_nn, io_nn, oo_nn);
      for (int ku = 0; ku < NU; ++ku) {
        // rows of the next unit in flight during this unit's MFMAs: the next chunks of this group, or the next group
        if (ku + 1 < NU) gather(io_c, ku + 1, a_n);
        else gather(io_n, 0, a_n);
        f32x4 d[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NKU; ++j) {
          const int kc = ku * NKU + j;
          float4 wf[NTW];
#pragma unroll
          for (int t = 0; t < NTW; ++t) wf[t] = *reinterpret_cast<const float4*>(wl + (size_t)(kc * NTW + t) * 1024);
          const float av[4] = {a_c[j].x, a_c[j].y, a_c[j].z, a_c[j].w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
              const float* wfs = reinterpret_cast<const float*>(&wf[t]);
              d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfs[s], av[s], d[t], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          float4* p = reinterpret_cast<float4*>(acc_b + oo_c + t * 64);
          float4 v = *p;
          v.x += d[t][0]; v.y += d[t][1]; v.z += d[t][2]; v.w += d[t][3];
          *p = v;
        }
#pragma unroll
        for (int j = 0; j < NKU; ++j) a_c[j] = a_n[j];
      }
      g += nwt;
      o_c = o_n; io_c = io_n; oo_c = oo_n;
      o_n = o_nn; io_n = io_nn; oo_n = oo_nn;
    }
    if (o + 1 < K) stage_write(wbuf + (cur ^ 1) * slice);
    __syncthreads();
  }

  // ---- ordered sum of the tile's nwt accumulators; each output element is written exactly once
  if (tile >= ntiles) return;
  constexpr int V = CP / 4;
  const float* a0 = reinterpret_cast<const float*>(smem + (wv - sub) * ACCB);
  const int row0 = tile * 64;
  for (int i = sub * 64 + lane; i < 64 * V; i += 64 * nwt) {
    const int rr = i / V, c4 = i - rr * V;
    if (row0 + rr < A_out) {
      float4 sum = *reinterpret_cast<const float4*>(a0 + rr * LD + c4 * 4);
      for (int w2 = 1; w2 < nwt; ++w2) {
        const float4 p = *reinterpret_cast<const float4*>(a0 + w2 * (65 * LD) + rr * LD + c4 * 4);
        sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
      }
      *reinterpret_cast<float4*>(out + (int64_t)(row0 + rr) * ld_out + cg * CP + c4 * 4) = sum;
    }
  }
}

// Column-group width, waves per tile and tiles per block for a shape (tuning: MOPA_WS_PLAN="ntw,nwt,tpb").
static void ws_plan(int K, int64_t num_out, int cin, int cout, int* ntw, int* nwt, int* tpb) {
  const int nt = cout / 16;
  const int64_t tiles = cdiv64(num_out, 64);
  int w = (nt % 2 == 0) ? 2 : (nt % 3 == 0 ? 3 : 1);
  if (cin > 128 && w > 1) w = 1;                 // the staged slice (cin x 16 w floats, twice) stays <= 32 KB
  if (tiles * (nt / w) < 512) w = 1;             // short levels: more column groups = more blocks
  int wt, tb;
  if (tiles >= 1500) { wt = 1; tb = 4; }
  else if (tiles >= 400) { wt = 2; tb = 2; }
  else { wt = 4; tb = 1; }
  static const char* env = getenv("MOPA_WS_PLAN");
  if (env) {
    int a = 0, b = 0, c = 0;
    if (sscanf(env, "%d,%d,%d", &a, &b, &c) == 3) {
      if (a >= 1 && a <= 3 && nt % a == 0) w = a;
      if (b >= 1 && b <= 8) wt = b;
      if (c >= 1 && c <= 8) tb = c;
      if (wt * tb > 8) tb = 8 / wt > 0 ? 8 / wt : 1;
    }
  }
  *ntw = w; *nwt = wt; *tpb = tb;
}

// Column-group width the packed weights of this shape need (0 = shape not handled by the ws kernel).
MOPA_API int mopa_spconv_ws_ntw(int32_t K, int32_t num_out, int32_t cin, int32_t cout) {
  if (K <= 0 || K > 27 || num_out <= 0 || cin % 16 || cout % 16 || cin <= 0 || cout <= 0 || cin > 224 || cout > 224) return 0;
  int ntw, nwt, tpb;
  ws_plan(K, num_out, cin, cout, &ntw, &nwt, &tpb);
  return ntw;
}

template <int NTW, int NKU>
static int launch_ws(const int* gs, const int* go, const int* gi, const int* gout, int K, int A_out, const float* in, int ld_in,
                     int cin, const float* Wp, int cout, int w_flip, float* out, int ld_out, int nwt, int tpb, hipStream_t st) {
  constexpr int CP = NTW * 16, LD = CP + 4, ACCB = 65 * LD * 4;
  const int NW = nwt * tpb, NT = 64 * NW;
  const int ntiles = (int)cdiv64(A_out, 64);
  const size_t lds = (size_t)NW * ACCB + (size_t)2 * cin * CP * 4;
  if (lds > 160 * 1024) return MOPA_ERR_ARG;
  const int need = (int)cdiv64((int64_t)cin * CP / 4, NT);   // float4 staging registers per thread
  dim3 grid((unsigned)cdiv64(ntiles, tpb), cout / CP);
#define WS_GO(WR)                                                                                                        \
  {                                                                                                                      \
    auto kern = k_spconv_ws<NTW, NKU, WR>;                                                                               \
    if (lds > 64 * 1024 &&                                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return MOPA_ERR_LAUNCH;                                                                                            \
    kern<<<grid, NT, lds, st>>>(gs, go, gi, gout, K, A_out, ntiles, nwt, in, ld_in, cin, Wp, w_flip, out, ld_out);       \
  }
  if (need <= 1) WS_GO(1) else if (need <= 2) WS_GO(2) else if (need <= 4) WS_GO(4) else if (need <= 8) WS_GO(8) else return MOPA_ERR_ARG;
#undef WS_GO
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// Same contract as mopa_spconv_fwd_grouped on weights packed by mopa_spconv_pack_weight_ws(ntw = mopa_spconv_ws_ntw(...)).
MOPA_API int mopa_spconv_fwd_ws(const int32_t* grp_start, const int32_t* grp_o, const int32_t* grp_in, const int32_t* grp_out,
                                int32_t K, int32_t num_out, const float* in, int32_t ld_in, int32_t cin, const float* weight_ws,
                                int32_t cout, int32_t w_flip, float* out, int32_t ld_out, void* stream) {
  if (K <= 0 || K > 27 || num_out <= 0 || cin <= 0 || cout <= 0 || ld_in < cin || ld_out < cout) return MOPA_ERR_ARG;
  if (cin % 16 || cout % 16 || cin > 224 || cout > 224 || ld_in % 4 || ld_out % 4) return MOPA_ERR_ARG;
  if ((((uintptr_t)in | (uintptr_t)out | (uintptr_t)weight_ws) & 15) != 0) return MOPA_ERR_ARG;
  // 32-bit byte offsets into the input rows (an input has at most 8x the output's rows)
  if ((int64_t)num_out * 8 * ld_in * 4 >= (1ll << 32)) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int ntw, nwt, tpb;
  ws_plan(K, num_out, cin, cout, &ntw, &nwt, &tpb);
  const int nkc = cin / 16;
  // unit = group x NKU chunks: the whole Cin up to 112 channels, halves above
  const int nku = nkc <= 7 ? nkc : (nkc % 2 == 0 ? nkc / 2 : (nkc % 3 == 0 ? nkc / 3 : 1));
#define WS(N, KU) return launch_ws<N, KU>(grp_start, grp_o, grp_in, grp_out, K, num_out, in, ld_in, cin, weight_ws, cout, w_flip & 1, out, ld_out, nwt, tpb, st)
#define WS_N(N)                    \
  switch (nku) {                   \
    case 1: WS(N, 1);              \
    case 2: WS(N, 2);              \
    case 3: WS(N, 3);              \
    case 4: WS(N, 4);              \
    case 5: WS(N, 5);              \
    case 6: WS(N, 6);              \
    case 7: WS(N, 7);              \
    default: return MOPA_ERR_ARG;  \
  }
  if (ntw == 1) WS_N(1)
  if (ntw == 2) WS_N(2)
  WS_N(3)
#undef WS_N
#undef WS
}
