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

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TM 64  // output rows per wave-tile

// ----------------------------------------------------------------------------------------------
// Forward / backward-data.  One wave per tile of 64 output rows; per filter offset the wave
// compacts the valid rules (ballot), gathers 16 input rows at a time straight into the MFMA A-operand
// layout with 16-byte loads, multiplies by W[o] and adds the 16 result rows into the tile's LDS
// accumulator (within one offset every output row occurs at most once, so no atomics).  Each output row is
// written to HBM exactly once.
//
// Operand mapping (k and n are permuted consistently so every global load is contiguous per lane):
//   lane l: r = l&15, q = l>>4.   A[i=r][k] = in[pair r][16*kk + 4*q + s]   (one float4 per kk)
//                                 B[k][j=r] = W[o][16*kk + 4*q + s][r*NT + t]  (NT contiguous floats)
//   D tile t: lane holds rows 4*q+j (j<4), i.e. pairs, column r*NT + t.
template <int NT, bool ALIGNED>
__global__ __launch_bounds__(64) void k_spconv_fwd(const int* __restrict__ nbr, int K, int A_out,
                                                    const float* __restrict__ in, int ld_in, int cin,
                                                    const float* __restrict__ W, int cout, int w_flip,
                                                    float* __restrict__ out, int ld_out) {
  constexpr int CP = NT * 16;      // padded Cout
  constexpr int LD = CP + 4;       // LDS row stride (floats); +4 keeps 16-B alignment, breaks pow2 strides
  __shared__ __attribute__((aligned(16))) float acc[TM * LD];
  __shared__ int l_in[TM];
  __shared__ int l_out[TM];
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * TM;
  const int cin16 = (cin + 15) >> 4;

  for (int i = lane; i < TM * LD; i += 64) acc[i] = 0.f;
  __syncthreads();

  for (int o = 0; o < K; ++o) {
    const int row = row0 + lane;
    const int nb = (row < A_out) ? nbr[(int64_t)o * A_out + row] : -1;
    const unsigned long long bal = __ballot(nb >= 0);
    if (bal == 0) continue;  // wave-uniform
    const int n_o = __popcll(bal);
    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    if (nb >= 0) { l_in[pos] = nb; l_out[pos] = lane; }
    __syncthreads();
    const float* __restrict__ wo = W + (int64_t)(w_flip ? K - 1 - o : o) * cin * cout;
    for (int g0 = 0; g0 < n_o; g0 += 16) {
      const int p = g0 + r;
      const int irow = (p < n_o) ? l_in[p] : -1;
      const float* __restrict__ arow = in + (int64_t)(irow < 0 ? 0 : irow) * ld_in;
      f32x4 d[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kk = 0; kk < cin16; ++kk) {
        const int kb = kk * 16 + q * 4;
        float a[4];
        if (ALIGNED) {
          float4 v = (irow >= 0) ? *reinterpret_cast<const float4*>(arow + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
          a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) a[s] = (irow >= 0 && kb + s < cin) ? arow[kb + s] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float b[NT];
          const float* __restrict__ wr = wo + (int64_t)(kb + s) * cout + r * NT;
          if (ALIGNED) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = wr[t];
          } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = (kb + s < cin && r * NT + t < cout) ? wr[t] : 0.f;
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t], d[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pr = g0 + q * 4 + j;
        if (pr < n_o) {
          float* ap = acc + l_out[pr] * LD + r * NT;
#pragma unroll
          for (int t = 0; t < NT; ++t) ap[t] += d[t][j];
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  // write the tile: each output row exactly once, contiguous per row
  if (ALIGNED) {
    constexpr int V = CP / 4;  // float4 per row
    for (int i = lane; i < TM * V; i += 64) {
      const int rr = i / V, c4 = i - rr * V;
      if (row0 + rr < A_out)
        *reinterpret_cast<float4*>(out + (int64_t)(row0 + rr) * ld_out + c4 * 4) =
            *reinterpret_cast<const float4*>(acc + rr * LD + c4 * 4);
    }
  } else {
    for (int i = lane; i < TM * CP; i += 64) {
      const int rr = i / CP, c = i - rr * CP;
      if (row0 + rr < A_out && c < cout) out[(int64_t)(row0 + rr) * ld_out + c] = acc[rr * LD + c];
    }
  }
}

template <int NT>
static int launch_fwd(const int* nbr, int K, int A_out, const float* in, int ld_in, int cin, const float* W,
                      int cout, int w_flip, float* out, int ld_out, hipStream_t st) {
  const int grid = (A_out + TM - 1) / TM;
  const bool aligned = (cin % 16 == 0) && (cout % 16 == 0) && (ld_in % 4 == 0) && (ld_out % 4 == 0) &&
                       (((uintptr_t)in | (uintptr_t)out | (uintptr_t)W) % 16 == 0);
  if (aligned)
    k_spconv_fwd<NT, true><<<grid, 64, 0, st>>>(nbr, K, A_out, in, ld_in, cin, W, cout, w_flip, out, ld_out);
  else
    k_spconv_fwd<NT, false><<<grid, 64, 0, st>>>(nbr, K, A_out, in, ld_in, cin, W, cout, w_flip, out, ld_out);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// out[A_out][cout] (row stride ld_out) = sum_o in[nbr[o][.]] @ W[w_flip ? K-1-o : o]
MOPA_API int mopa_spconv_fwd(const int32_t* nbr, int32_t K, int32_t num_out, const float* in, int32_t ld_in,
                             int32_t cin, const float* weight, int32_t cout, int32_t w_flip, float* out,
                             int32_t ld_out, void* stream) {
  if (K <= 0 || num_out <= 0 || cin <= 0 || cout <= 0 || cout > 192 || ld_in < cin || ld_out < cout) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  switch ((cout + 15) / 16) {
#define CASE(N) case N: return launch_fwd<N>(nbr, K, num_out, in, ld_in, cin, weight, cout, w_flip, out, ld_out, st)
    CASE(1); CASE(2); CASE(3); CASE(4); CASE(5); CASE(6); CASE(7); CASE(8); CASE(9); CASE(10); CASE(11); CASE(12);
#undef CASE
  }
  return MOPA_ERR_ARG;
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
  for (int base = rbeg; base < rend; base += 64) {
    const int row = base + lane;
    const int nb = (row < rend) ? nbr[(int64_t)o * A_out + row] : -1;
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

__global__ void k_reduce_slabs(const float* __restrict__ slabs, int nchunks, int64_t n, float* __restrict__ dw,
                               int accumulate) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = accumulate ? dw[i] : 0.f;
    for (int c = 0; c < nchunks; ++c) s += slabs[(int64_t)c * n + i];
    dw[i] = s;
  }
}

static void wgrad_plan(int K, int A_out, int cin, int* mu, int* mblocks, int* nchunks, int* rows_per_chunk) {
  const int mt = (cin + 15) / 16;
  int m = 1;
  for (int c = 4; c >= 1; --c)
    if (mt % c == 0) { m = c; break; }
  *mu = m;
  *mblocks = mt / m;
  int64_t per = (int64_t)K * (*mblocks);
  int nc = (int)cdiv64(1536, per);          // aim for ~1.5k waves
  int maxc = (int)cdiv64(A_out, 512);       // at least 512 rows per chunk
  if (nc > maxc) nc = maxc;
  if (nc > 64) nc = 64;
  if (nc < 1) nc = 1;
  int rpc = (int)cdiv64(cdiv64(A_out, nc), 64) * 64;
  *nchunks = (int)cdiv64(A_out, rpc);
  *rows_per_chunk = rpc;
}

MOPA_API size_t mopa_spconv_wgrad_workspace_bytes(int32_t K, int32_t num_out, int32_t cin, int32_t cout) {
  int mu, mb, nc, rpc;
  wgrad_plan(K, num_out, cin, &mu, &mb, &nc, &rpc);
  return align_up((size_t)nc * K * cin * cout * sizeof(float), 256);
}

template <int MU>
static int launch_wgrad(int nt, dim3 grid, hipStream_t st, const int* nbr, int A_out, int rpc, const float* in,
                        int ld_in, int cin, const float* dout, int ld_do, int cout, float* slabs, int K) {
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
  int mu, mb, nc, rpc;
  wgrad_plan(K, num_out, cin, &mu, &mb, &nc, &rpc);
  dim3 grid(K, nc, mb);
  float* slabs = (float*)ws;
  const int nt = (cout + 15) / 16;
  int rc;
  switch (mu) {
    case 1: rc = launch_wgrad<1>(nt, grid, st, nbr, num_out, rpc, in, ld_in, cin, dout, ld_dout, cout, slabs, K); break;
    case 2: rc = launch_wgrad<2>(nt, grid, st, nbr, num_out, rpc, in, ld_in, cin, dout, ld_dout, cout, slabs, K); break;
    case 3: rc = launch_wgrad<3>(nt, grid, st, nbr, num_out, rpc, in, ld_in, cin, dout, ld_dout, cout, slabs, K); break;
    default: rc = launch_wgrad<4>(nt, grid, st, nbr, num_out, rpc, in, ld_in, cin, dout, ld_dout, cout, slabs, K); break;
  }
  if (rc) return rc;
  const int64_t n = (int64_t)K * cin * cout;
  k_reduce_slabs<<<stream_grid(n, 256), 256, 0, st>>>(slabs, nc, n, dweight, accumulate);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
