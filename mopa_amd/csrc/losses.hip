// Loss kernels of the hot path: cross-modal KL, weighted cross-entropy with ignore label, softmax over the
// class dimension, and the SAM-mask consistency loss.  All are HBM-bound row reductions over (N, C) with small C:
// one thread per row, wave-shuffle + LDS block reduction, deterministic two-stage sums (block partials -> one
// finalize block accumulating in double).  Scalars (loss value, upstream gradient, normalisers) stay on the device:
// no host sync between forward and backward.
//
// Reference call sites (all in mopa/train/train_xmuda_mopa.py): KL :389-398,:440-445; CE :354-363,:456-465,:563-567;
// softmax + mask_cons_loss :472-480 with mopa/common/utils/loss.py:241-283.  Oracle: oracle/losses.py.
#include "common.h"

#define MAXC 64
#define LOSS_BLOCK 256

__device__ __forceinline__ double block_sum_d(double v, double* lds) {
  v = wave_sum_d(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += lds[k];
  return t;
}

__device__ __forceinline__ float row_lse(const float* __restrict__ z, int C) {
  float mx = z[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(z[c] - mx);
  return mx + logf(s);
}

// ------------------------------------------------------------------------------------------ softmax KL
// loss = mean_i sum_c q_ic (log q_ic - log p_ic),  p = softmax(a_i), q = softmax(b_i)  (b is the detached target).
__global__ __launch_bounds__(LOSS_BLOCK) void k_kl_partial(const float* __restrict__ a, const float* __restrict__ b, int N, int C,
                                                            double* __restrict__ partial) {
  __shared__ double lds[8];
  double acc = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const float* ar = a + (int64_t)i * C;
    const float* br = b + (int64_t)i * C;
    const float la = row_lse(ar, C), lb = row_lse(br, C);
    float t = 0.f;
    for (int c = 0; c < C; ++c) {
      const float lq = br[c] - lb, q = expf(lq);
      t += (q > 0.f) ? q * (lq - (ar[c] - la)) : 0.f;
    }
    acc += (double)t;
  }
  const double s = block_sum_d(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void k_scalar_finalize(const double* __restrict__ partial, int n, double scale, float* __restrict__ out) {
  __shared__ double lds[8];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[i];
  const double s = block_sum_d(acc, lds);
  if (threadIdx.x == 0) *out = (float)(s * scale);
}
// da_ic = gout * (p_ic - q_ic) / N
__global__ void k_kl_bwd(const float* __restrict__ a, const float* __restrict__ b, int N, int C, const float* __restrict__ gout,
                         float* __restrict__ da) {
  const float g = *gout / (float)N;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const float* ar = a + (int64_t)i * C;
    const float* br = b + (int64_t)i * C;
    const float la = row_lse(ar, C), lb = row_lse(br, C);
    for (int c = 0; c < C; ++c) da[(int64_t)i * C + c] = g * (expf(ar[c] - la) - expf(br[c] - lb));
  }
}

MOPA_API size_t mopa_loss_workspace_bytes(int64_t n_rows) { return align_up((size_t)(2 * 2048 + 16) * sizeof(double), 256); }

MOPA_API int mopa_softmax_kl_fwd(const float* logit_p, const float* logit_q, int32_t N, int32_t C, float* loss, void* ws,
                                 size_t ws_bytes, void* stream) {
  if (N <= 0 || C <= 0 || C > MAXC) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_loss_workspace_bytes(N)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int g = stream_grid(N, LOSS_BLOCK);
  k_kl_partial<<<g, LOSS_BLOCK, 0, st>>>(logit_p, logit_q, N, C, (double*)ws);
  k_scalar_finalize<<<1, 256, 0, st>>>((const double*)ws, g, 1.0 / N, loss);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_softmax_kl_bwd(const float* logit_p, const float* logit_q, int32_t N, int32_t C, const float* gout,
                                 float* dlogit_p, void* stream) {
  if (N <= 0 || C <= 0 || C > MAXC) return MOPA_ERR_ARG;
  k_kl_bwd<<<stream_grid(N, LOSS_BLOCK), LOSS_BLOCK, 0, (hipStream_t)stream>>>(logit_p, logit_q, N, C, gout, dlogit_p);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ weighted CE, ignore label
// loss = sum_i w[y_i] * (lse(z_i) - z_i[y_i]) / sum_i w[y_i]   over rows with y_i != ignore (torch cross_entropy 'mean').
__global__ __launch_bounds__(LOSS_BLOCK) void k_wce_partial(const float* __restrict__ z, const int64_t* __restrict__ y,
                                                             const float* __restrict__ w, int N, int C, int64_t ignore,
                                                             double* __restrict__ partial, int* __restrict__ status) {
  __shared__ double lds[8];
  double num = 0.0, den = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int64_t yi = y[i];
    if (yi == ignore) continue;
    if (yi < 0 || yi >= C) { atomicOr(status, 1); continue; }
    const float* zr = z + (int64_t)i * C;
    const float wi = w ? w[yi] : 1.f;
    num += (double)(wi * (row_lse(zr, C) - zr[yi]));
    den += (double)wi;
  }
  const double sn = block_sum_d(num, lds);
  const double sd = block_sum_d(den, lds);
  if (threadIdx.x == 0) { partial[blockIdx.x] = sn; partial[gridDim.x + blockIdx.x] = sd; }
}
__global__ void k_wce_finalize(const double* __restrict__ partial, int n, float* __restrict__ loss, float* __restrict__ den_out) {
  __shared__ double lds[8];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { a += partial[i]; b += partial[n + i]; }
  const double sn = block_sum_d(a, lds);
  const double sd = block_sum_d(b, lds);
  if (threadIdx.x == 0) { *loss = (float)(sn / sd); *den_out = (float)sd; }
}
__global__ void k_wce_bwd(const float* __restrict__ z, const int64_t* __restrict__ y, const float* __restrict__ w, int N, int C,
                          int64_t ignore, const float* __restrict__ den, const float* __restrict__ gout, float* __restrict__ dz) {
  const float g = *gout / *den;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int64_t yi = y[i];
    float* d = dz + (int64_t)i * C;
    if (yi == ignore || yi < 0 || yi >= C) {
      for (int c = 0; c < C; ++c) d[c] = 0.f;
      continue;
    }
    const float* zr = z + (int64_t)i * C;
    const float l = row_lse(zr, C);
    const float s = g * (w ? w[yi] : 1.f);
    for (int c = 0; c < C; ++c) d[c] = s * (expf(zr[c] - l) - (c == yi ? 1.f : 0.f));
  }
}

// status: device int, bit0 set when a label is outside [0,C) and != ignore.
MOPA_API int mopa_wce_fwd(const float* logits, const int64_t* labels, const float* class_weight, int32_t N, int32_t C,
                          int64_t ignore_index, float* loss, float* den, int32_t* status, void* ws, size_t ws_bytes,
                          void* stream) {
  if (N <= 0 || C <= 0 || C > MAXC) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_loss_workspace_bytes(N)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int g = stream_grid(N, LOSS_BLOCK);
  k_wce_partial<<<g, LOSS_BLOCK, 0, st>>>(logits, labels, class_weight, N, C, ignore_index, (double*)ws, status);
  k_wce_finalize<<<1, 256, 0, st>>>((const double*)ws, g, loss, den);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_wce_bwd(const float* logits, const int64_t* labels, const float* class_weight, int32_t N, int32_t C,
                          int64_t ignore_index, const float* den, const float* gout, float* dlogits, void* stream) {
  if (N <= 0 || C <= 0 || C > MAXC) return MOPA_ERR_ARG;
  k_wce_bwd<<<stream_grid(N, LOSS_BLOCK), LOSS_BLOCK, 0, (hipStream_t)stream>>>(logits, labels, class_weight, N, C,
                                                                                  ignore_index, den, gout, dlogits);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ softmax over the last dim
__global__ void k_softmax_fwd(const float* __restrict__ z, int64_t N, int C, float* __restrict__ p) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float* zr = z + i * C;
    const float l = row_lse(zr, C);
    for (int c = 0; c < C; ++c) p[i * C + c] = expf(zr[c] - l);
  }
}
// dz = p * (dp - sum_c dp*p)
__global__ void k_softmax_bwd(const float* __restrict__ p, const float* __restrict__ dp, int64_t N, int C, float* __restrict__ dz) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot = fmaf(p[i * C + c], dp[i * C + c], dot);
    for (int c = 0; c < C; ++c) dz[i * C + c] = p[i * C + c] * (dp[i * C + c] - dot);
  }
}
MOPA_API int mopa_softmax_fwd(const float* logits, int64_t n_rows, int32_t C, float* probs, void* stream) {
  if (n_rows <= 0 || C <= 0 || C > MAXC) return MOPA_ERR_ARG;
  k_softmax_fwd<<<stream_grid(n_rows, 256), 256, 0, (hipStream_t)stream>>>(logits, n_rows, C, probs);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_softmax_bwd(const float* probs, const float* dprobs, int64_t n_rows, int32_t C, float* dlogits, void* stream) {
  if (n_rows <= 0 || C <= 0 || C > MAXC) return MOPA_ERR_ARG;
  k_softmax_bwd<<<stream_grid(n_rows, 256), 256, 0, (hipStream_t)stream>>>(probs, dprobs, n_rows, C, dlogits);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ SAM-mask consistency loss
// probs (B, HW, C); masks (B, HW) int32 with ids in [0, MAXID) valid, negative = ignore (loss.py:265-266).
// Per (image b, id m): n = #pixels, mu = mean_n P, V = sum_{n,c} (P - mu)^2.
//   loss_m = V / (n C) - [min_entropy] sum_c mu_c log2(mu_c + 1e-30) / log2(Knorm)     (Knorm = probs.shape[1], B.2)
//   loss   = mean_b ( mean_{m valid in b} loss_m )   (an image without valid ids adds 0 but counts, loss.py:278-281)
// Two segmented-reduction passes (means, then centred squares) keep fp32 accurate.  The per-id sums are ORDERED: a wave
// takes 64 consecutive pixels (lane = pixel), walks the distinct ids among them (leader = the first lane not yet served)
// and sums the matching lanes with the fixed xor-butterfly of wave_sum -- wavefront shuffles, no float atomics -- into
// a wave-private LDS accumulator that a single lane per channel read-add-writes; the waves of a block take their
// 64-pixel groups in a fixed order and their accumulators are added in wave order; per-block slabs + an ordered slab
// reduction.  Same bits run to run (tests/test_gpu_losses.py::test_mask_cons_loss_is_bit_reproducible).  SAM masks are
// spatially coherent: a group of 64 row-adjacent pixels holds 1-3 ids, so the leader loop is short.
#define MAXID 256
#define MC_PIX_PER_BLOCK 4096

static inline int mc_waves(int W) {   // waves per block: the wave-private accumulators have to fit 64 KB of LDS
  const int nw = 65536 / (MAXID * W * (int)sizeof(float));
  return nw >= 4 ? 4 : nw >= 2 ? 2 : 1;
}

__global__ __launch_bounds__(256) void k_mc_pass(const float* __restrict__ probs, const int* __restrict__ masks, int HW, int C,
                                                  const float* __restrict__ mu /*null in pass 1: [B][MAXID][C]*/,
                                                  float* __restrict__ slabs /*[B][nblk][MAXID][W]*/, int nblk, int W) {
  extern __shared__ float acc[];  // [waves][MAXID][W]   pass1: W=C+1 (sums, count) ; pass2: W=1 (centred squares)
  const int b = blockIdx.y, nthr = blockDim.x, nw = nthr >> 6;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < nw * MAXID * W; i += nthr) acc[i] = 0.f;
  __syncthreads();
  float* mine = acc + wv * MAXID * W;
  const int p0 = blockIdx.x * MC_PIX_PER_BLOCK, p1 = min(HW, p0 + MC_PIX_PER_BLOCK);
  for (int g0 = p0 + wv * 64; g0 < p1; g0 += nthr) {   // wave-uniform: whole groups of 64 pixels
    const int pix = g0 + lane;
    int id = pix < p1 ? masks[(int64_t)b * HW + pix] : -1;
    if (id >= MAXID) id = -1;
    const float* pr = probs + ((int64_t)b * HW + (pix < p1 ? pix : p0)) * C;
    float v2 = 0.f;
    if (mu && id >= 0) {
      const float* m = mu + ((int64_t)b * MAXID + id) * C;
      for (int c = 0; c < C; ++c) { const float d = pr[c] - m[c]; v2 = fmaf(d, d, v2); }
    }
    unsigned long long todo = __ballot(id >= 0);
    while (todo) {
      const int lid = __shfl(id, __ffsll((long long)todo) - 1, 64);
      const bool hit = id == lid;
      const unsigned long long mb = __ballot(hit);
      todo &= ~mb;
      if (!mu) {
        float keep = 0.f;   // lane c ends up with channel c's sum, lane C with the pixel count
        for (int c = 0; c < C; ++c) {
          const float s = wave_sum(hit ? pr[c] : 0.f);
          if (lane == c) keep = s;
        }
        if (lane == C) keep = (float)__popcll(mb);
        if (lane <= C) mine[lid * W + lane] += keep;
      } else {
        const float s = wave_sum(hit ? v2 : 0.f);
        if (lane == 0) mine[lid] += s;
      }
    }
  }
  __syncthreads();
  float* dst = slabs + ((int64_t)b * nblk + blockIdx.x) * MAXID * W;
  for (int i = threadIdx.x; i < MAXID * W; i += nthr) {
    float s = acc[i];
    for (int w = 1; w < nw; ++w) s += acc[w * MAXID * W + i];
    dst[i] = s;
  }
}

// tab[b][id] = {n, V} ; mu[b][id][c]
__global__ void k_mc_means(const float* __restrict__ slabs, int nblk, int C, float* __restrict__ mu, float* __restrict__ cnt) {
  const int b = blockIdx.y, W = C + 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < MAXID * W; i += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += (double)slabs[((int64_t)b * nblk + k) * MAXID * W + i];
    const int id = i / W, c = i - id * W;
    if (c == C) cnt[b * MAXID + id] = (float)s;
    else mu[((int64_t)b * MAXID + id) * C + c] = (float)s;  // still a sum; divided below once cnt is known
  }
}
__global__ void k_mc_divide(float* __restrict__ mu, const float* __restrict__ cnt, int B, int C) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * MAXID * C; i += gridDim.x * blockDim.x) {
    const float n = cnt[i / C];
    mu[i] = n > 0.f ? mu[i] / n : 0.f;
  }
}
// one block: per-image / total loss; also nvalid[b] (number of valid ids) for the backward.
__global__ void k_mc_finalize(const float* __restrict__ slabs2, int nblk, const float* __restrict__ mu, const float* __restrict__ cnt,
                              int B, int C, float log2K, int min_entropy, float* __restrict__ nvalid, float* __restrict__ loss) {
  __shared__ double lds[8];
  double total = 0.0;
  for (int b = 0; b < B; ++b) {
    double acc = 0.0, nv = 0.0;
    for (int id = threadIdx.x; id < MAXID; id += blockDim.x) {
      const float n = cnt[b * MAXID + id];
      if (n <= 0.f) continue;
      double V = 0.0;
      for (int k = 0; k < nblk; ++k) V += (double)slabs2[((int64_t)b * nblk + k) * MAXID + id];
      double l = V / ((double)n * C);
      if (min_entropy) {
        double e = 0.0;
        for (int c = 0; c < C; ++c) {
          const float m = mu[((int64_t)b * MAXID + id) * C + c];
          e += (double)(m * log2f(m + 1e-30f));
        }
        l -= e / (double)log2K;
      }
      acc += l;
      nv += 1.0;
    }
    const double sa = block_sum_d(acc, lds);
    const double sv = block_sum_d(nv, lds);
    if (threadIdx.x == 0) nvalid[b] = (float)sv;
    if (sv > 0.0) total += sa / sv;
  }
  if (threadIdx.x == 0) *loss = (float)(total / B);
}
// dP = g/(B*M_b) * [ 2 (P - mu)/(n C) - (log2(mu+1e-30) + mu/((mu+1e-30) ln2)) / (n log2K) ]
__global__ void k_mc_bwd(const float* __restrict__ probs, const int* __restrict__ masks, int B, int HW, int C, const float* __restrict__ mu,
                         const float* __restrict__ cnt, const float* __restrict__ nvalid, float log2K, int min_entropy,
                         const float* __restrict__ gout, float* __restrict__ dprobs) {
  const float g = *gout / (float)B;
  const int64_t total = (int64_t)B * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int id = masks[i];
    float* d = dprobs + i * C;
    if (id < 0 || id >= MAXID) {
      for (int c = 0; c < C; ++c) d[c] = 0.f;
      continue;
    }
    const float n = cnt[b * MAXID + id];
    const float s = g / nvalid[b];
    const float* m = mu + ((int64_t)b * MAXID + id) * C;
    for (int c = 0; c < C; ++c) {
      float v = 2.f * (probs[i * C + c] - m[c]) / (n * C);
      if (min_entropy) v -= (log2f(m[c] + 1e-30f) + m[c] / ((m[c] + 1e-30f) * 0.6931471805599453f)) / (n * log2K);
      d[c] = s * v;
    }
  }
}

static inline int mc_nblk(int HW) { return (HW + MC_PIX_PER_BLOCK - 1) / MC_PIX_PER_BLOCK; }

// ws layout: slabs1 [B][nblk][MAXID][C+1] | slabs2 [B][nblk][MAXID]
MOPA_API size_t mopa_mask_cons_workspace_bytes(int32_t B, int32_t HW, int32_t C) {
  return align_up((size_t)B * mc_nblk(HW) * MAXID * (C + 2) * sizeof(float), 256);
}
// state (saved for backward): mu [B][MAXID][C] | cnt [B][MAXID] | nvalid [B]   -> (B*MAXID*(C+1) + B) floats
MOPA_API size_t mopa_mask_cons_state_floats(int32_t B, int32_t C) { return (size_t)B * MAXID * (C + 1) + B; }

MOPA_API int mopa_mask_cons_fwd(const float* probs, const int32_t* masks, int32_t B, int32_t HW, int32_t C, int32_t k_norm,
                                int32_t min_entropy, float* loss, float* state, void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0 || HW <= 0 || C <= 0 || C > 32 || k_norm < 2) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_mask_cons_workspace_bytes(B, HW, C)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = mc_nblk(HW);
  float* slabs1 = (float*)ws;
  float* slabs2 = slabs1 + (size_t)B * nblk * MAXID * (C + 1);
  float* mu = state;
  float* cnt = mu + (size_t)B * MAXID * C;
  float* nvalid = cnt + (size_t)B * MAXID;
  dim3 grid(nblk, B);
  const int nw1 = mc_waves(C + 1), nw2 = mc_waves(1);
  k_mc_pass<<<grid, 64 * nw1, (size_t)nw1 * MAXID * (C + 1) * sizeof(float), st>>>(probs, masks, HW, C, nullptr, slabs1, nblk, C + 1);
  k_mc_means<<<dim3(4, B), 256, 0, st>>>(slabs1, nblk, C, mu, cnt);
  k_mc_divide<<<stream_grid((int64_t)B * MAXID * C, 256), 256, 0, st>>>(mu, cnt, B, C);
  k_mc_pass<<<grid, 64 * nw2, (size_t)nw2 * MAXID * sizeof(float), st>>>(probs, masks, HW, C, mu, slabs2, nblk, 1);
  k_mc_finalize<<<1, 256, 0, st>>>(slabs2, nblk, mu, cnt, B, C, log2f((float)k_norm), min_entropy, nvalid, loss);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_mask_cons_bwd(const float* probs, const int32_t* masks, int32_t B, int32_t HW, int32_t C, int32_t k_norm,
                                int32_t min_entropy, const float* state, const float* gout, float* dprobs, void* stream) {
  if (B <= 0 || HW <= 0 || C <= 0 || C > 32 || k_norm < 2) return MOPA_ERR_ARG;
  const float* mu = state;
  const float* cnt = mu + (size_t)B * MAXID * C;
  const float* nvalid = cnt + (size_t)B * MAXID;
  k_mc_bwd<<<stream_grid((int64_t)B * HW, 256), 256, 0, (hipStream_t)stream>>>(probs, masks, B, HW, C, mu, cnt, nvalid,
                                                                                log2f((float)k_norm), min_entropy, gout, dprobs);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
