// Pseudo-label update of the MoPA phase on the device: EMA teacher weights, entropy-weighted 2D/3D
// probability fusion, and the per-class median refinement of the pseudo labels -- the reference does these on the host
// with 8 `.cpu().numpy()` round trips per iteration.
//
// Reference: mopa/train/train_xmuda_mopa.py:264-335 (teacher prediction -> softmax -> fusion -> refine), :587-591 (EMA
// update through torch_ema.ExponentialMovingAverage; 'torch-ema' is an unpinned pip dependency of mopa/setup.py:20 -- its
// published update rule is restated here), mopa/models/losses.py:10-19 (prob_2_entropy),
// mopa/data/utils/refine_pseudo_labels.py:5-22 (refine_pseudo_labels).  Oracle: oracle/pseudo.py, fixture G5.
#include "common.h"

#define PS_MAXC 32

// ------------------------------------------------------------------------------------------ fusion
// p = softmax(logit) per modality; fused: w_m = (1/(ety_m + 1e-30)) / sum_m', ety = -p log2(p + 1e-30) / log2(C),
// p_xm = w_2d p_2d + w_3d p_3d (elementwise, train_xmuda_mopa.py:285-291); outputs max prob and first argmax.
__device__ __forceinline__ void ps_softmax(const float* __restrict__ z, int C, float* p) {
  float mx = z[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) { p[c] = expf(z[c] - mx); s += p[c]; }
  for (int c = 0; c < C; ++c) p[c] = p[c] / s;
}
__global__ void k_pseudo_fuse(const float* __restrict__ la, const float* __restrict__ lb, int N, int C, float inv_log2c,
                              float* __restrict__ maxp, int64_t* __restrict__ label) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    float pa[PS_MAXC], pb[PS_MAXC];
    ps_softmax(la + (int64_t)i * C, C, pa);
    if (lb) {
      ps_softmax(lb + (int64_t)i * C, C, pb);
      for (int c = 0; c < C; ++c) {
        const float ea = -(pa[c] * log2f(pa[c] + 1e-30f)) * inv_log2c, eb = -(pb[c] * log2f(pb[c] + 1e-30f)) * inv_log2c;
        const float ra = 1.f / (ea + 1e-30f), rb = 1.f / (eb + 1e-30f);
        pa[c] = (ra / (ra + rb)) * pa[c] + (rb / (ra + rb)) * pb[c];
      }
    }
    float best = pa[0];
    int arg = 0;
    for (int c = 1; c < C; ++c)
      if (pa[c] > best) { best = pa[c]; arg = c; }
    maxp[i] = best;
    label[i] = arg;
  }
}

// logit_b may be null (single-modality pseudo labels).  maxp (N) fp32, label (N) int64.
MOPA_API int mopa_pseudo_fuse(const float* logit_a, const float* logit_b, int32_t N, int32_t C, float* maxp, int64_t* label,
                              void* stream) {
  if (N <= 0 || C <= 1 || C > PS_MAXC) return MOPA_ERR_ARG;
  k_pseudo_fuse<<<stream_grid(N, 256), 256, 0, (hipStream_t)stream>>>(logit_a, logit_b, N, C, 1.0f / log2f((float)C), maxp, label);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ per-class median refinement
// For every class c present: thresh_c = min(lower median of {prob_i : label_i = c}, 0.9); label_i = ignore where
// prob_i < thresh_c (torch.median of an even-sized set returns the lower middle element = sorted[(n-1)/2]).
// Exact selection without sorting: probabilities are non-negative floats, so their bit patterns order like the values;
// four passes of an 8-bit radix select per class (histogram of the next byte among the elements that match the prefix
// found so far).  Integer histograms -> deterministic, bit-exact with the reference.
// state: [0, C) prefix bits, [C, 2C) remaining rank k, [2C, 3C) class counts; hist: [C][256].
__global__ __launch_bounds__(256) void k_ps_hist(const float* __restrict__ prob, const int64_t* __restrict__ label, int N, int C, int pass,
                                                  const unsigned* __restrict__ state, unsigned* __restrict__ hist) {
  extern __shared__ unsigned lh[];  // [C][256]
  for (int i = threadIdx.x; i < C * 256; i += 256) lh[i] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const int64_t c = label[i];
    if (c < 0 || c >= C) continue;
    const unsigned bits = __float_as_uint(prob[i]);
    if ((bits & himask) == (state[c] & himask)) atomicAdd(&lh[c * 256 + ((bits >> shift) & 255u)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 256; i += 256)
    if (lh[i]) atomicAdd(&hist[i], lh[i]);
}
// one wave per class: pass 0 also derives the class count and the median rank; then the bucket holding rank k
__global__ __launch_bounds__(64) void k_ps_select(int C, int pass, unsigned* __restrict__ state, unsigned* __restrict__ hist) {
  const int c = blockIdx.x, lane = threadIdx.x;
  unsigned h[4], tot = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { h[j] = hist[c * 256 + lane * 4 + j]; tot += h[j]; }
  // inclusive scan of the per-lane totals
  unsigned incl = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  const unsigned n = __shfl(incl, 63, 64);
  unsigned k;
  if (pass == 0) {
    k = n ? (n - 1) / 2 : 0;
    if (lane == 0) state[2 * C + c] = n;
  } else {
    k = state[C + c];
  }
  const unsigned excl = incl - tot;
  const bool mine = n > 0 && k >= excl && k < incl;
  if (mine) {
    unsigned run = excl;
    int b = 3;
    bool found = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!found) {
        if (k < run + h[j]) { found = true; b = j; }
        else run += h[j];
      }
    }
    const int shift = 24 - 8 * pass;
    state[c] = (pass == 0 ? 0u : state[c]) | ((unsigned)(lane * 4 + b) << shift);
    state[C + c] = k - run;
  }
  if (n == 0 && lane == 0 && pass == 0) { state[c] = 0; state[C + c] = 0; }
#pragma unroll
  for (int j = 0; j < 4; ++j) hist[c * 256 + lane * 4 + j] = 0;  // ready for the next pass
}
__global__ void k_ps_apply(const float* __restrict__ prob, const int64_t* __restrict__ label, int N, int C, const unsigned* __restrict__ state,
                           int64_t ignore, int64_t* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int64_t c = label[i];
    int64_t o = c;
    if (c >= 0 && c < C) {
      const float thresh = fminf(__uint_as_float(state[c]), 0.9f);
      if (prob[i] < thresh) o = ignore;
    }
    out[i] = o;
  }
}

MOPA_API size_t mopa_refine_pseudo_labels_workspace_bytes(int32_t C) { return align_up((size_t)(3 * C + C * 256) * sizeof(unsigned), 256); }

// prob (N) fp32 >= 0, label_in (N) int64 in [0, C) (other values pass through), label_out (N) int64 (may alias label_in).
MOPA_API int mopa_refine_pseudo_labels(const float* prob, const int64_t* label_in, int32_t N, int32_t C, int64_t ignore_label,
                                       int64_t* label_out, void* ws, size_t ws_bytes, void* stream) {
  if (N <= 0 || C <= 0 || C > PS_MAXC) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_refine_pseudo_labels_workspace_bytes(C)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  unsigned* state = (unsigned*)ws;
  unsigned* hist = state + 3 * C;
  if (hipMemsetAsync(ws, 0, (size_t)(3 * C + C * 256) * sizeof(unsigned), st) != hipSuccess) return MOPA_ERR_LAUNCH;
  for (int pass = 0; pass < 4; ++pass) {
    k_ps_hist<<<stream_grid(N, 256), 256, (size_t)C * 256 * sizeof(unsigned), st>>>(prob, label_in, N, C, pass, state, hist);
    k_ps_select<<<C, 64, 0, st>>>(C, pass, state, hist);
  }
  k_ps_apply<<<stream_grid(N, 256), 256, 0, st>>>(prob, label_in, N, C, state, ignore_label, label_out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ EMA teacher
// torch_ema.ExponentialMovingAverage.update on one flat buffer: shadow -= (1 - decay) * (shadow - param).
__global__ void k_ema_update(float* __restrict__ shadow, const float* __restrict__ param, int64_t n4, float one_minus_decay) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<float4*>(shadow)[i];
    const float4 p = reinterpret_cast<const float4*>(param)[i];
    s.x -= one_minus_decay * (s.x - p.x); s.y -= one_minus_decay * (s.y - p.y);
    s.z -= one_minus_decay * (s.z - p.z); s.w -= one_minus_decay * (s.w - p.w);
    reinterpret_cast<float4*>(shadow)[i] = s;
  }
}
// n must be a multiple of 4 (FlatAdam pads every parameter to 4 floats); decay already includes torch_ema's warm-up.
MOPA_API int mopa_ema_update(float* shadow, const float* param, int64_t n, float decay, void* stream) {
  if (n <= 0 || (n & 3) || (((uintptr_t)shadow | (uintptr_t)param) & 15)) return MOPA_ERR_ARG;
  k_ema_update<<<stream_grid(n / 4, 256), 256, 0, (hipStream_t)stream>>>(shadow, param, n / 4, 1.0f - decay);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
