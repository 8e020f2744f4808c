// Adam over one flat fp32 parameter buffer (all parameters of a network are views into it, so one launch updates
// the whole model and the same flat gradient buffer is what RCCL all-reduces).  Matches torch.optim.Adam
// (the reference's optimiser: mopa/common/solver/build.py:7-21 -> getattr(torch.optim, 'Adam'), yaml lr 1e-3):
//   g += wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "common.h"

__global__ __launch_bounds__(256) void k_adam_flat(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, float grad_scale) {
  const int64_t n4 = n >> 2;
  const float step = lr / bc1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gp[k] * grad_scale + wd * pp[k];
      mp[k] = b1 * mp[k] + (1.f - b1) * gg;
      vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
      pp[k] -= step * mp[k] / (sqrtf(vp[k]) / bc2_sqrt + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gg = g[i] * grad_scale + wd * p[i];
    m[i] = b1 * m[i] + (1.f - b1) * gg;
    v[i] = b2 * v[i] + (1.f - b2) * gg * gg;
    p[i] -= step * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
  }
}

// bias_correction1 = 1 - beta1^t, bias_correction2_sqrt = sqrt(1 - beta2^t) are computed by the caller (host scalars).
// grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
MOPA_API int mopa_adam_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                            float bias_correction2_sqrt, float grad_scale, void* stream) {
  if (n <= 0) return MOPA_ERR_ARG;
  if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return MOPA_ERR_ARG;
  k_adam_flat<<<stream_grid(n / 4 + 1, 256), 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                                                                            eps, weight_decay, bias_correction1,
                                                                            bias_correction2_sqrt, grad_scale);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
