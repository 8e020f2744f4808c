// Micro-benchmark: BatchNorm forward statistics as (partial sums kernel + finalize kernel) against ONE kernel whose last block
// (device-scope atomic ticket, __threadfence) reduces the partials -- the 3D branch's tensor shapes.  What it answers: does folding
// the finalize launch into the statistics kernel save more (a dependent launch: ~7 us kernel + gap) than the fences and the serial
// last block cost?   hipcc --offload-arch=gfx950 -O3 bn_lastblock.hip -o bn_lastblock && ./bn_lastblock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// partial[blk][0][c] = sum(x - x0), partial[blk][1][c] = sum((x - x0)^2)   (as csrc/rows.hip::k_bn_stats_partial)
__device__ __forceinline__ void stats_block(const float* __restrict__ x, int ld, int A, int C, int rpb, float* partial, float* lds) {
  const int CQ = C >> 2, RL = 256 / CQ;
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const int rbeg = blockIdx.x * rpb, rend = min(A, rbeg + rpb);
  float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (rl < RL) {
    const float4 k = *reinterpret_cast<const float4*>(x + cq * 4);
    for (int row = rbeg + rl; row < rend; row += RL) {
      const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)row * ld + cq * 4);
      float d0 = v.x - k.x, d1 = v.y - k.y, d2 = v.z - k.z, d3 = v.w - k.w;
      s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
      ss[0] += d0 * d0; ss[1] += d1 * d1; ss[2] += d2 * d2; ss[3] += d3 * d3;
    }
    for (int j = 0; j < 4; ++j) { lds[(0 * RL + rl) * C + cq * 4 + j] = s[j]; lds[(1 * RL + rl) * C + cq * 4 + j] = ss[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float t = 0.f;
    for (int k = 0; k < RL; ++k) t += lds[(which * RL + k) * C + c];
    partial[(int64_t)blockIdx.x * 2 * C + i] = t;
  }
}
__global__ __launch_bounds__(256) void k_stats(const float* x, int ld, int A, int C, int rpb, float* partial) {
  extern __shared__ float lds[];
  stats_block(x, ld, A, C, rpb, partial, lds);
}
__global__ __launch_bounds__(256) void k_finalize(const float* __restrict__ partial, int nblk, const float* x0, int A, int C, float* stats) {
  __shared__ double red[2][4];
  const int c = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) { s += (double)partial[(int64_t)b * 2 * C + c]; ss += (double)partial[(int64_t)b * 2 * C + C + c]; }
  s = wave_sum_d(s); ss = wave_sum_d(ss);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]; ss = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    const double m = s / A; double v = ss / A - m * m; if (v < 0) v = 0;
    stats[c] = (float)((double)x0[c] + m); stats[C + c] = (float)(1.0 / sqrt(v + 1e-4));
  }
}
// one kernel: every block writes its partials, takes a ticket; the last one reduces all of them (fixed order: deterministic)
__global__ __launch_bounds__(256) void k_stats_last(const float* x, int ld, int A, int C, int rpb, float* partial, int nblk, unsigned* counter, float* stats) {
  extern __shared__ float lds[];
  __shared__ int s_last;
  stats_block(x, ld, A, C, rpb, partial, lds);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(counter, 1u) == (unsigned)nblk - 1);
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  double* red = reinterpret_cast<double*>(lds);   // [2][256] doubles = 4 KB of the 8 KB dynamic LDS
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int Cc = min(256, C - c0), nseg = 256 / Cc;
    const int cl = threadIdx.x % Cc, seg = threadIdx.x / Cc;
    double s = 0.0, ss = 0.0;
    if (seg < nseg)
      for (int b = seg; b < nblk; b += nseg) {
        s += (double)__hip_atomic_load(&partial[(int64_t)b * 2 * C + c0 + cl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ss += (double)__hip_atomic_load(&partial[(int64_t)b * 2 * C + C + c0 + cl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    __syncthreads();
    if (seg < nseg) { red[threadIdx.x] = s; red[256 + threadIdx.x] = ss; }
    __syncthreads();
    if (seg == 0) {
      for (int k = 1; k < nseg; ++k) { s += red[k * Cc + cl]; ss += red[256 + k * Cc + cl]; }
      const double m = s / A; double v = ss / A - m * m; if (v < 0) v = 0;
      stats[c0 + cl] = (float)((double)x[c0 + cl] + m); stats[C + c0 + cl] = (float)(1.0 / sqrt(v + 1e-4));
    }
  }
  if (threadIdx.x == 0) *counter = 0u;   // ready for the next launch on this stream
}
__global__ void k_apply(const float* x, float* y, int64_t n4, int C, const float* stats) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = fmaxf((v.x - stats[c]) * stats[C + c], 0.f); v.y = fmaxf((v.y - stats[c + 1]) * stats[C + c + 1], 0.f);
    v.z = fmaxf((v.z - stats[c + 2]) * stats[C + c + 2], 0.f); v.w = fmaxf((v.w - stats[c + 3]) * stats[C + c + 3], 0.f);
    reinterpret_cast<float4*>(y)[i] = v;
  }
}
int main() {
  const int shapes[][2] = {{257465, 16}, {257465, 32}, {193135, 64}, {103554, 96}, {49022, 128}, {19312, 160}, {6789, 192}, {2330, 224}, {2330, 112}};
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *x, *y, *partial, *stats, *stats2; unsigned* counter;
  CK(hipMalloc(&x, (size_t)257465 * 64 * 4)); CK(hipMalloc(&y, (size_t)257465 * 64 * 4)); CK(hipMalloc(&partial, 2048 * 2 * 256 * 4));
  CK(hipMalloc(&stats, 4096)); CK(hipMalloc(&stats2, 4096)); CK(hipMalloc(&counter, 256)); CK(hipMemset(counter, 0, 256));
  std::vector<float> h((size_t)257465 * 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.003f - 1.f;
  CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("%8s %4s | %6s %6s | 3 kernels us | fused(nblk as now) us | fused(nblk<=256) us | max diff of (mean, invstd)\n", "rows", "C", "nblk", "nblk2");
  for (auto& sh : shapes) {
    const int A = sh[0], C = sh[1];
    if ((size_t)A * C > h.size()) continue;
    auto rpb_of = [&](int cap) { long r = ((long)A + cap - 1) / cap; if (r < 32) r = 32; if (r > 1024 && cap == 2048) r = 1024; return (int)r; };
    const int rpb1 = rpb_of(2048), nblk1 = (A + rpb1 - 1) / rpb1, rpb2 = rpb_of(256), nblk2 = (A + rpb2 - 1) / rpb2;
    const size_t lds = (size_t)2 * (256 / (C >> 2)) * C * 4 > 4096 ? (size_t)2 * (256 / (C >> 2)) * C * 4 : 4096;
    const int64_t n4 = (int64_t)A * C / 4;
    const int ag = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    float t[3];
    for (int mode = 0; mode < 3; ++mode) {
      const int reps = 200;
      for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) {
          if (mode == 0) {
            hipLaunchKernelGGL(k_stats, dim3(nblk1), dim3(256), lds, st, x, C, A, C, rpb1, partial);
            hipLaunchKernelGGL(k_finalize, dim3(C), dim3(256), 0, st, partial, nblk1, x, A, C, stats);
          } else if (mode == 1) {
            hipLaunchKernelGGL(k_stats_last, dim3(nblk1), dim3(256), lds, st, x, C, A, C, rpb1, partial, nblk1, counter, stats2);
          } else {
            hipLaunchKernelGGL(k_stats_last, dim3(nblk2), dim3(256), lds, st, x, C, A, C, rpb2, partial, nblk2, counter, stats2);
          }
          hipLaunchKernelGGL(k_apply, dim3(ag), dim3(256), 0, st, x, y, n4, C, mode == 0 ? stats : stats2);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[mode], e0, e1));
        t[mode] *= 1e3f / reps;
      }
    }
    std::vector<float> a(2 * C), b(2 * C);
    CK(hipMemcpy(a.data(), stats, 2 * C * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), stats2, 2 * C * 4, hipMemcpyDeviceToHost));
    float md = 0.f; for (int i = 0; i < 2 * C; ++i) md = fmaxf(md, fabsf(a[i] - b[i]) / (fabsf(a[i]) + 1e-6f));
    printf("%8d %4d | %6d %6d | %11.2f | %20.2f | %19.2f | %.2e\n", A, C, nblk1, nblk2, t[0], t[1], t[2], md);
  }
  return 0;
}
