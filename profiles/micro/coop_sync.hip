// Micro-benchmark: cost of a cooperative launch + grid.sync() on gfx950 (one stream, and beside a busy second stream).
// hipcc --offload-arch=gfx950 -O3 coop_sync.hip -o coop_sync && ./coop_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__global__ void k_three_phase(float* x, float* part, float* stat, int n, int nsync) {
  cg::grid_group grid = cg::this_grid();
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  float s = 0.f;
  for (int i = tid; i < n; i += nt) s += x[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&part[blockIdx.x], s);
  for (int k = 0; k < nsync; ++k) grid.sync();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float t = 0.f;
    for (int b = 0; b < (int)gridDim.x; ++b) t += part[b];
    stat[0] = t / n;
  }
  grid.sync();
  const float m = stat[0];
  for (int i = tid; i < n; i += nt) x[i] -= m;
}
__global__ void k_plain(float* x, int n) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int i = tid; i < n; i += nt) x[i] += 1.f;
}
__global__ void k_busy(float* x, int n, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int i = tid; i < n; i += nt) { float v = x[i]; for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f; x[i] = v; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  int dev = 0, coop = 0, ncu = 0;
  CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_three_phase, 256, 0));
  printf("cooperative launch supported: %d, CUs %d, co-resident blocks of 256 per CU: %d\n", coop, ncu, per_cu);
  const int n = 1 << 20;
  float *x, *y, *part, *stat;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, (size_t)64 << 20)); CK(hipMalloc(&part, 4096 * 4)); CK(hipMalloc(&stat, 16));
  CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, (size_t)64 << 20));
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int busy = 0; busy < 2; ++busy)
    for (int grid : {64, 256, 512, 1024}) {
      for (int nsync : {0, 1, 4}) {
        int nn = n, ns = nsync;
        void* args[] = {&x, &part, &stat, &nn, &ns};
        float ms = 0.f;
        const int reps = 200;
        for (int w = 0; w < 2; ++w) {
          if (busy) hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s2, y, 16 << 20, 4000);
          CK(hipEventRecord(e0, s1));
          for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(part, 0, grid * 4, s1));
            CK(hipLaunchCooperativeKernel((void*)k_three_phase, dim3(grid), dim3(256), args, 0, s1));
          }
          CK(hipEventRecord(e1, s1));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
          CK(hipDeviceSynchronize());
        }
        printf("busy %d grid %4d extra syncs %d: %.2f us per (memset + cooperative launch)\n", busy, grid, nsync, 1e3 * ms / reps);
      }
    }
  // reference: three plain launches
  float ms = 0.f;
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, s1));
    for (int r = 0; r < 200; ++r) for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, s1, x, n);
    CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("three plain launches of 256 blocks over 4 MB: %.2f us\n", 1e3 * ms / 200);
  return 0;
}
