// Shared helpers for libmopa_hip.so (gfx950 only; no CUDA dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define MOPA_OK 0
#define MOPA_ERR_ARG (-1)       // bad shape / unsupported channel count
#define MOPA_ERR_WORKSPACE (-2) // workspace too small
#define MOPA_ERR_LAUNCH (-3)    // hipGetLastError() != success after a launch

#define MOPA_API extern "C" __attribute__((visibility("default")))

#define MOPA_CHECK_LAUNCH()                                  \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return MOPA_ERR_LAUNCH; \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Grid size for grid-stride, HBM-bound kernels: enough blocks to fill 256 CUs x 8,
// capped so tiny inputs do not launch empty blocks (guide: Guideline 11).
static inline int stream_grid(int64_t work_items, int block) {
  int64_t g = cdiv64(work_items, block);
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

// CU count of the CURRENT device (cached per device: a process may drive several GPUs), 0 on error.
#include <atomic>
static inline int mopa_device_index() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 ? dev : -1;
}
static inline int mopa_cu_count() {
  static std::atomic<int> cus[64];
  const int dev = mopa_device_index();
  if (dev < 0) return 0;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

#define WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
