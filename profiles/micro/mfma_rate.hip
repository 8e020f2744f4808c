// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 from ONE wave per SIMD (256 blocks x 256 threads), 2 or 4 independent
// accumulator chains, and the shader clock under that load (clock64 vs wall_clock64).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void k(float* out, int n, long long* clk) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    if (CH == 4) {
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
    } else {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a1, 0, 0, 0);
    }
  }
  long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
  float* out; long long* clk; hipMalloc(&out, 1024 * 256 * 4); hipMallocManaged(&clk, 16);
  const int n = 20000;
  for (int blocks : {256, 512, 1024}) for (int ch : {2, 4}) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(s);
      if (ch == 2) k<2><<<blocks, 256>>>(out, n, clk); else k<4><<<blocks, 256>>>(out, n, clk);
      hipEventRecord(e); hipEventSynchronize(e);
    }
    float ms; hipEventElapsedTime(&ms, s, e);
    double per = ms * 1e6 / (4.0 * n);   // ns per MFMA per wave
    double tf = (double)blocks * 4 * 4.0 * n * 2048 / (ms * 1e-3) / 1e12;
    printf("blocks %4d chains %d: %.2f ms, %.1f ns per MFMA per wave, %.1f TFLOP/s, shader clock %.0f MHz (clock64 %lld / wall %lld @100MHz)\n", blocks, ch, ms, per, tf,
           (double)clk[0] / ((double)clk[1] / 100.0), clk[0], clk[1]);
  }
  return 0;
}
