// F(4x4,3x3) Winograd transform arithmetic shared by wino2d.hip (transforms, one-kernel convolutions) and wino4wg.hip (the
// one-kernel weight gradient): Lavin & Gray's matrices, every multiply-add of B^T d written as ONE fused operation so that every
// instantiation rounds alike ("V has the same bits on every path" is a tested property).
#pragma once
#include "common.h"

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float w4_fma(float a, float x, float y) { return fmaf(a, x, y); }
__device__ __forceinline__ f32x2v w4_fma(float a, f32x2v x, f32x2v y) { return __builtin_elementwise_fma((f32x2v)(a), x, y); }
__device__ __forceinline__ f32x4v w4_fma(float a, f32x4v x, f32x4v y) { return __builtin_elementwise_fma((f32x4v)(a), x, y); }
template <class T>
__device__ __forceinline__ void w4_bt6(T d0, T d1, T d2, T d3, T d4, T d5, T& t0, T& t1, T& t2, T& t3, T& t4, T& t5) {
  const T e42 = d4 - d2;
  t0 = w4_fma(4.f, d0, w4_fma(-5.f, d2, d4));
  t1 = w4_fma(-4.f, d1 + d2, d3 + d4);
  t2 = w4_fma(4.f, d1 - d2, d4 - d3);
  t3 = w4_fma(2.f, d3 - d1, e42);
  t4 = w4_fma(2.f, d1 - d3, e42);
  t5 = w4_fma(4.f, d1, w4_fma(-5.f, d3, d5));
}
__device__ __forceinline__ void w4_bt(const float d[6], float t[6]) { w4_bt6(d[0], d[1], d[2], d[3], d[4], d[5], t[0], t[1], t[2], t[3], t[4], t[5]); }
__device__ __forceinline__ void w4_at(const float m[6], float y[4]) {   // y = A^T m
  const float a = m[1] + m[2], b = m[1] - m[2], c = m[3] + m[4], e = m[3] - m[4];
  y[0] = m[0] + a + c;
  y[1] = b + 2.f * e;
  y[2] = a + 4.f * c;
  y[3] = b + 8.f * e + m[5];
}
__device__ __forceinline__ void w4_a(const float d[4], float r[6]) {    // r = A d  (A = (A^T)^T, 6x4)
  r[0] = d[0];
  r[1] = d[0] + d[1] + d[2] + d[3];
  r[2] = d[0] - d[1] + d[2] - d[3];
  r[3] = d[0] + 2.f * d[1] + 4.f * d[2] + 8.f * d[3];
  r[4] = d[0] - 2.f * d[1] + 4.f * d[2] - 8.f * d[3];
  r[5] = d[3];
}
__device__ __forceinline__ void w4_gt(const float u[6], float g[3]) {   // g = G^T u
  g[0] = 0.25f * u[0] - (1.f / 6.f) * (u[1] + u[2]) + (1.f / 24.f) * (u[3] + u[4]);
  g[1] = (1.f / 6.f) * (u[2] - u[1]) + (1.f / 12.f) * (u[3] - u[4]);
  g[2] = -(1.f / 6.f) * (u[1] + u[2]) + (1.f / 6.f) * (u[3] + u[4]) + u[5];
}


// conv2d.hip: slabs [nsplit][36][Cin][Cout] summed in split order, then dW = G^T dU G written (flags bit 0: accumulated) into the igemm
// layout or (flags bit 1) torch's OIHW tensor
int wino4_dw_launch(const float* slabs, int nsplit, int Cin, int Cout, float* dweight, int flags, hipStream_t st);

// wgemm.hip: the transform-domain weight gradient dU[p] = V[p]^T dM[p] of the 128-aligned layers as a ring-buffered LDS-DMA GEMM
bool wgemm_tn_ok(int np, int64_t T, int Cin, int Cout);
void wgemm_tn_split(int np, int64_t T, int Cin, int Cout, int* nsplit, int* k_per_split);
int wgemm_tn_launch(int np, const float* V, const float* dM, int T, int Cin, int Cout, float* slabs, int nsplit, int k_per_split, hipStream_t st);
