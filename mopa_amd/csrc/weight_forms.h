// Weight re-layouts / transforms of the 2D convolutions as device functions over an element range [i0, n) with stride istep, so
// that the single-weight kernels (conv2d.hip, wino2d.hip) and the batched refresh of every stale form of a network
// (k_weight_forms_batched, wino2d.hip) run the same code.
#pragma once
#include "common.h"

__device__ __forceinline__ void w4_g(const float g[3], float t[6]) {    // t = G g
  t[0] = 0.25f * g[0];
  t[1] = (-1.f / 6.f) * (g[0] + g[1] + g[2]);
  t[2] = (-1.f / 6.f) * (g[0] - g[1] + g[2]);
  t[3] = (1.f / 24.f) * g[0] + (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
  t[4] = (1.f / 24.f) * g[0] - (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
  t[5] = g[2];
}

// igemm layout [kh][kw][R][C] from the parameter layout (modes as mopa_conv2d_relayout_weight, forward direction only)
__device__ __forceinline__ void wf_relayout_body(const float* __restrict__ src, float* __restrict__ dst, int O, int I, int KH, int KW, int mode,
                                                 int64_t i0, int64_t istep) {
  const int64_t n = (int64_t)O * I * KH * KW;
  const int R = (mode == 0 || mode == 2) ? I : O, C = (mode == 0 || mode == 2) ? O : I;
  for (int64_t i = i0; i < n; i += istep) {
    const int c = (int)(i % C);
    int64_t r1 = i / C;
    const int r = (int)(r1 % R);
    r1 /= R;
    const int kw = (int)(r1 % KW), kh = (int)(r1 / KW);
    const int o = (mode == 0 || mode == 2) ? c : r, ii = (mode == 0 || mode == 2) ? r : c;
    const int64_t p = (mode < 2) ? (((int64_t)o * I + ii) * KH + kh) * KW + kw   // OIHW
                                 : (((int64_t)ii * O + o) * KH + kh) * KW + kw;  // IOHW
    dst[i] = src[p];
  }
}

// F(2x2,3x3): U[p][r][c], p = 4*i + j (k_wino_w)
__device__ __forceinline__ void wf_wino2_body(const float* __restrict__ w, int O, int I, int dgrad, float* __restrict__ U, int i0, int istep) {
  const int R = dgrad ? O : I, C = dgrad ? I : O;
  const int n = R * C;
  for (int i = i0; i < n; i += istep) {
    const int r = i / C, c = i - r * C;
    const int o = dgrad ? r : c, ci = dgrad ? c : r;
    const float* g9 = w + ((int64_t)o * I + ci) * 9;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = dgrad ? g9[(2 - a) * 3 + (2 - b)] : g9[a * 3 + b];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
      U[(int64_t)(a * 4 + 0) * n + i] = u0;
      U[(int64_t)(a * 4 + 1) * n + i] = u1;
      U[(int64_t)(a * 4 + 2) * n + i] = u2;
      U[(int64_t)(a * 4 + 3) * n + i] = u3;
    }
  }
}

// F(4x4,3x3): U[p][r][c], p = 6*i + j; transpose = 1: U^T[p][c][r] (k_wino4_w); transpose = 2: the B-operand fragments of
// k_wino4_conv, U_f[p][r / 16][c / 16][lane = 16 ((r % 16) / 4) + c % 16][r % 4] -- one 1-KiB run per (point, 16 input channels,
// 16 output channels), lane-linear, the four k steps of a lane adjacent; transpose = 3: the B-operand fragments of k_wino4_conv9 (wino4c9.hip),
// U_q[p][r / 16][c / 32][(r % 16) / 8][lane = 32 (r % 2) + c % 32][(r % 8) / 2] -- two 1-KiB runs per (point, 16 input channels, 32 output
// channels): v_mfma_f32_32x32x2_f32's lane = (k parity, column), the four k pairs of a run adjacent.
__device__ __forceinline__ void wf_wino4_body(const float* __restrict__ w, int O, int I, int dgrad, float* __restrict__ U, int transpose,
                                              int i0, int istep) {
  const int R = dgrad ? O : I, C = dgrad ? I : O;
  const int n = R * C;
  for (int i = i0; i < n; i += istep) {
    const int r = i / C, c = i - r * C;
    const int o = dgrad ? r : c, ci = dgrad ? c : r;
    const float* g9 = w + ((int64_t)o * I + ci) * 9;
    float t[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float col[3], tc[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) col[a] = dgrad ? g9[(2 - a) * 3 + (2 - b)] : g9[a * 3 + b];
      w4_g(col, tc);
#pragma unroll
      for (int a = 0; a < 6; ++a) t[a][b] = tc[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float u[6];
      w4_g(t[a], u);
      const int64_t at = transpose == 2 ? ((((int64_t)(r >> 4) * (C >> 4) + (c >> 4)) * 64 + (((r & 15) >> 2) << 4) + (c & 15)) << 2) + (r & 3)
                         : transpose == 3 ? (((((int64_t)(r >> 4) * (C >> 5) + (c >> 5)) * 2 + ((r & 15) >> 3)) * 64 + ((r & 1) << 5) + (c & 31)) << 2) + ((r & 7) >> 1)
                                          : (transpose ? c * R + r : i);
#pragma unroll
      for (int b = 0; b < 6; ++b) U[(int64_t)(a * 6 + b) * n + at] = u[b];
    }
  }
}
