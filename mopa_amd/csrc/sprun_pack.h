// Shared by sprun.hip and spconv.hip (the batched weight-form refresh): the weight layout of k_spconv_run.
#pragma once
#include <stdint.h>
// Weights for k_spconv_run: per (offset, column group of NT 16-column tiles) one contiguous slice in LDS-image order,
//   wr[o][cg][kc][t][lane][s] = Wc[o][16 kc + 4 (lane >> 4) + s][(cg NT + t) 16 + (lane & 15)]
// (Wc = w [K][cin][cout], or its per-offset transpose for backward-data): a lane's MFMA operand of (chunk kc, tile t) is one
// conflict-free ds_read_b128 at (kc NT + t) 1024 + 16 lane, and staging a slice is a straight copy.
__device__ __forceinline__ void run_pack_elem(const float* __restrict__ w, float* __restrict__ wr, int i, int K, int cin_w, int cout_w,
                                              int transpose, int nt) {
  const int cin_c = transpose ? cout_w : cin_w, cout_c = transpose ? cin_w : cout_w;
  const int nkc = cin_c >> 4, ncg = (cout_c >> 4) / nt;
  int rem = i;
  const int s = rem & 3; rem >>= 2;
  const int lane = rem & 63; rem >>= 6;
  const int t = rem % nt; rem /= nt;
  const int kc = rem % nkc; rem /= nkc;
  const int cg = rem % ncg;
  const int o = rem / ncg;
  const int k = kc * 16 + (lane >> 4) * 4 + s, c = (cg * nt + t) * 16 + (lane & 15);
  wr[i] = transpose ? w[((int64_t)o * cin_w + c) * cout_w + k] : w[((int64_t)o * cin_w + k) * cout_w + c];
}
// Column-group width (16-column tiles per block) of the run kernel for cout output channels: all of them up to 7 tiles,
// else the largest divisor <= 7 (128 -> 4, 160 -> 5, 192 -> 6, 224 -> 7, 144 -> 3).  0 = shape not supported.
static inline int run_nt(int cin, int cout) {
  if (cin % 16 || cout % 16 || cin < 16 || cin > 224 || cout < 16 || cout > 224) return 0;
  const int tiles = cout / 16;
  if (tiles <= 7) return tiles;
  for (int c = 7; c >= 2; --c)
    if (tiles % c == 0) return c;
  return 0;
}
