// Internal interface of the persistent sparse-convolution kernel (spconv_ring.hip) towards the dispatcher in spconv.hip.
#pragma once
#include "common.h"

// Column-group width (16-column tiles per block: 1 or 2) the ring kernel wants for this shape, or 0 when the shape is not its.
// cin / cout are those of the convolution to run (already swapped for backward-data).
int mopa_ring_plan(int K, int64_t num_out, int cin, int cout);

// Runs the convolution on the grouped rulebook with weights packed for column groups of `ntw` tiles (mopa_spconv_pack_weight).
int mopa_ring_launch(const int* gs, const int* go, const int* gi, const int* gout, int K, int num_out, const float* in, int ld_in,
                     int cin, const float* Wp, int cout, int w_flip, float* out, int ld_out, int ntw, hipStream_t st);
