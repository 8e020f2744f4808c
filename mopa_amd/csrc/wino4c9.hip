// A whole F(4x4,3x3) convolution -- input transform, 36 GEMMs, output transform -- in one kernel, second form (the first: k_wino4_conv /
// k_wino4_conv32 in wino2d.hip; same contract: torch.nn.Conv2d(3x3, stride 1, padding 1) of /root/reference/mopa/models/resnet34_unet.py:
// 97-110 and its backward-data).  Neither V nor M reaches HBM.
//
// Why a second form: the first gives a wave ALL 36 transform points of a 16-tile x 16-channel block (v_mfma_f32_16x16x4_f32, 144
// accumulators), so a 1-KiB weight fragment from L2 feeds four 32-cycle MFMAs -- the CU's vector-memory front end takes ~16 cycles per
// wave instruction whatever its width (measured in wino4wg.hip), 108 of them per 144 MFMAs: it is bound there at about a third of
// the matrix peak.  Here a wave owns NINE points -- a 3 x 3 block of the 6 x 6 transform grid -- of a 32-tile x 32-channel block
// (v_mfma_f32_32x32x2_f32, the same 144 accumulators): a 1-KiB fragment feeds four 64-cycle MFMAs on twice the tiles, the weights
// of a block are read once (each wave reads only its own points'), 18 + the patch loads per 72 MFMAs of 64 cycles.
//
// Workgroup = 8 waves = (3 x 3 point block g, output-channel half h), one per CU (77 KB of LDS); item = 32 consecutive tiles x 64 output
// channels; input channels in steps of 16:
//   transform  thread (tile = t / 16, channel = t % 16) holds the 6x6 patch of its (tile, channel) -- loaded during the multiplication of
//              the step before, four loads behind each point's MFMAs; zero outside the image; a deferred BatchNorm + ReLU on the way
//              in as mopa_wino4_input_bn -- transforms it (12 operations per 6-point transform) and writes the 36 points into
//              Vs[p][channel][tile] (the MFMA A operand: lane = (channel parity, tile));
//   multiply   wave (g, h), point q of its nine: 8 ds_read_b32 + 2 16-byte weight loads (fragment order: mopa_wino4_weight_q) + 8 MFMAs.
// After the last step every wave applies the output transform to ITS 3 x 3 block (Y = A^T M A is a sum over the four blocks) and the
// four partial 4x4 outputs meet in LDS, two output pixels at a time; the summing pass adds the bias and stores whole 256-byte
// pixel rows.
#include "wino4.h"
#include <stdlib.h>
#include <stdio.h>

typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef float f32x4c __attribute__((ext_vector_type(4)));

#define C9_TM 32        // tiles per item
#define C9_CN 64        // output channels per item (two halves of 32: one MFMA column block per wave)
#define C9_KC 16        // input channels per step
#define C9_PITCH 34     // floats per (point, channel) row of Vs: 32 tiles + 2 (the 16 channels of a tile write different banks pairwise)

struct C9Args {
  const float* in; const float* Uq; const float* bias; float* out; const float* stats;
  int ld_in, ld_out, B, H, W, th, tw, T, Cin, Cout, accumulate, imgs_per_group, bn_c0, nco;
  long long* prof;
};

// 6-point input transform t = B^T d in 12 operations (wino4wg.hip's; the values of w4_bt6 up to rounding)
__device__ __forceinline__ void c9_bt(float d0, float d1, float d2, float d3, float d4, float d5, float& t0, float& t1, float& t2, float& t3,
                                      float& t4, float& t5) {
  t0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
  const float p = fmaf(-4.f, d2, d4), q = fmaf(-4.f, d1, d3);
  t1 = p + q;
  t2 = p - q;
  const float r = d4 - d2, s = d3 - d1;
  t3 = fmaf(2.f, s, r);
  t4 = fmaf(-2.f, s, r);
  t5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}

template <bool BN>
__global__ __launch_bounds__(512, 2) void k_wino4_conv9(const C9Args a) {
  __shared__ __attribute__((aligned(16))) float c9_lds[36 * C9_KC * C9_PITCH + 36 * 512];
  float* const Vs = c9_lds;                            // 78,336 B: the transformed patches [point][channel][tile]
  float* const Ps = c9_lds + 36 * C9_KC * C9_PITCH;    // 73,728 B: the next step's raw patches, [element][thread]
  // (the output exchange of the epilogue, 96 KB, goes over both)
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  // blockIdx.x -> (tile group, output-channel block): the nco blocks of a tile group on ONE XCD (ids equal modulo 8 share an L2)
  const int nco = a.nco;
  const int grp = blockIdx.x / (8 * nco), r8 = blockIdx.x - grp * 8 * nco;
  const int cob = r8 >> 3, tg = grp * 8 + (r8 & 7);
  const int t0 = tg * C9_TM;
  if (t0 >= a.T) return;
  const int co0 = cob * C9_CN;
  const int H = a.H, W = a.W, tw = a.tw, thw = a.th * a.tw;
#ifdef C9_PROFILE
  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define C9_T(K_) { const long long n_ = __builtin_readcyclecounter(); pt[K_] += n_ - tprev; tprev = n_; }
#else
#define C9_T(K_)
#endif

  // ---- transform role: tile slot = t / 16, channel of the step = t % 16
  const int ts = t >> 4, ch = t & 15;
  const int tile = t0 + ts;
  const bool tv = tile < a.T;
  const int tb = tile / thw, rt = tile - tb * thw, ty = rt / tw, tx = rt - ty * tw;
  // validity of the patch rows / columns (bits 0-5 / 8-13) -- applied when the patch is transformed; loads are unconditional, from
  // CLAMPED pixels where the patch leaves the image (column offsets kept, row offsets recomputed per patch row: 8 registers, no branch)
  const int ab = tv ? tb : 0, ay = tv ? ty : 0, ax = tv ? tx : 0;
  const uint32_t ld4 = (uint32_t)a.ld_in * 4u;
  unsigned okm = 0;
  uint32_t cof[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    okm |= (tv && (unsigned)(4 * ay - 1 + i) < (unsigned)H ? 1u : 0u) << i;
    okm |= ((unsigned)(4 * ax - 1 + i) < (unsigned)W ? 1u : 0u) << (8 + i);
    cof[i] = (uint32_t)min(max(4 * ax - 1 + i, 0), W - 1) * ld4 + (uint32_t)ch * 4u;
  }
  const int abH = ab * H, y0 = 4 * ay - 1;
  const bool interior = __all(tv && ty >= 1 && tx >= 1 && 4 * ty + 4 < H && 4 * tx + 4 < W) != 0;   // wave-uniform: no zeroing needed
  const char* __restrict__ inb = reinterpret_cast<const char*>(a.in);
  float* __restrict__ vw = Vs + ch * C9_PITCH + ts;   // + p * 16 * PITCH
  float sc = 1.f, sh = 0.f;
  bool bn_ch = false;

  // ---- multiply role: wave = (3 x 3 point block g = wv % 4: row block g / 2, column block g % 2; output-channel half h = wv / 4)
  const int l32 = lane & 31, lk = lane >> 5, g = wv & 3, coh = wv >> 2;
  const int pa0 = 3 * (g >> 1), pb0 = 3 * (g & 1);
  const int nci = a.Cin / C9_KC;
  const int64_t pstride = (int64_t)a.Cin * a.Cout;   // floats per point of Uq
  const int nco32 = a.Cout / 32;
  // weights: a wave-uniform base (scalar registers) + the lane's 16 bytes as a 32-bit offset: + cib * nco32 * 512 + half * 256 floats
  const char* __restrict__ ub = reinterpret_cast<const char*>(a.Uq + (int64_t)(pa0 * 6 + pb0) * pstride + (int64_t)(cob * 2 + coh) * 512);
  const uint32_t ul = (uint32_t)lane * 16u;

  f32x16c acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

  // The raw patch of the NEXT step goes global -> LDS by LDS-DMA (global_load_lds_dword: no destination registers -- 144 accumulators
  // + 36 patch values + operands do not fit 256 registers), element n of thread t at Ps[n][t]: a wave writes and reads its own 64
  // columns, so its own vmcnt orders it (the compiler puts that wait in front of the first LDS read it can see after a DMA -- which is
  // why the multiplication's operand reads below are inline asm: they must not wait for the patches).
  auto row_off = [&](const int i) -> uint32_t { return (uint32_t)((abH + min(max(y0 + i, 0), H - 1)) * W) * ld4; };
  const unsigned ps0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Ps[0] + (unsigned)wv * 256u;   // + n * 2048
  auto patch_dma = [&](const int n, const uint32_t cb) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(inb + (row_off(n / 6) + cof[n % 6] + cb)),
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(ps0 + (unsigned)n * 2048u), 4, 0, 0);
  };
  auto patch_bn = [&](int cib) {
    if (BN) {
      const int c_ = cib * C9_KC + ch;
      bn_ch = c_ >= a.bn_c0;
      if (bn_ch) {
        const float* __restrict__ sg = a.stats + (int64_t)((tv ? tb : 0) / a.imgs_per_group) * 4 * (a.Cin - a.bn_c0) + (c_ - a.bn_c0);
        sc = sg[0];
        sh = sg[a.Cin - a.bn_c0];
      }
    }
  };
#pragma unroll
  for (int n = 0; n < 36; ++n) patch_dma(n, 0u);
  patch_bn(0);
  const unsigned vr_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Vs[0] +
                        (unsigned)(((pa0 * 6 + pb0) * C9_KC + lk) * C9_PITCH + l32) * 4u;   // + ((i * 6 + j) * 16 + 2 kk) * PITCH * 4

  for (int cib = 0; cib < nci; ++cib) {
    C9_T(0)
    // ---- transform the patch of (tile, channel 16 cib + ch) that the step before loaded
    {
      // (the compiler does NOT order an LDS read behind an LDS-DMA of the same wave: without this wait the first step read Ps before its
      //  patches had landed -- right on a few blocks, wrong under load)
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
      float d[36];
#pragma unroll
      for (int n = 0; n < 36; ++n) d[n] = Ps[n * 512 + t];
      if (BN || !interior) {
#pragma unroll
        for (int n = 0; n < 36; ++n) {
          float v = d[n];
          if (BN) {
            const float o = fmaf(v, sc, sh);
            v = bn_ch ? (o > 0.f ? o : o * 0.f) : v;
          }
          d[n] = ((okm >> (n / 6)) & (okm >> (8 + n % 6)) & 1u) ? v : 0.f;
        }
      }
      C9_T(1)
      float m[6][6];
#pragma unroll
      for (int j = 0; j < 6; ++j)   // B^T d, column by column
        c9_bt(d[j], d[6 + j], d[12 + j], d[18 + j], d[24 + j], d[30 + j], m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]);
#pragma unroll
      for (int i = 0; i < 6; ++i) {   // (.) B
        float v[6];
        c9_bt(m[i][0], m[i][1], m[i][2], m[i][3], m[i][4], m[i][5], v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
        for (int j = 0; j < 6; ++j) vw[((i * 6 + j) * C9_KC) * C9_PITCH] = v[j];
      }
    }
    C9_T(2)
    __syncthreads();   // V of this step is complete
    C9_T(3)
    // ---- multiply: 9 points x 8 channel pairs; the next step's patch loads go out four behind each point
    {
      // (unconditional: the last step fetches its own patch again -- a branch around the DMAs makes the compiler wait for vmcnt(0) in
      //  front of every point's weights)
      const bool more = cib + 1 < nci;
      const uint32_t cb = (uint32_t)((more ? cib + 1 : cib) * C9_KC) * 4u;
      // Weights by hand (inline asm loads + counted waits): with LDS-DMA instructions in flight the compiler waits for vmcnt(0) before
      // every use of a loaded register, i.e. for every patch DMA issued so far -- the memory latency nine times per step.
      const char* __restrict__ up = ub + (int64_t)cib * nco32 * 2048;
      f32x4c b0, b1, n0, n1;
#define C9_WLOAD(X0, X1, PTR) \
  asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024" : "=&v"(X0), "=&v"(X1) : "v"(ul), "s"(PTR))
#define C9_WWAIT(N, X0, X1) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(X0), "+v"(X1))
      C9_WLOAD(b0, b1, up);
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (q + 1 < 9) {
          const char* __restrict__ un = up + (int64_t)(((q + 1) / 3) * 6 + (q + 1) % 3) * pstride * 4;
          if (q & 1) { C9_WLOAD(b0, b1, un); } else { C9_WLOAD(n0, n1, un); }
        }
        float av[8];
#define C9_OFF(KK) "i"((((q / 3) * 6 + q % 3) * C9_KC + 2 * (KK)) * C9_PITCH * 4)
        asm volatile("ds_read_b32 %0, %8 offset:%9\n\tds_read_b32 %1, %8 offset:%10\n\tds_read_b32 %2, %8 offset:%11\n\tds_read_b32 %3, %8 offset:%12\n\t"
                     "ds_read_b32 %4, %8 offset:%13\n\tds_read_b32 %5, %8 offset:%14\n\tds_read_b32 %6, %8 offset:%15\n\tds_read_b32 %7, %8 offset:%16\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(av[0]), "=&v"(av[1]), "=&v"(av[2]), "=&v"(av[3]), "=&v"(av[4]), "=&v"(av[5]), "=&v"(av[6]), "=&v"(av[7])
                     : "v"(vr_b), C9_OFF(0), C9_OFF(1), C9_OFF(2), C9_OFF(3), C9_OFF(4), C9_OFF(5), C9_OFF(6), C9_OFF(7));
        // in order after the two loads of point q: the six patch DMAs of point q - 1 (points 0-5 issue them: the last ones have three
        // points' MFMAs to land behind) and the two loads of point q + 1
        if (q & 1) {   // the odd points' weights sit in (n0, n1)
          if (q == 7) { C9_WWAIT(2, n0, n1); } else { C9_WWAIT(8, n0, n1); }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], n0[kk], acc[q], 0, 0, 0);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 + kk], n1[kk], acc[q], 0, 0, 0);
        } else {
          if (q == 0) { C9_WWAIT(2, b0, b1); } else if (q == 8) { C9_WWAIT(0, b0, b1); } else { C9_WWAIT(8, b0, b1); }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc[q], 0, 0, 0);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 + kk], b1[kk], acc[q], 0, 0, 0);
        }
        if (q < 6) {
#pragma unroll
          for (int n = 6 * q; n < 6 * q + 6; ++n) patch_dma(n, cb);
        }
      }
      if (more) patch_bn(cib + 1);
    }
    C9_T(4)
    // everyone is done reading V (every operand read above was waited for: a RAW barrier -- __syncthreads()' fence would also wait for
    // the patch DMAs in flight, whose wait belongs to the top of the next step)
    __builtin_amdgcn_s_barrier();
    C9_T(5)
  }

  // ---- output transform.  Y = A^T M A: wave (row block ra, column block cb) applies the ROW half to its 3 x 3 block,
  //   Tg[i][b] = sum_{a in rows of ra} AT[i][a] M[a][b]        (4 x 3 values per (tile, channel)),
  // the four partials meet in LDS one output row i at a time, and the summing pass adds the two row blocks and applies the COLUMN half,
  //   Y[i][j] = sum_{b = 0..5} AT[j][b] (T0[i][b] + T1[i][b]):   12 LDS values in, 4 pixels out per (i, tile, channel).
  // (accumulator register e of a 32x32 tile = row (tile) 8 (e / 4) + 4 lk + e % 4, column (channel) l32)
  constexpr float AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -2.f, 0.f}, {0.f, 1.f, 1.f, 4.f, 4.f, 0.f},
                              {0.f, 1.f, -1.f, 8.f, -8.f, 1.f}};
  float ar_[4][3];   // AT[i][3 ra + a] (wave-uniform selects of a compile-time table)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) ar_[i][k] = (g >> 1) ? AT[i][3 + k] : AT[i][k];
  // Ts[row block ra][b 0..5][tile 32][channel 64] = 96 KB over Vs | Ps (both free now: the last step's patch DMAs have landed), one
  // output row i at a time.  Summing pass: thread -> (tile t / 16, channel quad t % 16): 12 16-byte LDS reads, four 16-byte pixel stores.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's last patch DMAs (a re-fetch nobody reads) are not in flight any more
  __syncthreads();
  float* __restrict__ tsb = c9_lds;
  const int stl = t0 + (t >> 4), sq = (t & 15) * 4;
  const f32x4c bv = a.bias ? *reinterpret_cast<const f32x4c*>(a.bias + co0 + sq) : (f32x4c){0.f, 0.f, 0.f, 0.f};
  const int sb_ = stl / thw, sr_ = stl - sb_ * thw, sty = sr_ / tw, stx = sr_ - sty * tw;
  float* __restrict__ obase = a.out + ((int64_t)(sb_ * H + 4 * sty) * W + 4 * stx) * a.ld_out + co0 + sq;
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int til = 8 * (e >> 2) + 4 * lk + (e & 3);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float tv_ = fmaf(ar_[i][0], acc[b][e], fmaf(ar_[i][1], acc[3 + b][e], ar_[i][2] * acc[6 + b][e]));
        tsb[(((g >> 1) * 6 + pb0 + b) * C9_TM + til) * C9_CN + coh * 32 + l32] = tv_;
      }
    }
    __syncthreads();
    {
      f32x4c tt[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const f32x4c u0 = *reinterpret_cast<const f32x4c*>(tsb + (b * C9_TM + (t >> 4)) * C9_CN + sq);
        const f32x4c u1 = *reinterpret_cast<const f32x4c*>(tsb + ((6 + b) * C9_TM + (t >> 4)) * C9_CN + sq);
        tt[b] = u0 + u1;
      }
      if (stl < a.T && 4 * sty + i < H) {
        float* __restrict__ o = obase + (int64_t)i * W * a.ld_out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4c y = bv;
#pragma unroll
          for (int b = 0; b < 6; ++b)
            if (AT[j][b] != 0.f) y += AT[j][b] * tt[b];
          if (4 * stx + j < W) {
            f32x4c* __restrict__ op = reinterpret_cast<f32x4c*>(o + (int64_t)j * a.ld_out);
            *op = a.accumulate ? *op + y : y;
          }
        }
      }
    }
    __syncthreads();
  }
#ifdef C9_PROFILE
  C9_T(6)
  if (blockIdx.x == 100 && t == 0)
    for (int k = 0; k < 8; ++k) a.prof[k] = pt[k];
#endif
}

// (mopa_wino4_weight_q, the B-operand fragments of this kernel, lives in wino2d.hip beside the other weight forms: one translation unit for
//  the single-form kernel and the batched refresh keeps the two bit-identical.)

// out (NHWC, row stride ld_out) = conv3x3(in) (+ bias) (+= if accumulate) through F(4x4,3x3) in one kernel, nine transform points per wave.
// Uq: mopa_wino4_weight_q.  Cin % 16 == 0, Cout % 64 == 0.  stats / n_groups / bn_c0 as mopa_wino4_conv.
MOPA_API int mopa_wino4_conv9(const float* in, int32_t ld_in, const float* Uq, const float* bias, float* out, int32_t ld_out, int32_t B, int32_t H,
                              int32_t W, int32_t Cin, int32_t Cout, int32_t accumulate, const float* stats, int32_t n_groups, int32_t bn_c0,
                              void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % C9_KC || Cout % C9_CN || ld_in < Cin || ld_out < Cout ||
      ((uintptr_t)Uq & 15) || ((uintptr_t)out & 15) || (ld_out & 3) || ((uintptr_t)bias & 15) || !in || !Uq || !out)
    return MOPA_ERR_ARG;
  if (stats && (n_groups < 1 || B % n_groups || bn_c0 < 0 || bn_c0 >= Cin)) return MOPA_ERR_ARG;
  if ((int64_t)B * H * W * ld_in >= (1ll << 30) || (int64_t)B * H * W * ld_out >= (1ll << 31)) return MOPA_ERR_ARG;   // 32-bit offsets
  C9Args a;
  a.in = in; a.Uq = Uq; a.bias = bias; a.out = out; a.stats = stats;
  a.ld_in = ld_in; a.ld_out = ld_out; a.B = B; a.H = H; a.W = W; a.th = (H + 3) / 4; a.tw = (W + 3) / 4;
  const int64_t T = (int64_t)B * a.th * a.tw;
  if (T >= (1 << 30)) return MOPA_ERR_ARG;
  a.T = (int)T; a.Cin = Cin; a.Cout = Cout; a.accumulate = accumulate;
  a.imgs_per_group = stats ? B / n_groups : 1;
  a.bn_c0 = stats ? bn_c0 : 0;
  a.nco = Cout / C9_CN;
  a.prof = nullptr;
  const int64_t ntg = (cdiv64(T, C9_TM) + 7) / 8 * 8;   // tile groups, padded to whole XCD rounds (blocks beyond T return at once)
  const int64_t nblk = ntg * a.nco;
  if (nblk >= (1ll << 31)) return MOPA_ERR_ARG;
#ifdef C9_PROFILE
  static long long* prof = nullptr;
  if (!prof) hipMallocManaged(&prof, 128);
  a.prof = prof;
#endif
  if (stats) k_wino4_conv9<true><<<(unsigned)nblk, 512, 0, (hipStream_t)stream>>>(a);
  else k_wino4_conv9<false><<<(unsigned)nblk, 512, 0, (hipStream_t)stream>>>(a);
  MOPA_CHECK_LAUNCH();
#ifdef C9_PROFILE
  hipStreamSynchronize((hipStream_t)stream);
  {
    const double n_ = Cin / C9_KC;
    printf("[c9 profile] block 100 wave 0, cycles per step: loop head %.0f | patch loads + apply %.0f | transform + lds stores %.0f | barrier %.0f | "
           "multiply %.0f | barrier %.0f | output transform + exchange + store (per item) %.0f   (steps per item %.0f)\n",
           prof[0] / n_, prof[1] / n_, prof[2] / n_, prof[3] / n_, prof[4] / n_, prof[5] / n_, (double)prof[6], n_);
  }
#endif
  return MOPA_OK;
}
