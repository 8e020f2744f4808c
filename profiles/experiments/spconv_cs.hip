// Sparse 3D convolution, "column-slice" kernel for the layers that carry the family's flops (Cin >= 48): same contract
// as mopa_spconv_fwd_grouped (out[i] = sum_o in[nbr[o][i]] @ W[o], forward and backward-data), different tiling.
//
// Why (round 3): k_spconv_t4 loads the 16 x 16 weight operand of every MFMA from L2 -- a B element
// feeds exactly one MFMA (16 rules), and with 1-2 column tiles per wave the gathered rows are re-used once or twice.  At the
// f32 MFMA rate that is 48-64 B/clk/CU of operand traffic through a 64 B/clk vector L1: the kernel sits at 40-54 % of the
// matrix pipe with the pipe, the L1 and the issue port all half busy.  Here
//   * a tile is TM = 128 or 256 output rows, so the rules of one (tile, filter offset) fill whole 16-rule groups (79-91 %
//     instead of 49-78 %: 1.2-1.6x fewer MFMAs) and an offset owns a RUN of 2-5 consecutive groups;
//   * wave w of the block owns the 16-column slice w of the tile's accumulator for ALL groups: private LDS columns, so no
//     atomics and no cross-wave sum, each output element is written once, in a fixed summation order;
//   * the weight slice W[o][:, 16w:16w+16] of a run lives in REGISTERS (one packed 1 KiB load per 16 input channels and
//     run, prefetched one stage ahead) and is re-used by every group of the run;
//   * the gathered input rows come through a per-wave register ring, as in k_spconv_t4; the block's waves (the column slices
//     of one tile) ask for the same rows at about the same time, so the repeats are vector-L1 hits.
// No barrier in the main loop.
//
// Reference semantics: sparseconvnet SubmanifoldConvolution / Convolution / Deconvolution behind
// mopa/models/scn_unet.py:27-28; oracle: oracle/scn3d.py::sparse_conv.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <atomic>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------------
// Grouped rulebook over TM-row tiles (TM = 64 * NSUB), all tables of a geometry in one launch each.
//   desc[t] = {table pointer, K, rows, first global tile};  tile t owns groups grp_start[t] .. grp_start[t+1]-1, ordered
//   by filter offset;  grp_o[g] = offset | (groups left in this (tile, offset) run, this one included) << 8;
//   grp_in[g][16] = input rows (-1 = padding);  grp_out[g][16] = output row within the tile (0 .. TM-1, -1 = padding).
#define CS_MAX_TABLES 32
struct CsDescs { int64_t v[CS_MAX_TABLES * 4]; };
__device__ __forceinline__ int cs_find_table(const CsDescs& d, int ntables, int tile) {
  int t = 0;
  for (int k = 1; k < ntables; ++k)
    if (tile >= (int)d.v[k * 4 + 3]) t = k;
  return t;
}

template <int NSUB>
__global__ __launch_bounds__(64) void k_cs_rb_count(const CsDescs desc, int ntables, int* __restrict__ tile_groups) {
  const int t = cs_find_table(desc, ntables, blockIdx.x);
  const int* __restrict__ nbr = reinterpret_cast<const int*>(desc.v[t * 4]);
  const int K = (int)desc.v[t * 4 + 1], A_out = (int)desc.v[t * 4 + 2];
  const int row0 = (blockIdx.x - (int)desc.v[t * 4 + 3]) * 64 * NSUB + threadIdx.x;
  int ng = 0;
  for (int o = 0; o < K; ++o) {
    int n = 0;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const int row = row0 + 64 * s;
      const int nb = (row < A_out) ? nbr[(int64_t)o * A_out + row] : -1;
      n += __popcll(__ballot(nb >= 0));
    }
    ng += (n + 15) >> 4;
  }
  if (threadIdx.x == 0) tile_groups[blockIdx.x] = ng;
}

template <int NSUB>
__global__ __launch_bounds__(64) void k_cs_rb_fill(const CsDescs desc, int ntables, const int* __restrict__ grp_start,
                                                    int* __restrict__ grp_o, int* __restrict__ grp_in, int* __restrict__ grp_out) {
  const int t = cs_find_table(desc, ntables, blockIdx.x);
  const int* __restrict__ nbr = reinterpret_cast<const int*>(desc.v[t * 4]);
  const int K = (int)desc.v[t * 4 + 1], A_out = (int)desc.v[t * 4 + 2];
  const int lane = threadIdx.x;
  const int row0 = (blockIdx.x - (int)desc.v[t * 4 + 3]) * 64 * NSUB + lane;
  int g = grp_start[blockIdx.x];
  for (int o = 0; o < K; ++o) {
    int nb[NSUB];
    int n = 0;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const int row = row0 + 64 * s;
      nb[s] = (row < A_out) ? nbr[(int64_t)o * A_out + row] : -1;
      const unsigned long long bal = __ballot(nb[s] >= 0);
      const int pos = n + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
      if (nb[s] >= 0) { grp_in[(int64_t)g * 16 + pos] = nb[s]; grp_out[(int64_t)g * 16 + pos] = 64 * s + lane; }
      n += __popcll(bal);
    }
    if (n == 0) continue;
    const int ng = (n + 15) >> 4;
    if (lane < ng * 16 - n) { grp_in[(int64_t)g * 16 + n + lane] = -1; grp_out[(int64_t)g * 16 + n + lane] = -1; }
    for (int j = lane; j < ng; j += 64) grp_o[g + j] = o | ((ng - j) << 8);
    g += ng;
  }
}

// Upper bound of the group count of one table (no host round trip needed to size the arrays): every (tile, offset) run
// wastes less than one group, so groups <= rules / 16 + tiles * K <= K * (rows / 16 + tiles) .
MOPA_API size_t mopa_rulebook_cs_group_bound(int32_t K, int32_t num_out, int32_t tile_rows) {
  if (K <= 0 || num_out <= 0 || tile_rows <= 0) return 0;
  return (size_t)K * (size_t)(cdiv64(num_out, 16) + cdiv64(num_out, tile_rows));
}

MOPA_API int mopa_rulebook_cs_count(const int64_t* desc_host, int32_t ntables, int32_t total_tiles, int32_t tile_rows,
                                    int32_t* tile_groups, void* stream) {
  if (ntables <= 0 || ntables > CS_MAX_TABLES || total_tiles <= 0 || (tile_rows != 128 && tile_rows != 256)) return MOPA_ERR_ARG;
  CsDescs desc;
  memcpy(desc.v, desc_host, (size_t)ntables * 4 * sizeof(int64_t));
  if (tile_rows == 128) k_cs_rb_count<2><<<total_tiles, 64, 0, (hipStream_t)stream>>>(desc, ntables, tile_groups);
  else k_cs_rb_count<4><<<total_tiles, 64, 0, (hipStream_t)stream>>>(desc, ntables, tile_groups);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
// grp_start: exclusive scan of tile_groups over ALL tiles (+ the total at [total_tiles]); grp_o / grp_in / grp_out: shared arrays
// sized by the sum of mopa_rulebook_cs_group_bound over the tables.
MOPA_API int mopa_rulebook_cs_fill(const int64_t* desc_host, int32_t ntables, int32_t total_tiles, int32_t tile_rows,
                                   const int32_t* grp_start, int32_t* grp_o, int32_t* grp_in, int32_t* grp_out, void* stream) {
  if (ntables <= 0 || ntables > CS_MAX_TABLES || total_tiles <= 0 || (tile_rows != 128 && tile_rows != 256)) return MOPA_ERR_ARG;
  CsDescs desc;
  memcpy(desc.v, desc_host, (size_t)ntables * 4 * sizeof(int64_t));
  if (tile_rows == 128) k_cs_rb_fill<2><<<total_tiles, 64, 0, (hipStream_t)stream>>>(desc, ntables, grp_start, grp_o, grp_in, grp_out);
  else k_cs_rb_fill<4><<<total_tiles, 64, 0, (hipStream_t)stream>>>(desc, ntables, grp_start, grp_o, grp_in, grp_out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ----------------------------------------------------------------------------------------------
// The convolution.  NKC = Cin / 16, NTW = 16-column tiles per wave (1 or 2), D = ring depth (groups of row gathers in flight
// per wave); waves per block = column slices of 16 NTW columns per block (run time).
//   lane l: r = l & 15, q = l >> 4.  A fragment of (group, chunk kk): lane holds in[rule r][16 kk + 4 q + s], s < 4.
//   B fragment of (offset, chunk kk, tile t): lane holds Wc[o][16 kk + 4 q + s][16 (NTW ct + t) + r]  (packed by
//   mopa_spconv_pack_weight, ntw = 1: one contiguous 1 KiB piece per (column tile, offset, chunk)).
//   D tile t: lane holds rules 4 q + j, column 16 t + r of the wave's slice.
//
// What bounds k_spconv_t4 (profiles/r3_spconv_tcp_pmc.md): the vector L1 (TCP) is busy 82-87 % of the kernel's duration on the
// layers that carry the bytes -- a 16-row gather touches 16 cache lines per KiB, a packed weight piece 8, and with the
// weight operand re-loaded for every 16-rule group that is 4 line accesses per MFMA at two column tiles per wave.  Here the
// weight slice of a RUN (the consecutive groups of one filter offset: 2-5 with 128 / 256-row tiles) is loaded ONCE.
//
// Every wave walks ALL groups of the tile on its own (no barrier in the main loop: a barrier-synchronous version with the
// gathered rows shared through LDS paid a memory round trip per 2-3 group stage and ran 2-4x slower than k_spconv_t4):
//   * its own register ring of row gathers (unconditional, in-bounds loads: the compiler's vmcnt counting stays exact);
//   * the NEXT run's weight slice is prefetched at a run's first group by LDS-DMA (global_load_lds, inline asm) into the
//     wave's private LDS slot and moved to registers at the next run's start.  It has to bypass the compiler: a load issued
//     under a (wave-uniform) branch makes hipcc's wait-count pass drain the whole ring with `s_waitcnt vmcnt(0)` at every
//     join (measured: a memory round trip per run, 1.3-4x slower than k_spconv_t4), and an inline-asm load into REGISTERS is
//     not safe either (the compiler copies the "already written" output registers before the data has landed).  Data in
//     flight into LDS has no compiler-visible name; loads return in order, so "the slot has landed" = "at most the loads
//     issued since are outstanding" = s_waitcnt vmcnt(iterations since the prefetch x NKC), counted by hand.  The compiler's own
//     counted waits only get stricter by loads it does not know about, never weaker;
//   * its own 16 NTW columns of the tile's accumulator.
// The block's waves gather the same rows at about the same time, so all but the first hit the vector L1.
// Summation order per output element: filter offsets ascending, per rule k ascending -- a rule's whole Cin product is formed in
// the MFMA accumulator and added to the output row once (as k_spconv_fwd / k_spconv_blk do: bit-identical to them).
template <int NKC, int NTW, int D>
__global__ __launch_bounds__(512) void k_spconv_cw(const int* __restrict__ grp_start, const int* __restrict__ grp_o,
                                                    const int* __restrict__ grp_in, const int* __restrict__ grp_out,
                                                    int K, int A_out, const float* __restrict__ in, int ld_in,
                                                    const float* __restrict__ Wp, int w_flip, float* __restrict__ out,
                                                    int ld_out, int TM, int MU) {
  constexpr int CP = 16 * NTW;                    // columns per wave
  constexpr int NP = NKC * NTW;                   // 1 KiB weight pieces per run and wave
  extern __shared__ float4 cs_smem4[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int NW = blockDim.x >> 6;
  const int LD = NW * CP + 4;                                                           // accumulator row stride (floats)
  const int MS = MU + 2 * D;                                                            // staged groups: chunk + ring look-ahead (the ring issues up to 2 D - 2 past the chunk)
  float4* BS = cs_smem4;                                                                // [NW][NP][64] weight slot per wave (first: low LDS addresses for M0)
  float* ACC = reinterpret_cast<float*>(BS + (size_t)NW * NP * 64);                     // [TM + 1][LD], row TM = sink of padding rules
  unsigned* m_in = reinterpret_cast<unsigned*>(ACC + (size_t)(TM + 1) * LD);            // [MS][16] byte offset / 16 of the input row
  unsigned short* m_out = reinterpret_cast<unsigned short*>(m_in + MS * 16);            // [MS][16] accumulator row
  unsigned* m_o = reinterpret_cast<unsigned*>(m_out + MS * 16);                         // [MS]     filter offset | groups left in the run << 8
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, ntiles = gridDim.x;
  const int row0 = tile * TM;
  const int ct0 = (blockIdx.y * NW + wv) * NTW;   // this wave's first 16-column tile of the output
  const int gb = grp_start[tile], ge = grp_start[tile + 1], G = grp_start[ntiles];
  const unsigned ld4 = (unsigned)ld_in >> 2;
  for (int i = tid; i < (TM + 1) * LD / 4; i += blockDim.x) reinterpret_cast<float4*>(ACC)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gb < ge) {
    const float4* __restrict__ in4 = reinterpret_cast<const float4*>(in) + q;
    const float4* __restrict__ wl = reinterpret_cast<const float4*>(Wp) + (size_t)ct0 * K * NKC * 64 + lane;   // + tile t: K * NKC * 64
    float4* slot = BS + (size_t)wv * NP * 64;
    const unsigned slot_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)slot);
    float* acc_lane = ACC + wv * CP + r;
    const unsigned o_last = (unsigned)grp_o[ge - 1] & 0xffu;

    float4 A[D][NKC];
    f32x4 Bc[NKC][NTW];
    unsigned o_prev = 0xffffu, o_slot = 0xffffu;   // offsets whose weights sit in Bc / in the LDS slot (landed or in flight)
    int since_b = 0;                               // loop iterations (= NKC row-gather loads each) since the last slot prefetch

    // piece (kk, t) of offset O_ -> slot[(kk * NTW + t) * 64 + lane]; M0 = LDS target (saved / restored: M0 is the compiler's)
#define CW_PREFETCH(O_)                                                                                       \
  {                                                                                                           \
    const float4* wo_ = wl + (size_t)(w_flip ? K - 1 - (int)(O_) : (int)(O_)) * (NKC * 64);                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the slot's previous content has been read out */    \
    _Pragma("unroll") for (int kk_ = 0; kk_ < NKC; ++kk_)                                                     \
      _Pragma("unroll") for (int t_ = 0; t_ < NTW; ++t_) {                                                    \
        unsigned sv_;                                                                                         \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(sv_)                                                                             \
                     : "v"(wo_ + (size_t)t_ * K * NKC * 64 + kk_ * 64), "s"(slot_lds + (unsigned)((kk_ * NTW + t_) * 1024)) \
                     : "memory");                                                                             \
      }                                                                                                       \
    o_slot = (O_);                                                                                            \
    since_b = 0;                                                                                              \
  }
#define CW_WAIT_N(N_) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N_) : "memory")
#define CW_WAIT_SLOT()                                                                                        \
  {                                                                                                           \
    if (since_b >= 4 && 4 * NKC <= 60) CW_WAIT_N(4 * NKC <= 60 ? 4 * NKC : 0);                                \
    else if (since_b == 3 && 3 * NKC <= 60) CW_WAIT_N(3 * NKC <= 60 ? 3 * NKC : 0);                           \
    else if (since_b >= 2 && 2 * NKC <= 60) CW_WAIT_N(2 * NKC <= 60 ? 2 * NKC : 0);                           \
    else if (since_b >= 1) CW_WAIT_N(NKC);                                                                    \
    else CW_WAIT_N(0);                                                                                        \
  }
#define CW_ISSUE(S_, C_)                                                                                      \
  {                                                                                                           \
    const unsigned io_ = m_in[(C_) * 16 + r];                                                                 \
    _Pragma("unroll") for (int kk_ = 0; kk_ < NKC; ++kk_) A[S_][kk_] = in4[(size_t)io_ + (unsigned)(kk_ * 4)]; \
  }

    for (int cb = gb; cb < ge; cb += MU) {
      __syncthreads();   // everyone is done with the previous chunk's metadata (first pass: the accumulator is zeroed)
      for (int e = tid * 4; e < MS * 16; e += 4 * blockDim.x) {
        const int src = min(cb * 16 + e, G * 16 - 4);
        const int4 vi = *reinterpret_cast<const int4*>(grp_in + src);
        const int4 vo = *reinterpret_cast<const int4*>(grp_out + src);
        const bool dead = cb + (e >> 4) >= ge || (e >> 4) >= MU;   // look-ahead groups: loaded, never accumulated
        uint4 wi;
        wi.x = dead ? 0u : (unsigned)max(vi.x, 0) * ld4; wi.y = dead ? 0u : (unsigned)max(vi.y, 0) * ld4;
        wi.z = dead ? 0u : (unsigned)max(vi.z, 0) * ld4; wi.w = dead ? 0u : (unsigned)max(vi.w, 0) * ld4;
        const unsigned o0 = (dead || vo.x < 0) ? (unsigned)TM : (unsigned)vo.x, o1 = (dead || vo.y < 0) ? (unsigned)TM : (unsigned)vo.y;
        const unsigned o2 = (dead || vo.z < 0) ? (unsigned)TM : (unsigned)vo.z, o3 = (dead || vo.w < 0) ? (unsigned)TM : (unsigned)vo.w;
        *reinterpret_cast<uint4*>(m_in + e) = wi;
        *reinterpret_cast<uint2*>(m_out + e) = make_uint2(o0 | (o1 << 16), o2 | (o3 << 16));
      }
      // offset | groups left in the run << 8.  A look-ahead entry stands for the group it covers (the next chunk's first groups,
      // or the tile's last offset past its end), so the weight prefetch below sees run boundaries across chunks, never a fake one
      for (int e = tid; e < MS; e += blockDim.x) m_o[e] = (cb + e < ge) ? (unsigned)grp_o[cb + e] : (o_last | 0x100u);
      __syncthreads();
      const int ng = min(MU, ge - cb);
#pragma unroll
      for (int s = 0; s < D; ++s) {
        CW_ISSUE(s, s);
        __builtin_amdgcn_sched_barrier(0);   // keep the ring in issue order: the loop's counted vmcnt relies on it
      }
      since_b += D;   // (the staging loads above came after any prefetch in flight as well: the hand count only errs on the strict side)
      // metadata of the next group to multiply, fetched one group ahead: accumulator rows (per lane) and filter offsets
      uint2 mo_n = *reinterpret_cast<const uint2*>(m_out + q * 4);
      unsigned w_c = __builtin_amdgcn_readfirstlane(m_o[0]), w_n = __builtin_amdgcn_readfirstlane(m_o[1]);
      for (int u = 0; u < ng; u += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
          const int c = u + s;                                  // chunk position of this group (>= ng: a dead look-ahead group)
          float* p0 = acc_lane + (mo_n.x & 0xffffu) * LD;
          float* p1 = acc_lane + (mo_n.x >> 16) * LD;
          float* p2 = acc_lane + (mo_n.y & 0xffffu) * LD;
          float* p3 = acc_lane + (mo_n.y >> 16) * LD;
          float v[4][NTW];                                      // the 4 rules of a lane are distinct output rows (or the sink)
#pragma unroll
          for (int t = 0; t < NTW; ++t) { v[0][t] = p0[16 * t]; v[1][t] = p1[16 * t]; v[2][t] = p2[16 * t]; v[3][t] = p3[16 * t]; }
          const unsigned w_nn = m_o[c + 2 < MS ? c + 2 : MS - 1];
          __builtin_amdgcn_sched_barrier(0);                    // keep the accumulator reads in flight under the MFMAs
          const unsigned o_c = w_c & 0xffu;
          if (o_c != o_prev) {                                  // a new run starts
            if (o_slot != o_c) CW_PREFETCH(o_c);                // (not prefetched: first run of the tile, or a run cut by a chunk)
            CW_WAIT_SLOT();
#pragma unroll
            for (int kk = 0; kk < NKC; ++kk)
#pragma unroll
              for (int t = 0; t < NTW; ++t) {
                const float4 b_ = slot[(kk * NTW + t) * 64 + lane];
                Bc[kk][t] = (f32x4){b_.x, b_.y, b_.z, b_.w};
              }
            o_prev = o_c;
            // fetch the NEXT run's weights now: they have this whole run to arrive
            const int nx = min(c + (int)(w_c >> 8), MS - 1);
            const unsigned o_nx = __builtin_amdgcn_readfirstlane(m_o[nx]) & 0xffu;
            CW_PREFETCH(o_nx);
          }
          f32x4 d[NTW];
#pragma unroll
          for (int t = 0; t < NTW; ++t) d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < NKC; ++kk) {
            const float av[4] = {A[s][kk].x, A[s][kk].y, A[s][kk].z, A[s][kk].w};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
              for (int t = 0; t < NTW; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], Bc[kk][t][s2], d[t], 0, 0, 0);
          }
          mo_n = *reinterpret_cast<const uint2*>(m_out + (c + 1) * 16 + q * 4);
          w_c = w_n;
          w_n = __builtin_amdgcn_readfirstlane(w_nn);
          CW_ISSUE(s, c + D);
          ++since_b;
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            p0[16 * t] = v[0][t] + d[t][0]; p1[16 * t] = v[1][t] + d[t][1];
            p2[16 * t] = v[2][t] + d[t][2]; p3[16 * t] = v[3][t] + d[t][3];
          }
        }
      }
    }
#undef CW_ISSUE
#undef CW_PREFETCH
#undef CW_WAIT_SLOT
#undef CW_WAIT_N
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a prefetch may still be in flight into this wave's slot
  }
  __syncthreads();
  // each output element is written exactly once
  const int V = NW * CP / 4;   // float4 per row of this block's columns
  for (int i = tid; i < TM * V; i += blockDim.x) {
    const int rr = i / V, c4 = i - rr * V;
    if (row0 + rr < A_out)
      *reinterpret_cast<float4*>(out + (int64_t)(row0 + rr) * ld_out + blockIdx.y * (NW * CP) + c4 * 4) =
          *reinterpret_cast<const float4*>(ACC + rr * LD + c4 * 4);
  }
}

static inline int cs_mu(int tile_rows) { return tile_rows == 256 ? 160 : 96; }

template <int NKC, int NTW, int D>
static int launch_cw(const int* gs, const int* go, const int* gi, const int* gout, int K, int A_out, int TM, const float* in, int ld_in,
                     const float* Wp, int cout, int w_flip, float* out, int ld_out, hipStream_t st) {
  const int MU = cs_mu(TM);
  const int nsl = cout / (16 * NTW);            // column slices = waves over all blocks of a tile
  int nw = nsl;                                 // waves per block: all slices up to 8, else an even split
  if (nw > 8) nw = (nsl % 2 == 0) ? nsl / 2 : (nsl % 3 == 0 ? nsl / 3 : 1);
  const size_t lds = (size_t)nw * NKC * NTW * 1024 + (size_t)(TM + 1) * (nw * 16 * NTW + 4) * 4 + (size_t)(MU + 2 * D) * 100;
  if (lds > 160 * 1024 || (size_t)nw * NKC * NTW * 1024 > 60 * 1024) return MOPA_ERR_ARG;   // weight slots stay below 64 KiB (M0)
  auto kern = k_spconv_cw<NKC, NTW, D>;
  static std::atomic<bool> attr_set{false};   // caches an idempotent call, carries no state a result depends on
  if (!attr_set.load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return MOPA_ERR_LAUNCH;
    attr_set.store(true, std::memory_order_release);
  }
  dim3 grid((unsigned)cdiv64(A_out, TM), nsl / nw);
  kern<<<grid, 64 * nw, lds, st>>>(gs, go, gi, gout, K, A_out, in, ld_in, Wp, w_flip, out, ld_out, TM, MU);
  return hipGetLastError() == hipSuccess ? MOPA_OK : MOPA_ERR_LAUNCH;
}

// 1 if mopa_spconv_fwd_cs has a kernel for the shape (cin / cout of the convolution to run, i.e. swapped for backward-data).
MOPA_API int mopa_spconv_cs_supported(int32_t cin, int32_t cout) {
  if (cin % 16 || cout % 16) return 0;
  const int nkc = cin / 16, nct = cout / 16;
  const bool kc_ok = (nkc >= 1 && nkc <= 8) || nkc == 10 || nkc == 12;
  return kc_ok && nct >= 1 && nct <= 16;
}

// Same contract as mopa_spconv_fwd_grouped with packed weights (mopa_spconv_pack_weight, ntw = 1), on the TM-row rulebook
// of the table (mopa_rulebook_cs_count / _fill).  w_flip bit 0: mirrored filter offsets (backward-data of a submanifold table);
// bits 8-15: 16-column tiles per wave (1 or 2; 0 = the default: 2 where Cout / 16 is even).
MOPA_API int mopa_spconv_fwd_cs(const int32_t* grp_start, const int32_t* grp_o, const int32_t* grp_in, const int32_t* grp_out,
                                int32_t K, int32_t num_out, int32_t tile_rows, const float* in, int32_t ld_in, int32_t cin,
                                const float* weight_packed, int32_t cout, int32_t w_flip, float* out, int32_t ld_out, void* stream) {
  if (K <= 0 || K > 27 || num_out <= 0 || (tile_rows != 128 && tile_rows != 256) || !mopa_spconv_cs_supported(cin, cout)) return MOPA_ERR_ARG;
  if (ld_in < cin || ld_out < cout || ld_in % 4 || ld_out % 4 || (((uintptr_t)in | (uintptr_t)out | (uintptr_t)weight_packed) & 15)) return MOPA_ERR_ARG;
  if ((int64_t)num_out * 8 * ld_in * 4 >= (1ll << 36)) return MOPA_ERR_ARG;   // input rows are addressed by 32-bit float4 indices
  hipStream_t st = (hipStream_t)stream;
  int ntw = (w_flip >> 8) & 0xff;
  if (ntw == 0) ntw = ((cout / 16) % 2 == 0) ? 2 : 1;
  if (ntw > 2 || (cout / 16) % ntw) return MOPA_ERR_ARG;
#define CW(NKC_, D1_, D2_)                                                                                                   \
  return ntw == 1 ? launch_cw<NKC_, 1, D1_>(grp_start, grp_o, grp_in, grp_out, K, num_out, tile_rows, in, ld_in, weight_packed, cout, w_flip & 1, out, ld_out, st) \
                  : launch_cw<NKC_, 2, D2_>(grp_start, grp_o, grp_in, grp_out, K, num_out, tile_rows, in, ld_in, weight_packed, cout, w_flip & 1, out, ld_out, st)
  switch (cin / 16) {
    case 1: CW(1, 8, 6);
    case 2: CW(2, 6, 4);
    case 3: CW(3, 4, 3);
    case 4: CW(4, 4, 3);
    case 5: CW(5, 3, 2);
    case 6: CW(6, 3, 2);
    case 7: CW(7, 2, 2);
    case 8: CW(8, 2, 2);
    case 10: CW(10, 2, 2);
    case 12: CW(12, 2, 2);
    default: return MOPA_ERR_ARG;
  }
#undef CW
}
