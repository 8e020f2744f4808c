// Row-wise ops of the sparse 3D branch: BatchNorm(+Leaky)ReLU over active rows, InputLayer (mode 4),
// OutputLayer fused with the two linear heads.  HBM-bound elementwise / reduction kernels: float4 access,
// wave-shuffle + LDS reductions, two-stage (partials -> finalize) so results are deterministic.
//
// Reference call sites: mopa/models/scn_unet.py:26 (InputLayer), :28-29 (BatchNormReLU inside scn.UNet and
// after it), :30 (OutputLayer); mopa/models/xmuda_arch.py:102,107,116,124 (linear heads).
// Semantics: SURVEY.md Appendix A.2, A.3, A.6; oracle: oracle/scn3d.py::{input_layer,output_layer,bn_relu}.
#include "common.h"

// Rows handled by one block of the two-stage reductions: enough blocks to fill 256 CUs several times over even for
// the short-and-wide tensors of the deep layers (e.g. 4,560 rows x 512 channels), capped at 1024 rows.
static inline int bn_rows_per_block(int64_t num_rows) {
  int64_t r = cdiv64(num_rows, 2048);
  if (r < 32) r = 32;
  if (r > 1024) r = 1024;
  return (int)r;
}
static inline int bn_num_blocks(int64_t num_rows) { return (int)cdiv64(num_rows, bn_rows_per_block(num_rows)); }

// Up to three consecutive row ranges of one tensor that BatchNorm treats as separate batches (the source / target / VGI batch of
// one iteration in one pass: mopa_amd Net2DSeg "bn_groups", Net3DSeg "bn_group_points").  One launch per kernel serves all groups
// (blockIdx.y = group): a layer is 3 launches per direction instead of 3 per group.  Every group keeps the block partition its
// own call would have (rows per block from ITS row count), so the sums -- and the results -- are bit-identical to per-group calls.
#define BN_MAX_GROUPS 3
struct BnGroups {
  int n;
  int row0[BN_MAX_GROUPS], rows[BN_MAX_GROUPS], rpb[BN_MAX_GROUPS], nblk[BN_MAX_GROUPS], poff[BN_MAX_GROUPS];  // poff: first partial block
  int nblk_max, nblk_total;
};
static inline bool bn_make_groups(BnGroups* g, int num_rows, int n_groups, int split1, int split2) {
  if (n_groups < 1 || n_groups > BN_MAX_GROUPS || num_rows <= 0) return false;
  const int b[4] = {0, n_groups > 1 ? split1 : num_rows, n_groups > 2 ? split2 : num_rows, num_rows};
  g->n = n_groups;
  g->nblk_max = g->nblk_total = 0;
  for (int k = 0; k < BN_MAX_GROUPS; ++k) { g->row0[k] = g->rows[k] = g->nblk[k] = g->poff[k] = 0; g->rpb[k] = 1; }
  for (int k = 0; k < n_groups; ++k) {
    const int r0 = b[k], r1 = k == n_groups - 1 ? num_rows : b[k + 1];
    if (r1 <= r0) return false;
    g->row0[k] = r0; g->rows[k] = r1 - r0;
    g->rpb[k] = bn_rows_per_block(r1 - r0);
    g->nblk[k] = bn_num_blocks(r1 - r0);
    g->poff[k] = g->nblk_total;
    g->nblk_total += g->nblk[k];
    if (g->nblk[k] > g->nblk_max) g->nblk_max = g->nblk[k];
  }
  return true;
}

// ------------------------------------------------------------------------------------------ BN statistics
// partial[blk][0][c] = sum(x - x0), partial[blk][1][c] = sum((x - x0)^2) with x0 = first row (shifted sums keep
// fp32 accurate when |mean| >> std).  C % 4 == 0.
__global__ __launch_bounds__(256) void k_bn_stats_partial(const float* __restrict__ x, int ld, int C, const BnGroups grp,
                                                           float* __restrict__ partial) {
  extern __shared__ float lds[];  // [2][RL][C]
  const int gi = blockIdx.y;
  if ((int)blockIdx.x >= grp.nblk[gi]) return;
  const int CQ = C >> 2;
  const int RL = 256 / CQ;
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const int rpb = grp.rpb[gi];
  const int rbeg = grp.row0[gi] + blockIdx.x * rpb, rend = min(grp.row0[gi] + grp.rows[gi], rbeg + rpb);
  partial += (int64_t)grp.poff[gi] * 2 * C;
  float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (rl < RL) {
    const float4 k = *reinterpret_cast<const float4*>(x + (int64_t)grp.row0[gi] * ld + cq * 4);
    for (int row = rbeg + rl; row < rend; row += RL) {
      const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)row * ld + cq * 4);
      float d0 = v.x - k.x, d1 = v.y - k.y, d2 = v.z - k.z, d3 = v.w - k.w;
      s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
      ss[0] += d0 * d0; ss[1] += d1 * d1; ss[2] += d2 * d2; ss[3] += d3 * d3;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lds[(0 * RL + rl) * C + cq * 4 + j] = s[j];
      lds[(1 * RL + rl) * C + cq * 4 + j] = ss[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float t = 0.f;
    for (int k = 0; k < RL; ++k) t += lds[(which * RL + k) * C + c];
    partial[(int64_t)blockIdx.x * 2 * C + i] = t;
  }
}

// One block of 256 per channel: the threads stride over the block partials (<= 8 independent loads each; one wave per channel
// walked them 32 deep and the launch took 11 us of pure latency), wave sums in double, the four wave sums added in order;
// thread 0 writes scale/shift (y = x*scale + shift), mean, invstd and updates the running statistics.
__device__ __forceinline__ void bn_block_sum2(const float* __restrict__ partial, int nblk, int C, int c, double& s, double& ss) {
  __shared__ double red[2][4];
  s = 0.0; ss = 0.0;
#pragma unroll 8
  for (int b = threadIdx.x; b < nblk; b += 256) {
    s += (double)partial[(int64_t)b * 2 * C + c];
    ss += (double)partial[(int64_t)b * 2 * C + C + c];
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  s = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
  ss = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
}
// Groups are finalised one after the other by the same block: the running statistics see them in call order.
__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ partial_all, const BnGroups grp, const float* __restrict__ x_all,
                              int ldx, int C,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                              float eps, int training, float* __restrict__ stats_all) {
  const int c = blockIdx.x;
 for (int gi = 0; gi < grp.n; ++gi) {
  const float* __restrict__ partial = partial_all + (int64_t)grp.poff[gi] * 2 * C;
  const float* __restrict__ x0 = x_all + (int64_t)grp.row0[gi] * ldx;
  const int nblk = grp.nblk[gi], A = grp.rows[gi];
  float* __restrict__ scale = stats_all + (int64_t)gi * 4 * C;
  float* __restrict__ shift = scale + C;
  float* __restrict__ save_mean = scale + 2 * C;
  float* __restrict__ save_invstd = scale + 3 * C;
  float mean, var;
  if (training) {
    double s, ss;
    bn_block_sum2(partial, nblk, C, c, s, ss);
    const double m = s / A;
    double v = ss / A - m * m;
    if (v < 0) v = 0;
    mean = (float)((double)x0[c] + m);
    var = (float)v;
    if (threadIdx.x == 0) {
      const float unbiased = (float)(v * ((double)A / (double)(A > 1 ? A - 1 : 1)));
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  if (threadIdx.x == 0) {
    const float invstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    save_mean[c] = mean;
    save_invstd[c] = invstd;
  }
  __syncthreads();   // (bn_block_sum2's shared scratch is re-used by the next group; thread 0's running statistics are ordered by it too)
 }
}

// y = act(x*scale + shift (+ res));  act: 0 = identity, 1 = leaky-ReLU(leak).
// Thread = (row lane rl, channel quad cq): the per-channel constants live in registers and the thread walks rows
// rl, rl + RL*grid, ... -- no per-element constant loads, no 64-bit division (the grid-stride form was TA-bound at 2.7 TB/s).
__global__ __launch_bounds__(256) void k_bn_relu_apply(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                        int ldy, const BnGroups grp, int C, const float* __restrict__ stats_all,
                                                        float leak, const float* __restrict__ res, int ld_res, int act) {
  const int CQ = C >> 2, RL = 256 / CQ;
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  if (rl >= RL) return;
  const int gi = blockIdx.y;
  const float* __restrict__ scale = stats_all + (int64_t)gi * 4 * C;
  const float4 sc = *reinterpret_cast<const float4*>(scale + cq * 4);
  const float4 sh = *reinterpret_cast<const float4*>(scale + C + cq * 4);
  const int A = grp.row0[gi] + grp.rows[gi];
  for (int row = grp.row0[gi] + blockIdx.x * RL + rl; row < A; row += gridDim.x * RL) {
    const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + cq * 4);
    float4 o;
    o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y); o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
    if (res) {
      const float4 rv = *reinterpret_cast<const float4*>(res + (int64_t)row * ld_res + cq * 4);
      o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    if (act) {
      o.x = o.x > 0.f ? o.x : o.x * leak; o.y = o.y > 0.f ? o.y : o.y * leak;
      o.z = o.z > 0.f ? o.z : o.z * leak; o.w = o.w > 0.f ? o.w : o.w * leak;
    }
    *reinterpret_cast<float4*>(y + (int64_t)row * ldy + cq * 4) = o;
  }
}

// Blocks for the (rl, cq) row-walking elementwise kernels: RL rows per block step, ~8 blocks per CU, at least 4 rows per thread.
static inline int bn_apply_grid(int64_t num_rows, int C) {
  const int RL = 256 / (C >> 2);
  int64_t g = cdiv64(num_rows, (int64_t)RL * 4);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

MOPA_API size_t mopa_bnrelu_rows_workspace_bytes(int32_t num_rows, int32_t C) {
  // partial blocks of a grouped call: every group partitions its own rows (>= 32 rows per block, <= 2048 blocks per group)
  int64_t nb = (int64_t)num_rows / 32 + BN_MAX_GROUPS;
  if (nb > BN_MAX_GROUPS * 2048) nb = BN_MAX_GROUPS * 2048;
  if (nb < bn_num_blocks(num_rows)) nb = bn_num_blocks(num_rows);
  return align_up((size_t)nb * 2 * C * sizeof(float), 256);
}

// y = act(batchnorm(x) (+ res)).  stats[4][C] receives scale, shift, mean, invstd (saved for backward).
// act: 0 identity / 1 leaky-ReLU(leak).  res (optional) is added before the activation (ResNet BasicBlock tail).
// n_groups (1..3) consecutive row ranges [0, split1), [split1, split2), [split2, num_rows) are normalised as separate batches, in that
// order (running statistics: group 0 first); stats = [n_groups][4][C].  (Workspace: mopa_bnrelu_rows_workspace_bytes of the whole
// tensor + one block per extra group.)
// y == null: statistics, running statistics and stats only -- the consumer applies scale / shift / activation while it reads x
// (mopa_wino4_input_bn: the BatchNorm between two convolutions of a ResNet block never materialises its output).
MOPA_API int mopa_bn_act_fwd_groups(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t num_rows, int32_t C,
                                    int32_t n_groups, int32_t split1, int32_t split2,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float momentum, float eps, float leak, int32_t act, const float* res, int32_t ld_res,
                                    int32_t training, float* stats, void* ws, size_t ws_bytes, void* stream) {
  if (num_rows <= 0 || C <= 0 || (C & 3) || C > 1024 || ldx < C || (ldx & 3) || (y && (ldy < C || (ldy & 3)))) return MOPA_ERR_ARG;
  if (res && (ld_res < C || (ld_res & 3) || !y)) return MOPA_ERR_ARG;
  BnGroups grp;
  if (!bn_make_groups(&grp, num_rows, n_groups, split1, split2)) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)ws;
  if (training) {
    if (ws_bytes < (size_t)grp.nblk_total * 2 * C * sizeof(float)) return MOPA_ERR_WORKSPACE;
    const int RL = 256 / (C >> 2);
    if (RL < 1) return MOPA_ERR_ARG;
    k_bn_stats_partial<<<dim3(grp.nblk_max, grp.n), 256, (size_t)2 * RL * C * sizeof(float), st>>>(x, ldx, C, grp, partial);
  }
  k_bn_finalize<<<C, 256, 0, st>>>(partial, grp, x, ldx, C, gamma, beta, running_mean, running_var, momentum, eps, training, stats);
  if (y) {
    int maxrows = 0;
    for (int k = 0; k < grp.n; ++k) maxrows = grp.rows[k] > maxrows ? grp.rows[k] : maxrows;
    k_bn_relu_apply<<<dim3(bn_apply_grid(maxrows, C), grp.n), 256, 0, st>>>(x, ldx, y, ldy, grp, C, stats, leak, res, ld_res, act);
  }
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_bn_act_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t num_rows, int32_t C,
                             const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, float leak, int32_t act, const float* res, int32_t ld_res,
                             int32_t training, float* stats, void* ws, size_t ws_bytes, void* stream) {
  return mopa_bn_act_fwd_groups(x, ldx, y, ldy, num_rows, C, 1, 0, 0, gamma, beta, running_mean, running_var, momentum, eps, leak, act,
                                res, ld_res, training, stats, ws, ws_bytes, stream);
}

MOPA_API int mopa_bnrelu_rows_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t num_rows, int32_t C,
                                  const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  float momentum, float eps, float leak, int32_t training, float* stats, void* ws,
                                  size_t ws_bytes, void* stream) {
  return mopa_bn_act_fwd(x, ldx, y, ldy, num_rows, C, gamma, beta, running_mean, running_var, momentum, eps, leak, 1,
                         nullptr, 0, training, stats, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------ BN backward
// dz = dy * act'(.)  where the activation mask comes from the saved output y (ymask, needed when a residual was
// added) or is recomputed from x (ymask == null).  act == 0: dz = dy.  Partial sums of dz and dz*xhat.
__device__ __forceinline__ float bn_dz(float g, float xv, float sc, float sh, float leak, int act, const float* ym, int j) {
  if (!act) return g;
  const float yv = ym ? ym[j] : fmaf(xv, sc, sh);
  return yv > 0.f ? g : g * leak;
}

__global__ __launch_bounds__(256) void k_bn_bwd_partial(const float* __restrict__ dy, int ld_dy,
                                                         const float* __restrict__ x, int ldx, int C,
                                                         const float* __restrict__ stats, float leak,
                                                         const float* __restrict__ ymask, int ld_ym, int act, const BnGroups grp,
                                                         float* __restrict__ partial) {
  extern __shared__ float lds[];
  const int gi = blockIdx.y;
  if ((int)blockIdx.x >= grp.nblk[gi]) return;
  const int CQ = C >> 2;
  const int RL = 256 / CQ;
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const int rpb = grp.rpb[gi];
  const int rbeg = grp.row0[gi] + blockIdx.x * rpb, rend = min(grp.row0[gi] + grp.rows[gi], rbeg + rpb);
  partial += (int64_t)grp.poff[gi] * 2 * C;
  stats += (int64_t)gi * 4 * C;
  float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (rl < RL) {
    float sc[4], sh[4], mu[4], is[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = stats[cq * 4 + j]; sh[j] = stats[C + cq * 4 + j];
      mu[j] = stats[2 * C + cq * 4 + j]; is[j] = stats[3 * C + cq * 4 + j];
    }
    // four rows of loads in flight per thread (the sums take the rows in the same order as a rolled loop: same bits)
    int row = rbeg + rl;
    for (; row + 3 * RL < rend; row += 4 * RL) {
      float4 xv[4], gv[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = *reinterpret_cast<const float4*>(x + (int64_t)(row + u * RL) * ldx + cq * 4);
        gv[u] = *reinterpret_cast<const float4*>(dy + (int64_t)(row + u * RL) * ld_dy + cq * 4);
        yv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ymask) yv[u] = *reinterpret_cast<const float4*>(ymask + (int64_t)(row + u * RL) * ld_ym + cq * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, gs[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w}, ys[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dz = bn_dz(gs[j], xs[j], sc[j], sh[j], leak, act, ymask ? ys : nullptr, j);
          s[j] += dz;
          ss[j] += dz * ((xs[j] - mu[j]) * is[j]);
        }
      }
    }
    for (; row < rend; row += RL) {
      const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + cq * 4);
      const float4 gv = *reinterpret_cast<const float4*>(dy + (int64_t)row * ld_dy + cq * 4);
      float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ymask) yv = *reinterpret_cast<const float4*>(ymask + (int64_t)row * ld_ym + cq * 4);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dz = bn_dz(gs[j], xs[j], sc[j], sh[j], leak, act, ymask ? ys : nullptr, j);
        s[j] += dz;
        ss[j] += dz * ((xs[j] - mu[j]) * is[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lds[(0 * RL + rl) * C + cq * 4 + j] = s[j];
      lds[(1 * RL + rl) * C + cq * 4 + j] = ss[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float t = 0.f;
    for (int k = 0; k < RL; ++k) t += lds[(which * RL + k) * C + c];
    partial[(int64_t)blockIdx.x * 2 * C + i] = t;
  }
}

// dgamma/dbeta (+= if accumulate) and the two per-channel means used by the apply pass (coef[2][C]).
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ partial_all, const BnGroups grp, int C,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                          float* __restrict__ coef_all) {
  const int c = blockIdx.x;   // one block per channel (see k_bn_finalize); the groups' parameter gradients add up in group order
  for (int gi = 0; gi < grp.n; ++gi) {
    double s, ss;
    bn_block_sum2(partial_all + (int64_t)grp.poff[gi] * 2 * C, grp.nblk[gi], C, c, s, ss);
    if (threadIdx.x == 0) {
      dbeta[c] = ((accumulate || gi > 0) ? dbeta[c] : 0.f) + (float)s;
      dgamma[c] = ((accumulate || gi > 0) ? dgamma[c] : 0.f) + (float)ss;
      coef_all[(int64_t)gi * 2 * C + c] = (float)(s / grp.rows[gi]);
      coef_all[(int64_t)gi * 2 * C + C + c] = (float)(ss / grp.rows[gi]);
    }
    __syncthreads();
  }
}

// training: dx = scale * (dz - mean(dz) - xhat * mean(dz*xhat));  eval: dx = scale * dz.   dx (+)= if acc_dx.
// dres (optional) receives dz, the gradient of the residual input (+= if acc_dres).
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dy, int ld_dy, const float* __restrict__ x,
                                                       int ldx, float* __restrict__ dx, int ld_dx, const BnGroups grp, int C,
                                                       const float* __restrict__ stats, const float* __restrict__ coef,
                                                       float leak, int training, int acc_dx, const float* __restrict__ ymask,
                                                       int ld_ym, int act, float* __restrict__ dres, int ld_dres, int acc_dres) {
  // thread = (row lane, channel quad), constants in registers (see k_bn_relu_apply)
  const int CQ = C >> 2, RL = 256 / CQ;
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  if (rl >= RL) return;
  const int gi = blockIdx.y;
  stats += (int64_t)gi * 4 * C;
  coef += (int64_t)gi * 2 * C;
  float sc[4], sh[4], mean[4], inv[4], c0[4], c1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = cq * 4 + j;
    sc[j] = stats[c]; sh[j] = stats[C + c]; mean[j] = stats[2 * C + c]; inv[j] = stats[3 * C + c];
    c0[j] = coef[c]; c1[j] = coef[C + c];
  }
  const int A = grp.row0[gi] + grp.rows[gi];
  for (int row = grp.row0[gi] + blockIdx.x * RL + rl; row < A; row += gridDim.x * RL) {
    const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + cq * 4);
    const float4 gv = *reinterpret_cast<const float4*>(dy + (int64_t)row * ld_dy + cq * 4);
    float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ymask) yv = *reinterpret_cast<const float4*>(ymask + (int64_t)row * ld_ym + cq * 4);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
    float o[4], dzv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dz = bn_dz(gs[j], xs[j], sc[j], sh[j], leak, act, ymask ? ys : nullptr, j);
      dzv[j] = dz;
      if (training) {
        const float xhat = (xs[j] - mean[j]) * inv[j];
        o[j] = sc[j] * (dz - c0[j] - xhat * c1[j]);
      } else {
        o[j] = sc[j] * dz;
      }
    }
    float4* dp = reinterpret_cast<float4*>(dx + (int64_t)row * ld_dx + cq * 4);
    if (acc_dx) {
      const float4 p = *dp;
      o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w;
    }
    *dp = make_float4(o[0], o[1], o[2], o[3]);
    if (dres) {
      float4* rp = reinterpret_cast<float4*>(dres + (int64_t)row * ld_dres + cq * 4);
      if (acc_dres) {
        const float4 p = *rp;
        dzv[0] += p.x; dzv[1] += p.y; dzv[2] += p.z; dzv[3] += p.w;
      }
      *rp = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
    }
  }
}

MOPA_API size_t mopa_bnrelu_rows_bwd_workspace_bytes(int32_t num_rows, int32_t C) {
  return mopa_bnrelu_rows_workspace_bytes(num_rows, C) + align_up((size_t)BN_MAX_GROUPS * 2 * C * sizeof(float), 256);
}

// General backward of mopa_bn_act_fwd.  ymask: the forward output y (required when a residual was added; optional
// otherwise).  dres: gradient of the residual input (optional).
// The grouped form: stats = the forward's [n_groups][4][C]; the groups' parameter gradients add up in group order.
MOPA_API int mopa_bn_act_bwd_groups(const float* dy, int32_t ld_dy, const float* x, int32_t ldx, float* dx, int32_t ld_dx,
                                    int32_t num_rows, int32_t C, int32_t n_groups, int32_t split1, int32_t split2, const float* stats,
                                    float leak, int32_t act, const float* ymask, int32_t ld_ym, float* dres, int32_t ld_dres,
                                    int32_t accumulate_dres, int32_t training, float* dgamma, float* dbeta,
                                    int32_t accumulate_param_grads, int32_t accumulate_dx, void* ws, size_t ws_bytes, void* stream) {
  if (num_rows <= 0 || C <= 0 || (C & 3) || C > 1024 || ldx < C || ld_dy < C || ld_dx < C || ((ldx | ld_dy | ld_dx) & 3))
    return MOPA_ERR_ARG;
  if ((ymask && (ld_ym < C || (ld_ym & 3))) || (dres && (ld_dres < C || (ld_dres & 3)))) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_bnrelu_rows_bwd_workspace_bytes(num_rows, C)) return MOPA_ERR_WORKSPACE;
  BnGroups grp;
  if (!bn_make_groups(&grp, num_rows, n_groups, split1, split2)) return MOPA_ERR_ARG;
  // the partial slabs of all groups must fit in front of the coefficient block (rows per block stop growing at 1024, so groups of
  // more than 2 M rows hold more blocks than the workspace's cap of 3 x 2048: the forward refuses the same case)
  if ((size_t)grp.nblk_total * 2 * C * sizeof(float) > mopa_bnrelu_rows_workspace_bytes(num_rows, C)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)ws;
  float* coef = (float*)((char*)ws + mopa_bnrelu_rows_workspace_bytes(num_rows, C));
  const int RL = 256 / (C >> 2);
  k_bn_bwd_partial<<<dim3(grp.nblk_max, grp.n), 256, (size_t)2 * RL * C * sizeof(float), st>>>(dy, ld_dy, x, ldx, C, stats, leak, ymask,
                                                                                               ld_ym, act, grp, partial);
  k_bn_bwd_finalize<<<C, 256, 0, st>>>(partial, grp, C, dgamma, dbeta, accumulate_param_grads, coef);
  int maxrows = 0;
  for (int k = 0; k < grp.n; ++k) maxrows = grp.rows[k] > maxrows ? grp.rows[k] : maxrows;
  k_bn_bwd_apply<<<dim3(bn_apply_grid(maxrows, C), grp.n), 256, 0, st>>>(
      dy, ld_dy, x, ldx, dx, ld_dx, grp, C, stats, coef, leak, training, accumulate_dx, ymask, ld_ym, act, dres,
      ld_dres, accumulate_dres);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
// The reduction half of mopa_bn_act_bwd_groups alone: parameter gradients and coef_out[n_groups][2][C] = (mean(dz), mean(dz * xhat)) per
// group -- for a consumer that applies dx = scale * (dz - coef0 - xhat * coef1) while it reads (dy, x) itself (mopa_stem_bwd_weight_bn:
// the stem's BatchNorm gradient is read once, by the stem's weight gradient, and never written).
MOPA_API int mopa_bn_bwd_sums_groups(const float* dy, int32_t ld_dy, const float* x, int32_t ldx, int32_t num_rows, int32_t C,
                                     int32_t n_groups, int32_t split1, int32_t split2, const float* stats, float leak, int32_t act,
                                     const float* ymask, int32_t ld_ym, float* dgamma, float* dbeta, int32_t accumulate_param_grads,
                                     float* coef_out, void* ws, size_t ws_bytes, void* stream) {
  if (num_rows <= 0 || C <= 0 || (C & 3) || C > 1024 || ldx < C || ld_dy < C || ((ldx | ld_dy) & 3) || !coef_out) return MOPA_ERR_ARG;
  if (ymask && (ld_ym < C || (ld_ym & 3))) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_bnrelu_rows_workspace_bytes(num_rows, C)) return MOPA_ERR_WORKSPACE;
  BnGroups grp;
  if (!bn_make_groups(&grp, num_rows, n_groups, split1, split2)) return MOPA_ERR_ARG;
  if ((size_t)grp.nblk_total * 2 * C * sizeof(float) > mopa_bnrelu_rows_workspace_bytes(num_rows, C)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)ws;
  const int RL = 256 / (C >> 2);
  k_bn_bwd_partial<<<dim3(grp.nblk_max, grp.n), 256, (size_t)2 * RL * C * sizeof(float), st>>>(dy, ld_dy, x, ldx, C, stats, leak, ymask,
                                                                                               ld_ym, act, grp, partial);
  k_bn_bwd_finalize<<<C, 256, 0, st>>>(partial, grp, C, dgamma, dbeta, accumulate_param_grads, coef_out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
MOPA_API int mopa_bn_act_bwd(const float* dy, int32_t ld_dy, const float* x, int32_t ldx, float* dx, int32_t ld_dx,
                             int32_t num_rows, int32_t C, const float* stats, float leak, int32_t act,
                             const float* ymask, int32_t ld_ym, float* dres, int32_t ld_dres, int32_t accumulate_dres,
                             int32_t training, float* dgamma, float* dbeta, int32_t accumulate_param_grads,
                             int32_t accumulate_dx, void* ws, size_t ws_bytes, void* stream) {
  return mopa_bn_act_bwd_groups(dy, ld_dy, x, ldx, dx, ld_dx, num_rows, C, 1, 0, 0, stats, leak, act, ymask, ld_ym, dres, ld_dres,
                                accumulate_dres, training, dgamma, dbeta, accumulate_param_grads, accumulate_dx, ws, ws_bytes, stream);
}

MOPA_API int mopa_bnrelu_rows_bwd(const float* dy, int32_t ld_dy, const float* x, int32_t ldx, float* dx, int32_t ld_dx,
                                  int32_t num_rows, int32_t C, const float* stats, float leak, int32_t training,
                                  float* dgamma, float* dbeta, int32_t accumulate_param_grads, int32_t accumulate_dx,
                                  void* ws, size_t ws_bytes, void* stream) {
  return mopa_bn_act_bwd(dy, ld_dy, x, ldx, dx, ld_dx, num_rows, C, stats, leak, 1, nullptr, 0, nullptr, 0, 0, training,
                         dgamma, dbeta, accumulate_param_grads, accumulate_dx, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------ synchronised BatchNorm
// Data-parallel option ("optional SyncBN"): the reference is single-process, its BatchNorm statistics run over the
// whole batch (2D: B*H*W pixels, 3D: every active row of the batch).  With scans sharded over ranks each rank holds a slice of
// those rows; the three-stage form below reproduces the single-process statistics:
//   forward   mopa_bn_sync_moments        rank-local (mean, M2 = sum (x - mean)^2, n) per channel, in double
//             [host: all_gather of the 2C+1 doubles over the process group -- RCCL / gloo]
//             mopa_bn_act_fwd_sync        Chan's pairwise combination in rank order (deterministic), running statistics from the
//                                         GLOBAL moments, then the same apply kernel as the local path
//   backward  mopa_bn_sync_bwd_sums       rank-local sum dz, sum dz*xhat: added to dbeta / dgamma as they are (the parameter
//                                         gradients are summed over ranks later, with every other gradient) and exported in double
//             [host: all_reduce(SUM) of the 2C doubles]
//             mopa_bn_act_bwd_sync        coef = global sums / global row count, then the same apply kernel
// Cost: 2 small collectives per BN layer and direction -- an equivalence / small-batch tool, not the throughput configuration
// (mopa_amd.syncbn; tests/test_gpu_syncbn.py checks 2 ranks x half batch == 1 rank x full batch).
__global__ __launch_bounds__(256) void k_bn_local_moments(const float* __restrict__ partial, int nblk, const float* __restrict__ x0, int A,
                                                           int C, double* __restrict__ moments) {
  const int c = blockIdx.x;
  double s, ss;
  bn_block_sum2(partial, nblk, C, c, s, ss);
  if (threadIdx.x == 0) {
    const double m = s / A;
    double m2 = ss - s * m;   // sum (d - m)^2 over the shifted values d = x - x0
    if (m2 < 0) m2 = 0;
    moments[c] = (double)x0[c] + m;
    moments[C + c] = m2;
    if (c == 0) moments[2 * C] = (double)A;
  }
}

__global__ void k_bn_finalize_sync(const double* __restrict__ gathered, int world, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int S = 2 * C + 1;
  double n = 0.0, mean = 0.0, m2 = 0.0;   // Chan et al.: fold rank r into the running (n, mean, M2)
  for (int r = 0; r < world; ++r) {
    const double nr = gathered[(int64_t)r * S + 2 * C], mr = gathered[(int64_t)r * S + c], qr = gathered[(int64_t)r * S + C + c];
    if (nr <= 0.0) continue;
    const double nt = n + nr, d = mr - mean;
    mean += d * (nr / nt);
    m2 += qr + d * d * (n * nr / nt);
    n = nt;
  }
  const double var = n > 0 ? m2 / n : 0.0;
  const float meanf = (float)mean, varf = (float)var;
  const float unbiased = (float)(m2 / (n > 1 ? n - 1 : 1));
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  const float invstd = 1.0f / sqrtf(varf + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - meanf * sc;
  save_mean[c] = meanf;
  save_invstd[c] = invstd;
}

// moments: 2C+1 doubles (mean[C], M2[C], n) of this rank's rows.  ws as for mopa_bn_act_fwd.
MOPA_API int mopa_bn_sync_moments(const float* x, int32_t ldx, int32_t num_rows, int32_t C, double* moments, void* ws, size_t ws_bytes,
                                  void* stream) {
  if (num_rows <= 0 || C <= 0 || (C & 3) || C > 1024 || ldx < C || (ldx & 3) || !moments) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_bnrelu_rows_workspace_bytes(num_rows, C)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bn_num_blocks(num_rows), RL = 256 / (C >> 2);
  if (RL < 1) return MOPA_ERR_ARG;
  float* partial = (float*)ws;
  BnGroups one;
  bn_make_groups(&one, num_rows, 1, 0, 0);
  k_bn_stats_partial<<<nblk, 256, (size_t)2 * RL * C * sizeof(float), st>>>(x, ldx, C, one, partial);
  k_bn_local_moments<<<C, 256, 0, st>>>(partial, nblk, x, num_rows, C, moments);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// gathered: [world][2C+1] doubles (every rank's mopa_bn_sync_moments output, rank order).  Training mode by definition.
MOPA_API int mopa_bn_act_fwd_sync(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t num_rows, int32_t C, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps, float leak,
                                  int32_t act, const float* res, int32_t ld_res, const double* gathered, int32_t world, float* stats,
                                  void* stream) {
  // num_rows == 0: a rank without rows at this layer still owns a copy of the running statistics -- they are updated from the
  // gathered (global) moments like everywhere else, nothing is applied
  if (num_rows < 0 || C <= 0 || (C & 3) || C > 1024 || ldx < C || ldy < C || (ldx & 3) || (ldy & 3) || world <= 0 || !gathered)
    return MOPA_ERR_ARG;
  if (res && (ld_res < C || (ld_res & 3))) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  k_bn_finalize_sync<<<(C + 255) / 256, 256, 0, st>>>(gathered, world, C, gamma, beta, running_mean, running_var, momentum, eps, stats,
                                                      stats + C, stats + 2 * C, stats + 3 * C);
  if (num_rows > 0) {
    BnGroups one;
    bn_make_groups(&one, num_rows, 1, 0, 0);
    k_bn_relu_apply<<<bn_apply_grid(num_rows, C), 256, 0, st>>>(x, ldx, y, ldy, one, C, stats, leak, res, ld_res, act);
  }
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

__global__ __launch_bounds__(256) void k_bn_bwd_local_sums(const float* __restrict__ partial, int nblk, int C, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int accumulate, double* __restrict__ sums) {
  const int c = blockIdx.x;
  double s, ss;
  bn_block_sum2(partial, nblk, C, c, s, ss);
  if (threadIdx.x == 0) {
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)ss;
    sums[c] = s;
    sums[C + c] = ss;
  }
}
__global__ void k_bn_bwd_coef_sync(const double* __restrict__ sums, const double* __restrict__ gathered, int world, int C,
                                   float* __restrict__ coef) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * C) return;
  double n = 0.0;   // global row count: the n entries of the forward pass's gathered moments (no host round trip)
  for (int r = 0; r < world; ++r) n += gathered[(int64_t)r * (2 * C + 1) + 2 * C];
  coef[i] = (float)(sums[i] / n);
}

// sums: 2C doubles (sum dz, sum dz*xhat over this rank's rows); dgamma / dbeta receive the LOCAL sums.
MOPA_API int mopa_bn_sync_bwd_sums(const float* dy, int32_t ld_dy, const float* x, int32_t ldx, int32_t num_rows, int32_t C,
                                   const float* stats, float leak, int32_t act, const float* ymask, int32_t ld_ym, float* dgamma,
                                   float* dbeta, int32_t accumulate_param_grads, double* sums, void* ws, size_t ws_bytes, void* stream) {
  if (num_rows <= 0 || C <= 0 || (C & 3) || C > 1024 || ldx < C || ld_dy < C || ((ldx | ld_dy) & 3) || !sums) return MOPA_ERR_ARG;
  if (ymask && (ld_ym < C || (ld_ym & 3))) return MOPA_ERR_ARG;
  if (ws_bytes < mopa_bnrelu_rows_workspace_bytes(num_rows, C)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bn_num_blocks(num_rows), RL = 256 / (C >> 2);
  float* partial = (float*)ws;
  BnGroups one;
  bn_make_groups(&one, num_rows, 1, 0, 0);
  k_bn_bwd_partial<<<nblk, 256, (size_t)2 * RL * C * sizeof(float), st>>>(dy, ld_dy, x, ldx, C, stats, leak, ymask, ld_ym, act, one, partial);
  k_bn_bwd_local_sums<<<C, 256, 0, st>>>(partial, nblk, C, dgamma, dbeta, accumulate_param_grads, sums);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// sums_global: the all-reduced (SUM) 2C doubles; gathered / world: the forward pass's moments (their n entries give the global row
// count).  coef_ws: 2C floats of scratch.
MOPA_API int mopa_bn_act_bwd_sync(const float* dy, int32_t ld_dy, const float* x, int32_t ldx, float* dx, int32_t ld_dx, int32_t num_rows,
                                  int32_t C, const float* stats, float leak, int32_t act, const float* ymask, int32_t ld_ym, float* dres,
                                  int32_t ld_dres, int32_t accumulate_dres, const double* sums_global, const double* gathered,
                                  int32_t world, int32_t accumulate_dx, float* coef_ws, void* stream) {
  if (num_rows <= 0 || C <= 0 || (C & 3) || C > 1024 || ldx < C || ld_dy < C || ld_dx < C || ((ldx | ld_dy | ld_dx) & 3) || world <= 0 ||
      !sums_global || !gathered || !coef_ws)
    return MOPA_ERR_ARG;
  if ((ymask && (ld_ym < C || (ld_ym & 3))) || (dres && (ld_dres < C || (ld_dres & 3)))) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  k_bn_bwd_coef_sync<<<(2 * C + 255) / 256, 256, 0, st>>>(sums_global, gathered, world, C, coef_ws);
  BnGroups one;
  bn_make_groups(&one, num_rows, 1, 0, 0);
  k_bn_bwd_apply<<<bn_apply_grid(num_rows, C), 256, 0, st>>>(dy, ld_dy, x, ldx, dx, ld_dx, one, C, stats, coef_ws, leak, 1,
                                                             accumulate_dx, ymask, ld_ym, act, dres, ld_dres, accumulate_dres);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ InputLayer mode 4
// out[row][c] = mean over the row's points (increasing point index) of feats[p][c]; columns [cin, ld) zero-filled.
__global__ void k_input_layer_fwd(const float* __restrict__ feats, int cin, const int* __restrict__ row_start,
                                  const int* __restrict__ row_points, int A, float* __restrict__ out, int ld) {
  const int64_t total = (int64_t)A * ld;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / ld), c = (int)(i - (int64_t)row * ld);
    float v = 0.f;
    if (c < cin) {
      const int s = row_start[row], e = row_start[row + 1];
      for (int k = s; k < e; ++k) v += feats[(int64_t)row_points[k] * cin + c];
      v /= (float)(e - s);
    }
    out[i] = v;
  }
}

// out[r][c] = a[r][c] + b[r][c] over [rows][C] slices of wider buffers (scn.AddTable of the ResNet-style UNet blocks,
// SURVEY A.7 / oracle/scn3d.py::unet_forward(residual_blocks=True)); C % 4 == 0, 16-byte aligned rows.
__global__ void k_rows_add(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, float* __restrict__ out, int ldo,
                           int rows, int C4) {
  const int64_t total = (int64_t)rows * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / C4), c = (int)(i - (int64_t)r * C4) * 4;
    const float4 x = *reinterpret_cast<const float4*>(a + (int64_t)r * lda + c), y = *reinterpret_cast<const float4*>(b + (int64_t)r * ldb + c);
    *reinterpret_cast<float4*>(out + (int64_t)r * ldo + c) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}
MOPA_API int mopa_rows_add(const float* a, int32_t lda, const float* b, int32_t ldb, float* out, int32_t ldo, int32_t rows, int32_t C,
                           void* stream) {
  if (rows <= 0 || C <= 0 || C % 4 || lda % 4 || ldb % 4 || ldo % 4 || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15)) return MOPA_ERR_ARG;
  k_rows_add<<<stream_grid((int64_t)rows * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(a, lda, b, ldb, out, ldo, rows, C / 4);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

MOPA_API int mopa_input_layer_fwd(const float* feats, int32_t cin, const int32_t* row_start, const int32_t* row_points,
                                  int32_t num_rows, float* out, int32_t ld_out, void* stream) {
  if (cin <= 0 || num_rows <= 0 || ld_out < cin) return MOPA_ERR_ARG;
  k_input_layer_fwd<<<stream_grid((int64_t)num_rows * ld_out, 256), 256, 0, (hipStream_t)stream>>>(
      feats, cin, row_start, row_points, num_rows, out, ld_out);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// dfeats[p][c] = dout[row(p)][c] / count(row(p))
__global__ void k_input_layer_bwd(const float* __restrict__ dout, int ld, const int* __restrict__ point_row,
                                  const int* __restrict__ row_start, int N, int cin, float* __restrict__ dfeats) {
  const int64_t total = (int64_t)N * cin;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i / cin), c = (int)(i - (int64_t)p * cin);
    const int row = point_row[p];
    dfeats[i] = dout[(int64_t)row * ld + c] / (float)(row_start[row + 1] - row_start[row]);
  }
}

MOPA_API int mopa_input_layer_bwd(const float* dout, int32_t ld_dout, const int32_t* point_row, const int32_t* row_start,
                                  int32_t n_points, int32_t cin, float* dfeats, void* stream) {
  if (cin <= 0 || n_points <= 0 || ld_dout < cin) return MOPA_ERR_ARG;
  k_input_layer_bwd<<<stream_grid((int64_t)n_points * cin, 256), 256, 0, (hipStream_t)stream>>>(
      dout, ld_dout, point_row, row_start, n_points, cin, dfeats);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ------------------------------------------------------------------------------------------ OutputLayer + heads
// feats[p] = y[point_row[p]] (A.3);  logit_h[p] = feats[p] @ W_h^T + b_h for h = 1,2 (xmuda_arch.py:116,124).
// 16 lanes per point (lane = feature channel group); M = feature width (multiple of 4, <= 64), NC <= 32 classes.
// M/4 lanes cooperate on one point: each lane moves one float4 of the row (the gather reads the row once, coalesced,
// and writes feats once) and holds the partial dot products of its 4 channels with every class row of W1 / W2; a
// shuffle tree over the lane group finishes the logits.  The lane group is M/4 rounded up to a power of two <= 16 (M = 16: SCN,
// M = 64: 2D; other multiples of 4 -- UNetSCN(m) with m = 8, 12, 20, ... -- leave the group's last lanes idle).
#define OH_MAXNC 16
__global__ __launch_bounds__(256) void k_output_heads_fwd(const float* __restrict__ y, int ld, const int* __restrict__ point_row,
                                                           int N, int M, int NC, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, float* __restrict__ feats,
                                                           float* __restrict__ logit1, float* __restrict__ logit2) {
  extern __shared__ float lw[];  // w1[NC][M] | b1[NC] | w2[NC][M] | b2[NC]
  const int nw = NC * M;
  for (int i = threadIdx.x; i < nw; i += 256) { lw[i] = w1[i]; if (w2) lw[nw + NC + i] = w2[i]; }
  for (int i = threadIdx.x; i < NC; i += 256) { lw[nw + i] = b1[i]; if (w2) lw[2 * nw + NC + i] = b2[i]; }
  __syncthreads();
  const int MQ = M >> 2;             // float4 pieces of a row
  int MQP = 1;                       // lanes per point: MQ rounded up to a power of two (the shuffle tree's width)
  while (MQP < MQ) MQP <<= 1;
  const int PPB = 256 / MQP;         // points per block iteration
  const int cq = threadIdx.x % MQP, pl = threadIdx.x / MQP;
  const bool lane_on = cq < MQ;
  for (int p0 = blockIdx.x * PPB; p0 < N; p0 += gridDim.x * PPB) {
    const int p = p0 + pl;
    const bool ok = p < N;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && lane_on) {
      v = *reinterpret_cast<const float4*>(y + (int64_t)point_row[p] * ld + cq * 4);
      *reinterpret_cast<float4*>(feats + (int64_t)p * M + cq * 4) = v;
    }
#pragma unroll
    for (int k = 0; k < OH_MAXNC; ++k) {
      if (k < NC) {
        const float* wr = lw + k * M + (lane_on ? cq : 0) * 4;   // (an idle lane multiplies zeros)
        float a1 = fmaf(v.x, wr[0], fmaf(v.y, wr[1], fmaf(v.z, wr[2], v.w * wr[3])));
        float a2 = 0.f;
        if (w2) {
          const float* wr2 = lw + nw + NC + k * M + (lane_on ? cq : 0) * 4;
          a2 = fmaf(v.x, wr2[0], fmaf(v.y, wr2[1], fmaf(v.z, wr2[2], v.w * wr2[3])));
        }
        for (int o = MQP >> 1; o > 0; o >>= 1) {
          a1 += __shfl_xor(a1, o, 64);
          a2 += __shfl_xor(a2, o, 64);
        }
        if (ok && cq == 0) {
          logit1[(int64_t)p * NC + k] = a1 + lw[nw + k];
          if (w2) logit2[(int64_t)p * NC + k] = a2 + lw[2 * nw + NC + k];
        }
      }
    }
  }
}

MOPA_API int mopa_output_layer_heads_fwd(const float* y, int32_t ld_y, const int32_t* point_row, int32_t n_points,
                                         int32_t M, int32_t num_classes, const float* w1, const float* b1,
                                         const float* w2, const float* b2, float* feats, float* logit1, float* logit2,
                                         void* stream) {
  if (n_points <= 0 || M <= 0 || M > 64 || (M & 3) || num_classes <= 0 || num_classes > OH_MAXNC || ld_y < M || (ld_y & 3))
    return MOPA_ERR_ARG;
  int mq = 1;
  while (mq < (M >> 2)) mq <<= 1;
  const size_t sh = (size_t)(2 * num_classes * M + 2 * num_classes) * sizeof(float);
  k_output_heads_fwd<<<stream_grid((int64_t)n_points * mq, 256), 256, sh, (hipStream_t)stream>>>(
      y, ld_y, point_row, n_points, M, num_classes, w1, b1, w2, b2, feats, logit1, logit2);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// Backward to the voxel rows: dy[row][c] = sum_{p in row} ( dfeats[p][c] + sum_k dl1[p][k] W1[k][c] + dl2[p][k] W2[k][c] ).
// Any of dfeats / dl1 / dl2 may be null.  Deterministic: points of a row are visited in increasing index (CSR).
__global__ __launch_bounds__(256) void k_output_heads_bwd_rows(const float* __restrict__ dfeats, const float* __restrict__ dl1,
                                                                const float* __restrict__ dl2, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, const int* __restrict__ row_start,
                                                                const int* __restrict__ row_points, int A, int M, int NC,
                                                                float* __restrict__ dy, int ld) {
  // thread = (row, channel quad): one float4 of the row (round 1-3: one thread per ELEMENT -- 64 threads of a 2D row each read the
  // row's CSR bounds and every class gradient of its points; 1.1 TB/s at 2.3 M mostly empty pixel rows).  HB_UN rows per thread and
  // iteration, their CSR bounds loaded first.  (Measured and not kept: also prefetching the first point and its gradients of every
  // row before the arithmetic -- three quarters of the pixel rows are empty, the unconditional loads cost more than the chain: 385 ->
  // 580 us.  What is left is the dependent chain bounds -> point -> gradients of the rows that do have a point.)
  extern __shared__ float lw[];  // w1[NC][M] | w2[NC][M]
  const int nw = NC * M;
  for (int i = threadIdx.x; i < nw; i += 256) { lw[i] = w1[i]; lw[nw + i] = w2 ? w2[i] : 0.f; }
  __syncthreads();
  const int MQ = M >> 2;
  const int64_t total = (int64_t)A * MQ;
  constexpr int HB_UN = 4;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < total; i0 += step * HB_UN) {
    int ka[HB_UN], kb[HB_UN];
#pragma unroll
    for (int u = 0; u < HB_UN; ++u) {
      const int64_t i = i0 + u * step;
      const int row = (int)((i < total ? i : total - 1) / MQ);
      ka[u] = row_start[row];
      kb[u] = row_start[row + 1];
    }
#pragma unroll
    for (int u = 0; u < HB_UN; ++u) {
      const int64_t i = i0 + u * step;
      if (i >= total) break;
      const int row = (int)(i / MQ), cq = (int)(i - (int64_t)row * MQ);
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int k = ka[u]; k < kb[u]; ++k) {
        const int p = row_points[k];
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (dfeats) {
          const float4 d = *reinterpret_cast<const float4*>(dfeats + (int64_t)p * M + cq * 4);
          v[0] = d.x; v[1] = d.y; v[2] = d.z; v[3] = d.w;
        }
        if (dl1)
          for (int j = 0; j < NC; ++j) {
            const float g = dl1[(int64_t)p * NC + j];
            const float* wr = lw + j * M + cq * 4;
            v[0] = fmaf(g, wr[0], v[0]); v[1] = fmaf(g, wr[1], v[1]); v[2] = fmaf(g, wr[2], v[2]); v[3] = fmaf(g, wr[3], v[3]);
          }
        if (dl2)
          for (int j = 0; j < NC; ++j) {
            const float g = dl2[(int64_t)p * NC + j];
            const float* wr = lw + nw + j * M + cq * 4;
            v[0] = fmaf(g, wr[0], v[0]); v[1] = fmaf(g, wr[1], v[1]); v[2] = fmaf(g, wr[2], v[2]); v[3] = fmaf(g, wr[3], v[3]);
          }
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
      }
      *reinterpret_cast<float4*>(dy + (int64_t)row * ld + cq * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

// Head parameter grads: dW[k][c] = sum_p dl[p][k] * feats[p][c], db[k] = sum_p dl[p][k].  Thread = (point lane,
// channel quad); block partials [nblk][NC][M+1] then an ordered reduction (deterministic).
#define HEAD_PTS_PER_BLOCK 1024
#define HEAD_MAXNC 32
template <int NCM>   // NCM: compile-time bound of the class count (8 or NCM): the accumulators are NCM x 4 registers
__global__ __launch_bounds__(256) void k_head_wgrad_partial(const float* __restrict__ dl, const float* __restrict__ feats, int N,
                                                             int M, int NC, float* __restrict__ partial) {
  extern __shared__ float red[];  // [PL][NC][M+1]
  const int MQ = M >> 2;
  int MQP = 1;   // lanes per point (M/4 rounded up to a power of two; the last lanes of a group idle when M/4 is not one)
  while (MQP < MQ) MQP <<= 1;
  const int PL = 256 / MQP;
  const int cq = threadIdx.x % MQP, pl = cq < MQ ? threadIdx.x / MQP : PL;
  const int p0 = blockIdx.x * HEAD_PTS_PER_BLOCK, p1 = min(N, p0 + HEAD_PTS_PER_BLOCK);
  float acc[NCM][4];
  float accb[NCM];
#pragma unroll
  for (int k = 0; k < NCM; ++k) { acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f; accb[k] = 0.f; }
  if (pl < PL) {
    // four points in flight per thread (their loads are independent; the sums keep the order p, p + PL, ...): one point per trip was a
    // chain of 64 dependent load latencies per thread, 260 us per call on 558,080 points
    int p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {
      float4 v[4];
      float g[4][NCM];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *reinterpret_cast<const float4*>(feats + (int64_t)(p + u * PL) * M + cq * 4);
#pragma unroll
        for (int k = 0; k < NCM; ++k) g[u][k] = k < NC ? dl[(int64_t)(p + u * PL) * NC + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < NCM; ++k)
          if (k < NC) {
            acc[k][0] = fmaf(g[u][k], v[u].x, acc[k][0]); acc[k][1] = fmaf(g[u][k], v[u].y, acc[k][1]);
            acc[k][2] = fmaf(g[u][k], v[u].z, acc[k][2]); acc[k][3] = fmaf(g[u][k], v[u].w, acc[k][3]);
            if (cq == 0) accb[k] += g[u][k];
          }
    }
    for (; p < p1; p += PL) {
      const float4 v = *reinterpret_cast<const float4*>(feats + (int64_t)p * M + cq * 4);
#pragma unroll
      for (int k = 0; k < NCM; ++k)
        if (k < NC) {
          const float g = dl[(int64_t)p * NC + k];
          acc[k][0] = fmaf(g, v.x, acc[k][0]); acc[k][1] = fmaf(g, v.y, acc[k][1]);
          acc[k][2] = fmaf(g, v.z, acc[k][2]); acc[k][3] = fmaf(g, v.w, acc[k][3]);
          if (cq == 0) accb[k] += g;
        }
    }
  }
  const int stride = NC * (M + 1);
  if (pl < PL) {
#pragma unroll
    for (int k = 0; k < NCM; ++k)
      if (k < NC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl * stride + k * (M + 1) + cq * 4 + j] = acc[k][j];
        if (cq == 0) red[pl * stride + k * (M + 1) + M] = accb[k];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < stride; i += 256) {
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * stride + i];
    partial[(int64_t)blockIdx.x * stride + i] = s;
  }
}
__global__ __launch_bounds__(256) void k_head_wgrad_reduce(const float* __restrict__ partial, int nblk, int M, int NC,
                                                            float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  const int nout = NC * (M + 1);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wv; i < nout; i += gridDim.x * 4) {
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)partial[(int64_t)b * nout + i];
    s = wave_sum_d(s);
    if (lane == 0) {
      const int k = i / (M + 1), c = i - k * (M + 1);
      float* dst = (c < M) ? &dw[k * M + c] : &db[k];
      *dst = (accumulate ? *dst : 0.f) + (float)s;
    }
  }
}

MOPA_API size_t mopa_output_layer_heads_bwd_workspace_bytes(int32_t n_points, int32_t M, int32_t num_classes) {
  return align_up((size_t)cdiv64(n_points, HEAD_PTS_PER_BLOCK) * num_classes * (M + 1) * sizeof(float), 256);
}

// feats: the forward's per-point features (N,M).  dfeats/dl1/dl2 may be null (no upstream grad on that output).
MOPA_API int mopa_output_layer_heads_bwd(const float* dfeats, const float* dl1, const float* dl2, const float* feats,
                                         const float* w1, const float* w2, const int32_t* row_start,
                                         const int32_t* row_points, int32_t num_rows, int32_t n_points, int32_t M,
                                         int32_t num_classes, float* dy, int32_t ld_dy, float* dw1, float* db1,
                                         float* dw2, float* db2, int32_t accumulate, void* ws, size_t ws_bytes,
                                         void* stream) {
  if (n_points <= 0 || num_rows <= 0 || M <= 0 || M > 64 || (M & 3) || num_classes <= 0 || num_classes > HEAD_MAXNC || ld_dy < M ||
      (ld_dy & 3) || (((uintptr_t)dy | (uintptr_t)dfeats) & 15))
    return MOPA_ERR_ARG;
  int mqp = 1;
  while (mqp < (M >> 2)) mqp <<= 1;
  if (ws_bytes < mopa_output_layer_heads_bwd_workspace_bytes(n_points, M, num_classes)) return MOPA_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  k_output_heads_bwd_rows<<<stream_grid(cdiv64((int64_t)num_rows * (M >> 2), 4), 256), 256, (size_t)2 * num_classes * M * sizeof(float), st>>>(
      dfeats, dl1, dl2, w1, w2, row_start, row_points, num_rows, M, num_classes, dy, ld_dy);
  const int nblk = (int)cdiv64(n_points, HEAD_PTS_PER_BLOCK);
  float* partial = (float*)ws;
  const size_t hsh = (size_t)(256 / mqp) * num_classes * (M + 1) * sizeof(float);
  if (hsh > 64 * 1024) return MOPA_ERR_ARG;
  if (dl1 && dw1) {
    if (num_classes <= 8) k_head_wgrad_partial<8><<<nblk, 256, hsh, st>>>(dl1, feats, n_points, M, num_classes, partial);
    else k_head_wgrad_partial<HEAD_MAXNC><<<nblk, 256, hsh, st>>>(dl1, feats, n_points, M, num_classes, partial);
    k_head_wgrad_reduce<<<16, 256, 0, st>>>(partial, nblk, M, num_classes, dw1, db1, accumulate);
  }
  if (dl2 && dw2) {
    if (num_classes <= 8) k_head_wgrad_partial<8><<<nblk, 256, hsh, st>>>(dl2, feats, n_points, M, num_classes, partial);
    else k_head_wgrad_partial<HEAD_MAXNC><<<nblk, 256, hsh, st>>>(dl2, feats, n_points, M, num_classes, partial);
    k_head_wgrad_reduce<<<16, 256, 0, st>>>(partial, nblk, M, num_classes, dw2, db2, accumulate);
  }
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}

// ----------------------------------------------------------------------------------------------
// Three byte movers so that a recorded pass (exec2d.hip) holds no launch that is not this library's own: zero a channel slice,
// copy a channel slice, add a constant to up to 64 int64 scalars (BatchNorm2d.num_batches_tracked of every layer that ran).
MOPA_API int mopa_zero_rows(float* x, int32_t ld, int64_t rows, int32_t C, void* stream) {
  if (!x || rows <= 0 || C <= 0 || ld < C) return MOPA_ERR_ARG;
  if (hipMemset2DAsync(x, (size_t)ld * 4, 0, (size_t)C * 4, (size_t)rows, (hipStream_t)stream) != hipSuccess) return MOPA_ERR_LAUNCH;
  return MOPA_OK;
}
MOPA_API int mopa_copy_rows(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int64_t rows, int32_t C, void* stream) {
  if (!src || !dst || rows <= 0 || C <= 0 || ld_src < C || ld_dst < C) return MOPA_ERR_ARG;
  if (hipMemcpy2DAsync(dst, (size_t)ld_dst * 4, src, (size_t)ld_src * 4, (size_t)C * 4, (size_t)rows, hipMemcpyDeviceToDevice,
                       (hipStream_t)stream) != hipSuccess)
    return MOPA_ERR_LAUNCH;
  return MOPA_OK;
}
struct I64Ptrs { int64_t* p[64]; };
__global__ void k_add_i64_many(const I64Ptrs d, int n, int64_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) *d.p[i] += v;
}
MOPA_API int mopa_add_i64_many(const int64_t* ptrs_host, int32_t n, int64_t value, void* stream) {
  if (!ptrs_host || n <= 0 || n > 64) return MOPA_ERR_ARG;
  I64Ptrs d;
  for (int i = 0; i < 64; ++i) d.p[i] = reinterpret_cast<int64_t*>(ptrs_host[i < n ? i : 0]);
  k_add_i64_many<<<1, 64, 0, (hipStream_t)stream>>>(d, n, value);
  MOPA_CHECK_LAUNCH();
  return MOPA_OK;
}
