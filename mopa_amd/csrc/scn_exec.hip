// Native executor of the 3D branch's layer program: the whole UNetSCN forward (InputLayer -> convolutions / BatchNormReLU /
// AddTable -> OutputLayer + linear heads) and the whole backward pass are ONE C-ABI call each.
//
// Why: the program is static (scn.Sequential of mopa/models/scn_unet.py:25-30 with scn.UNet unrolled, pinned by fixture G6),
// but round 2 walked it from Python -- ~420 kernel launches per training step through ctypes, 5.2 ms of host time around 6.4 ms
// of kernels: the 3D-only step followed the host's load, not the GPU's.  Here Python hands over tables (built once per
// model: ops, parameter pointers; once per geometry: rule tables, row counts; once per pass: buffer addresses) and this file
// walks them, calling the same kernels' entry points directly.  It also owns the derived weight forms (packed for the
// pipelined kernels / transposed for backward-data): all stale forms of a pass are rebuilt in one launch.
//
// All arguments whose name ends in _host are HOST arrays; every other pointer is a device pointer.  No allocation; the only
// state is the caller's `forms_host` array (which weight forms have been built, for which column-group width, at which
// weight epoch) -- it lives in caller memory, the library keeps nothing.
#include "common.h"
#pragma GCC visibility push(default)
#include "../../include/mopa_hip.h"   // the other translation units' entry points (this file only calls them)
#pragma GCC visibility pop
#include <string.h>

// ---- table layouts (shared with mopa_amd/sparse3d.py::NativeProgram)
// prog_host   int32 [n_ops][SCN_OP_W]:
enum { OP_KIND = 0, OP_CKIND, OP_LSRC, OP_LDST, OP_SBUF, OP_SCOL, OP_SC, OP_DBUF, OP_DCOL, OP_DC, OP_ABUF, OP_ACOL, SCN_OP_W };
enum { K_BN = 0, K_CONV = 1, K_ADD = 2 };
enum { C_SUBM = 0, C_DOWN = 1, C_UP = 2, C_NIN = 3 };
// params_host int64 [n_ops][4]: BatchNorm gamma, beta, running_mean, running_var | convolution weight [K][Cin][Cout], -, -, -
// forms_host  int64 [n_ops][2][3]: per convolution and pass (0 forward, 1 backward-data): buffer (K*Cin*Cout floats), column-group
//             width the buffer was packed for (0 = plain per-offset transpose, -1 = never built), weight epoch it was built at
// geom_host   int64: [0] levels L, [1] n_points, [2] point_row, [3] row_start, [4] row_points, [5] grp_o, [6] grp_in, [7] grp_out,
//             then per level l at 8 + 8 l: rows A_l, nbr27, its grp_start, ch (rows A_{l+1}), its grp_start, up (rows A_l), its grp_start,
//             split_l (0: one BatchNorm group; s: rows [0, s) and [s, A_l) are groups of scans whose BatchNorm statistics, running
//             updates and gradients are computed separately, first group first -- mopa_amd/sparse3d.py::Geometry3D.split);
//             behind the L + 1 level rows: a tail of 8 values, tail[l] = second split of level l (0: none) -- three groups;
//             behind the tail: 3 values per level l (up to 8 levels), the run-major rulebooks (sprun.hip) of nbr27 / ch / up at that
//             level, 0 = none built (the offset-major path then does not run on that table)
enum { G_L = 0, G_NPTS, G_PROW, G_RSTART, G_RPTS, G_GO, G_GI, G_GOUT, G_LEVELS = 8, G_LW = 8 };
enum { GL_A = 0, GL_NBR, GL_NBR_GS, GL_CH, GL_CH_GS, GL_UP, GL_UP_GS, GL_SPLIT };
// bufs_host   int64 [nbufs][2]: base pointer, row stride (floats) of the activation (or gradient) buffers
// io_host     int64: see IO_* below (doubles travel as their bit patterns)
enum { IO_TRAINING = 0, IO_EPOCH, IO_FEATS, IO_CIN, IO_X0BUF, IO_X0COL, IO_OUTBUF, IO_OUTCOL, IO_M, IO_NCLS, IO_W1, IO_B1, IO_W2, IO_B2,
       IO_OFEATS, IO_L1, IO_L2, IO_STATS, IO_MOMENTUM, IO_EPS, IO_LEAK, IO_DFEATS_OUT, IO_DL1, IO_DL2, IO_DFEATS_IN, IO_DW1, IO_DB1,
       IO_DW2, IO_DB2, IO_HEADS_ACC,
       // backward only, all optional (0 = the weight gradients run on `stream` like everything else): a second stream for the weight
       // gradients (they depend on a layer's input and output gradient only, nothing in the pass waits for them), its own scratch,
       // and two hipEvent_t the caller created: "output gradient ready" (recorded on `stream` per convolution) and "weight gradients
       // done" (recorded on the second stream at the end; `stream` waits for it before the call returns its work to the caller)
       IO_WSTREAM, IO_WS2, IO_WS2_BYTES, IO_EV_READY, IO_EV_DONE, IO_N };
// plan_host   int32 [n_steps][PL_W] (backward): step kind, op index, where the output gradient is read, where the input gradient goes
enum { PL_KIND = 0, PL_OP, PL_DYBUF, PL_DYCOL, PL_DYC, PL_DXBUF, PL_DXCOL, PL_DXC, PL_ACC, PL_SKIPDX, PL_W };
// grads_host  int64 [n_ops][3]: gradient destinations (BatchNorm: dgamma, dbeta | convolution: dweight), accumulate flag

static inline double io_f(const int64_t* io, int i) {
  double d;
  memcpy(&d, &io[i], sizeof(d));
  return d;
}

struct View { float* p; int ld; int C; int rows; };
static inline View view_of(const int64_t* bufs, const int64_t* geom, int buf, int col, int C, int level) {
  View v;
  v.p = reinterpret_cast<float*>(bufs[buf * 2]) + col;
  v.ld = (int)bufs[buf * 2 + 1];
  v.C = C;
  v.rows = (int)geom[G_LEVELS + G_LW * level + GL_A];
  return v;
}

// row ranges of the BatchNorm groups at a level: (0, rows) | (0, s1), (s1, rows) | (0, s1), (s1, s2), (s2, rows)
#define SCN_MAX_GROUPS 3
static inline int bn_groups(const int64_t* geom, int level, int rows, int r0[SCN_MAX_GROUPS], int r1[SCN_MAX_GROUPS]) {
  const int s1 = (int)geom[G_LEVELS + G_LW * level + GL_SPLIT];
  const int s2 = (int)geom[G_LEVELS + G_LW * ((int)geom[G_L] + 1) + level];
  if (s1 <= 0 || s1 >= rows) { r0[0] = 0; r1[0] = rows; return 1; }
  r0[0] = 0; r1[0] = s1; r0[1] = s1;
  if (s2 <= s1 || s2 >= rows) { r1[1] = rows; return 2; }
  r1[1] = s2; r0[2] = s2; r1[2] = rows;
  return 3;
}
// slots of 4 C floats per BatchNorm in the statistics arena (the same for every layer of a pass)
static inline int n_bn_groups(const int64_t* geom) {
  const int L = (int)geom[G_L];
  int n = 1;
  for (int l = 0; l < L; ++l) {
    if (geom[G_LEVELS + G_LW * l + GL_SPLIT] > 0 && n < 2) n = 2;
    if (geom[G_LEVELS + G_LW * (L + 1) + l] > 0) n = 3;
  }
  return n;
}

struct Table { const int32_t* nbr; const int32_t* gs; int K; int rows_out; const int32_t* runs; int one; };
static inline const int32_t* runs_of(const int64_t* geom, int l, int which) {   // which: 0 nbr27, 1 ch, 2 up
  return (const int32_t*)geom[G_LEVELS + G_LW * ((int)geom[G_L] + 1) + 8 + 3 * l + which];
}
// the rule table a convolution of `ckind` between levels runs on (forward), or its reverse (backward-data)
static inline Table table_of(const int64_t* geom, int ckind, int l, bool reversed) {
  const int64_t* g = geom + G_LEVELS + G_LW * l;
  Table t;
  const int A_l = (int)g[GL_A], A_c = (int)g[G_LW + GL_A];   // (A_c is only read for down / up, where level l + 1 exists)
  t.runs = nullptr; t.one = 0;
  if (ckind == C_SUBM) { t.nbr = (const int32_t*)g[GL_NBR]; t.gs = (const int32_t*)g[GL_NBR_GS]; t.K = 27; t.rows_out = A_l; t.runs = runs_of(geom, l, 0); }
  else if (ckind == C_NIN) { t.nbr = (const int32_t*)g[GL_NBR] + (int64_t)13 * A_l; t.gs = nullptr; t.K = 1; t.rows_out = A_l; }
  else if ((ckind == C_DOWN) != reversed) { t.nbr = (const int32_t*)g[GL_CH]; t.gs = (const int32_t*)g[GL_CH_GS]; t.K = 8; t.rows_out = A_c; t.runs = runs_of(geom, l, 1); }
  else { t.nbr = (const int32_t*)g[GL_UP]; t.gs = (const int32_t*)g[GL_UP_GS]; t.K = 8; t.rows_out = A_l; t.runs = runs_of(geom, l, 2); t.one = 1; }   // every fine row has one parent
  return t;
}

// Which weight form the convolution kernel of (table, cin -> cout) wants: > 0 packed for column groups of that many 16-column
// tiles, 0 = the plain [K][cin][cout] weight of the convolution to run, SCN_RUN_FORM + nt = the run layout of the offset-major
// kernel (sprun.hip; nt = its column-group width).  Mirrors mopa_amd/sparse3d.py::spconv_fwd.
#define SCN_RUN_FORM 100
static inline int wanted_ntw(const Table& t, int cin, int cout, int ld_in) {
  if ((int64_t)t.rows_out * 8 * ld_in * 4 >= (1ll << 32)) return 0;
  if (t.runs && mopa_spconv_run_wanted(t.K, t.rows_out, cin, cout, t.one)) return SCN_RUN_FORM + mopa_spconv_run_form(cin, cout);
  if (!t.gs) return 0;
  return mopa_spconv_grouped_wants_packed(t.K, t.rows_out, cin, cout);
}

// The weight gradient of a convolution of `ckind` at level l: on the run lists where sprun.hip's dispatcher wants them -- the
// table's own run-major rulebook, or (stride-2 convolution: its forward table has none) the deconvolution table's with the two index
// lists swapped -- else on the dense table.  Mirrors mopa_amd/sparse3d.py::spconv_bwd_weight_of.
struct WgradPlan { Table t; int run; int swap; };
static inline WgradPlan wgrad_plan_of(const int64_t* geom, int ckind, int l, int cin, int cout) {
  WgradPlan w;
  w.t = table_of(geom, ckind, l, false);
  w.run = 0; w.swap = 0;
  if (ckind == C_NIN) return w;
  if (w.t.runs) {
    w.run = mopa_spconv_wgrad_run_wanted(w.t.K, w.t.rows_out, cin, cout, w.t.one);
    return w;
  }
  if (ckind == C_DOWN) {
    const Table r = table_of(geom, ckind, l, true);   // the deconvolution table: the same (coarse, fine, offset) triples
    if (r.runs && r.one && mopa_spconv_wgrad_run_wanted(r.K, r.rows_out, cin, cout, 1)) { w.t = r; w.run = 1; w.swap = 1; }
  }
  return w;
}
static inline size_t wgrad_workspace(const WgradPlan& w, int cin, int cout) {
  return w.run ? mopa_spconv_wgrad_run_workspace_bytes(w.t.K, w.t.rows_out, cin, cout, w.t.one)
               : mopa_spconv_wgrad_workspace_bytes(w.t.K, w.t.rows_out, cin, cout);
}
static inline int run_wgrad(const WgradPlan& w, const View& x, const View& dy, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  if (w.run)
    return mopa_spconv_bwd_weight_run(w.t.runs, w.t.K, w.t.rows_out, w.t.one, w.swap, x.p, x.ld, x.C, dy.p, dy.ld, dy.C, dw, accumulate, ws, ws_bytes, st);
  return mopa_spconv_bwd_weight(w.t.nbr, w.t.K, w.t.rows_out, x.p, x.ld, x.C, dy.p, dy.ld, dy.C, dw, accumulate, ws, ws_bytes, st);
}

// Make the weight form of (op, pass) valid for `ntw` at `epoch`; returns the pointer the kernel reads.  pass 0: forward
// (ntw 0 = the parameter itself), pass 1: backward-data (the convolution to run is the per-offset transpose).
static const float* form_ptr(const int64_t* params, int64_t* forms, int op, int pass, int ntw) {
  if (pass == 0 && ntw == 0) return reinterpret_cast<const float*>(params[op * 4]);
  return reinterpret_cast<const float*>(forms[(op * 2 + pass) * 3]);
}

// one launch for every stale form of the pass (rows: src, dst, K, cin, cout, flags as mopa_spconv_pack_weights_batched wants them)
static int refresh_forms(const int32_t* prog, int n_ops, const int64_t* params, int64_t* forms, const int64_t* geom, const int64_t* bufs,
                         const int64_t* gbufs, const int32_t* plan, int n_steps, int64_t epoch, int pass, hipStream_t st) {
  int64_t rows[64 * 6];
  int n = 0;
  auto flush = [&]() -> int {
    if (n == 0) return MOPA_OK;
    const int rc = mopa_spconv_pack_weights_batched(rows, n, st);
    n = 0;
    return rc;
  };
  const int count = pass == 0 ? n_ops : n_steps;
  for (int i = 0; i < count; ++i) {
    int op = i;
    if (pass == 1) {
      if (plan[i * PL_W + PL_KIND] != K_CONV || plan[i * PL_W + PL_SKIPDX]) continue;
      op = plan[i * PL_W + PL_OP];
    }
    const int32_t* o = prog + op * SCN_OP_W;
    if (o[OP_KIND] != K_CONV) continue;
    const int ck = o[OP_CKIND], l = o[OP_LSRC] < o[OP_LDST] ? o[OP_LSRC] : o[OP_LDST];
    const Table t = table_of(geom, ck, l, pass == 1);
    const int cin_w = o[OP_SC], cout_w = o[OP_DC];                 // the layer weight is [K][cin_w][cout_w]
    const int cin = pass == 0 ? cin_w : cout_w, cout = pass == 0 ? cout_w : cin_w;
    const int ld_in = pass == 0 ? (int)bufs[o[OP_SBUF] * 2 + 1] : (int)gbufs[plan[i * PL_W + PL_DYBUF] * 2 + 1];
    const int ntw = wanted_ntw(t, cin, cout, ld_in);
    if (pass == 0 && ntw == 0) continue;                           // forward on the parameter itself
    int64_t* f = forms + (op * 2 + pass) * 3;
    if (f[1] == ntw && f[2] == epoch) continue;
    if (!f[0]) return MOPA_ERR_ARG;
    int64_t* r = rows + n * 6;
    r[0] = params[op * 4]; r[1] = f[0]; r[2] = t.K; r[3] = cin_w; r[4] = cout_w;
    if (ntw >= SCN_RUN_FORM) r[5] = (pass == 1 ? 1 : 0) | ((ntw - SCN_RUN_FORM) << 8) | 0x10000;   // the run layout
    else r[5] = ntw > 0 ? ((pass == 1 ? 1 : 0) | (ntw << 8)) : 1;  // ntw 0 (backward-data only): plain per-offset transpose
    f[1] = ntw; f[2] = epoch;
    if (++n == 64) { const int rc = flush(); if (rc) return rc; }
  }
  return flush();
}

// out = conv(x) on table t with the weight form the plan asks for
static int run_conv(const Table& t, const View& x, const float* wk, int ntw, const View& out, int w_flip, const int64_t* geom, void* ws,
                    size_t ws_bytes, hipStream_t st) {
  const int32_t* go = (const int32_t*)geom[G_GO];
  const int32_t* gi = (const int32_t*)geom[G_GI];
  const int32_t* gout = (const int32_t*)geom[G_GOUT];
  const int64_t tiles = cdiv64(t.rows_out, 64);
  if (ntw >= SCN_RUN_FORM)
    return mopa_spconv_fwd_run(t.runs, t.K, t.rows_out, x.p, x.ld, x.C, wk, out.C, w_flip, out.p, out.ld, t.one, ws, ws_bytes, st);
  if (ntw > 0)
    return mopa_spconv_fwd_grouped(t.gs, go, gi, gout, t.K, t.rows_out, x.p, x.ld, x.C, wk, out.C, w_flip | 2, out.p, out.ld, nullptr, 0, st);
  if (t.gs && x.C > 4 && tiles < (t.K == 27 ? 1500 : 200)) {   // (<= 4 input channels: the stem kernel inside mopa_spconv_fwd)
    const size_t need = mopa_spconv_grouped_workspace_bytes(t.K, t.rows_out, out.C);
    return mopa_spconv_fwd_grouped(t.gs, go, gi, gout, t.K, t.rows_out, x.p, x.ld, x.C, wk, out.C, w_flip, out.p, out.ld, ws,
                                   ws_bytes >= need ? ws_bytes : 0, st);
  }
  return mopa_spconv_fwd(t.nbr, t.K, t.rows_out, x.p, x.ld, x.C, wk, out.C, w_flip, out.p, out.ld, st);
}

static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

// Scratch both passes need at most (BatchNorm partials, split-offset partial outputs, weight-gradient slabs, head gradients).
MOPA_API size_t mopa_scn_workspace_bytes(const int32_t* prog_host, int32_t n_ops, const int64_t* geom_host, int32_t num_classes, int32_t M) {
  if (!prog_host || !geom_host || n_ops <= 0) return 0;
  size_t need = 256;
  for (int i = 0; i < n_ops; ++i) {
    const int32_t* o = prog_host + i * SCN_OP_W;
    const int rows_s = (int)geom_host[G_LEVELS + G_LW * o[OP_LSRC] + GL_A];
    if (o[OP_KIND] == K_BN) need = max_sz(need, mopa_bnrelu_rows_bwd_workspace_bytes(rows_s, o[OP_SC]));
    else if (o[OP_KIND] == K_CONV) {
      const int l = o[OP_LSRC] < o[OP_LDST] ? o[OP_LSRC] : o[OP_LDST];
      for (int rev = 0; rev < 2; ++rev) {
        const Table t = table_of(geom_host, o[OP_CKIND], l, rev);
        const int cout = rev ? o[OP_SC] : o[OP_DC];
        need = max_sz(need, mopa_spconv_grouped_workspace_bytes(t.K, t.rows_out, cout));
        const int cin = rev ? o[OP_DC] : o[OP_SC];
        if (t.runs && !t.one && mopa_spconv_run_wanted(t.K, t.rows_out, cin, cout, 0))
          need = max_sz(need, mopa_spconv_run_workspace_bytes(t.K, t.rows_out, cout));
      }
      need = max_sz(need, wgrad_workspace(wgrad_plan_of(geom_host, o[OP_CKIND], l, o[OP_SC], o[OP_DC]), o[OP_SC], o[OP_DC]));
    }
  }
  need = max_sz(need, mopa_output_layer_heads_bwd_workspace_bytes((int)geom_host[G_NPTS], M, num_classes));
  return need;
}

MOPA_API int mopa_scn_forward(const int32_t* prog_host, int32_t n_ops, const int64_t* params_host, int64_t* forms_host,
                              const int64_t* geom_host, const int64_t* bufs_host, const int64_t* io_host, void* ws, size_t ws_bytes,
                              void* stream) {
  if (!prog_host || !params_host || !forms_host || !geom_host || !bufs_host || !io_host || n_ops <= 0) return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t* io = io_host;
  const int training = (int)io[IO_TRAINING];
  const float momentum = (float)io_f(io, IO_MOMENTUM), eps = (float)io_f(io, IO_EPS), leak = (float)io_f(io, IO_LEAK);
  float* stats = reinterpret_cast<float*>(io[IO_STATS]);
  const int NG = n_bn_groups(geom_host);   // the statistics arena holds NG slots of 4 C floats per BatchNorm
  int rc = refresh_forms(prog_host, n_ops, params_host, forms_host, geom_host, bufs_host, nullptr, nullptr, 0, io[IO_EPOCH], 0, st);
  if (rc) return rc;
  {   // InputLayer (mode 4: mean of the points of a voxel)
    const View x0 = view_of(bufs_host, geom_host, (int)io[IO_X0BUF], (int)io[IO_X0COL], (int)io[IO_CIN], 0);
    rc = mopa_input_layer_fwd((const float*)io[IO_FEATS], (int)io[IO_CIN], (const int32_t*)geom_host[G_RSTART],
                              (const int32_t*)geom_host[G_RPTS], x0.rows, x0.p, x0.ld, st);
    if (rc) return rc;
  }
  for (int i = 0; i < n_ops; ++i) {
    const int32_t* o = prog_host + i * SCN_OP_W;
    const View src = view_of(bufs_host, geom_host, o[OP_SBUF], o[OP_SCOL], o[OP_SC], o[OP_LSRC]);
    const View dst = view_of(bufs_host, geom_host, o[OP_DBUF], o[OP_DCOL], o[OP_DC], o[OP_LDST]);
    if (o[OP_KIND] == K_BN) {
      const int64_t* p = params_host + i * 4;
      int r0[SCN_MAX_GROUPS], r1[SCN_MAX_GROUPS];
      const int ng = bn_groups(geom_host, o[OP_LSRC], src.rows, r0, r1);
      // all groups of the layer in one set of launches (bit-identical to one call per row range: rows.hip::BnGroups)
      rc = mopa_bn_act_fwd_groups(src.p, src.ld, dst.p, dst.ld, src.rows, src.C, ng, r1[0], ng > 1 ? r1[1] : 0, (const float*)p[0],
                                  (const float*)p[1], (float*)p[2], (float*)p[3], momentum, eps, leak, 1, nullptr, 0, training,
                                  stats + (int64_t)NG * o[OP_ABUF], ws, ws_bytes, st);
    } else if (o[OP_KIND] == K_CONV) {
      const int l = o[OP_LSRC] < o[OP_LDST] ? o[OP_LSRC] : o[OP_LDST];
      const Table t = table_of(geom_host, o[OP_CKIND], l, false);
      const int ntw = wanted_ntw(t, src.C, dst.C, src.ld);
      rc = run_conv(t, src, form_ptr(params_host, forms_host, i, 0, ntw), ntw, dst, 0, geom_host, ws, ws_bytes, st);
    } else {
      const View b = view_of(bufs_host, geom_host, o[OP_ABUF], o[OP_ACOL], o[OP_DC], o[OP_LDST]);
      rc = mopa_rows_add(src.p, src.ld, b.p, b.ld, dst.p, dst.ld, dst.rows, dst.C, st);
    }
    if (rc) return rc;
  }
  const int M = (int)io[IO_M];
  const View y = view_of(bufs_host, geom_host, (int)io[IO_OUTBUF], (int)io[IO_OUTCOL], M, 0);
  return mopa_output_layer_heads_fwd(y.p, y.ld, (const int32_t*)geom_host[G_PROW], (int)geom_host[G_NPTS], M, (int)io[IO_NCLS],
                                     (const float*)io[IO_W1], (const float*)io[IO_B1], (const float*)io[IO_W2], (const float*)io[IO_B2],
                                     (float*)io[IO_OFEATS], (float*)io[IO_L1], (float*)io[IO_L2], st);
}

// The backward pass of mopa_scn_forward: head gradients, then `plan_host` (the program in reverse with the gradient-buffer
// slices resolved: mopa_amd/sparse3d.py::Program.backward_plan), then the InputLayer's gradient when io[IO_DFEATS_IN] is set.
// bufs_host: the forward's activation buffers (kept by the caller), gbufs_host: the gradient buffers mirroring them.
MOPA_API int mopa_scn_backward(const int32_t* prog_host, int32_t n_ops, const int32_t* plan_host, int32_t n_steps,
                               const int64_t* params_host, int64_t* forms_host, const int64_t* grads_host, const int64_t* geom_host,
                               const int64_t* bufs_host, const int64_t* gbufs_host, const int64_t* io_host, void* ws, size_t ws_bytes,
                               void* stream) {
  if (!prog_host || !plan_host || !params_host || !forms_host || !grads_host || !geom_host || !bufs_host || !gbufs_host || !io_host ||
      n_ops <= 0 || n_steps <= 0)
    return MOPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t* io = io_host;
  const int training = (int)io[IO_TRAINING];
  const float leak = (float)io_f(io, IO_LEAK);
  const float* stats = reinterpret_cast<const float*>(io[IO_STATS]);
  const int NG = n_bn_groups(geom_host);
  const int M = (int)io[IO_M], NC = (int)io[IO_NCLS];
  hipStream_t wst = (hipStream_t)io[IO_WSTREAM];
  hipEvent_t ev_ready = (hipEvent_t)io[IO_EV_READY], ev_done = (hipEvent_t)io[IO_EV_DONE];
  void* ws2 = (void*)io[IO_WS2];
  const size_t ws2_bytes = (size_t)io[IO_WS2_BYTES];
  bool side = wst && wst != st && ev_ready && ev_done && ws2;
  // The second stream reads a convolution's output gradient `dy` after waiting for "dy complete" only.  That is enough while nothing
  // later in the pass writes dy again -- true for the plain UNet, NOT for residual blocks: Program.backward_plan aliases an identity
  // shortcut's gradient to the AddTable's output gradient, and the BatchNorm behind it ACCUMULATES into that slice (PL_ACC) on
  // `stream` while a lagging weight gradient may still be reading it.  Programs with accumulating steps keep everything on `stream`.
  for (int s = 0; s < n_steps && side; ++s)
    if (plan_host[s * PL_W + PL_ACC]) side = false;
  int rc = refresh_forms(prog_host, n_ops, params_host, forms_host, geom_host, bufs_host, gbufs_host, plan_host, n_steps, io[IO_EPOCH], 1, st);
  if (rc) return rc;
  {
    const View dy = view_of(gbufs_host, geom_host, (int)io[IO_OUTBUF], (int)io[IO_OUTCOL], M, 0);
    rc = mopa_output_layer_heads_bwd((const float*)io[IO_DFEATS_OUT], (const float*)io[IO_DL1], (const float*)io[IO_DL2],
                                     (const float*)io[IO_OFEATS], (const float*)io[IO_W1], (const float*)io[IO_W2],
                                     (const int32_t*)geom_host[G_RSTART], (const int32_t*)geom_host[G_RPTS], dy.rows,
                                     (int)geom_host[G_NPTS], M, NC, dy.p, dy.ld, (float*)io[IO_DW1], (float*)io[IO_DB1],
                                     (float*)io[IO_DW2], (float*)io[IO_DB2], (int)io[IO_HEADS_ACC], ws, ws_bytes, st);
    if (rc) return rc;
  }
  for (int s = 0; s < n_steps; ++s) {
    const int32_t* pl = plan_host + s * PL_W;
    const int op = pl[PL_OP];
    const int32_t* o = prog_host + op * SCN_OP_W;
    const int64_t* g = grads_host + op * 3;
    const View x = view_of(bufs_host, geom_host, o[OP_SBUF], o[OP_SCOL], o[OP_SC], o[OP_LSRC]);
    const View dy = view_of(gbufs_host, geom_host, pl[PL_DYBUF], pl[PL_DYCOL], pl[PL_DYC], o[OP_LDST]);
    const View dx = view_of(gbufs_host, geom_host, pl[PL_DXBUF], pl[PL_DXCOL], pl[PL_DXC], o[OP_LSRC]);
    if (pl[PL_KIND] == K_BN) {
      int r0[SCN_MAX_GROUPS], r1[SCN_MAX_GROUPS];
      const int ng = bn_groups(geom_host, o[OP_LSRC], x.rows, r0, r1);
      // (the parameter gradients of the groups add up, in group order)
      rc = mopa_bn_act_bwd_groups(dy.p, dy.ld, x.p, x.ld, dx.p, dx.ld, x.rows, x.C, ng, r1[0], ng > 1 ? r1[1] : 0,
                                  stats + (int64_t)NG * o[OP_ABUF], leak, 1, nullptr, 0, nullptr, 0, 0, training, (float*)g[0], (float*)g[1],
                                  (int)g[2], pl[PL_ACC], ws, ws_bytes, st);
      if (rc) break;
      continue;
    }
    const int ck = o[OP_CKIND], l = o[OP_LSRC] < o[OP_LDST] ? o[OP_LSRC] : o[OP_LDST];
    const WgradPlan wg = wgrad_plan_of(geom_host, ck, l, x.C, dy.C);
    if (side) {   // dy is complete at this point of `stream`; nothing later in the pass writes it or the layer's input again
      if (hipEventRecord(ev_ready, st) != hipSuccess || hipStreamWaitEvent(wst, ev_ready, 0) != hipSuccess) return MOPA_ERR_LAUNCH;
      rc = run_wgrad(wg, x, dy, (float*)g[0], (int)g[2], ws2, ws2_bytes, wst);
    } else {
      rc = run_wgrad(wg, x, dy, (float*)g[0], (int)g[2], ws, ws_bytes, st);
    }
    if (rc) break;
    if (pl[PL_SKIPDX]) continue;
    // backward-data: the same kernel on the reversed rules with the per-offset transposed weight (submanifold: the table is its own
    // reverse with mirrored offsets; stride-2 convolution <-> deconvolution swap tables)
    const Table tr = table_of(geom_host, ck, l, true);
    const int ntw = wanted_ntw(tr, dy.C, dx.C, dy.ld);
    rc = run_conv(tr, dy, form_ptr(params_host, forms_host, op, 1, ntw), ntw, dx, ck == C_SUBM ? 1 : 0, geom_host, ws, ws_bytes, st);
    if (rc) break;
  }
  if (side && (hipEventRecord(ev_done, wst) != hipSuccess || hipStreamWaitEvent(st, ev_done, 0) != hipSuccess)) return MOPA_ERR_LAUNCH;
  if (rc) return rc;
  if (io[IO_DFEATS_IN]) {
    const View dx0 = view_of(gbufs_host, geom_host, (int)io[IO_X0BUF], (int)io[IO_X0COL], (int)io[IO_CIN], 0);
    rc = mopa_input_layer_bwd(dx0.p, dx0.ld, (const int32_t*)geom_host[G_PROW], (const int32_t*)geom_host[G_RSTART], (int)geom_host[G_NPTS],
                              (int)io[IO_CIN], (float*)io[IO_DFEATS_IN], st);
  }
  return rc;
}
