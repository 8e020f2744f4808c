"""Host side of the sparse 3D branch: device geometry (hash + rule tables) and the
UNetSCN forward/backward schedule over the C-ABI kernels of libmopa_hip.so.

What it replaces: the ``sparseconvnet`` calls behind ``mopa/models/scn_unet.py:25-34``
(InputLayer -> SubMConv -> scn.UNet -> BatchNormReLU -> OutputLayer) plus the two
``nn.Linear`` heads of ``mopa/models/xmuda_arch.py:114-126``.  Semantics: SURVEY.md
Appendix A; oracle: ``oracle/scn3d.py``.

PyTorch here is plumbing only (device buffers, streams, autograd hand-off): every
arithmetic step is a HIP kernel.  The whole network is ONE autograd node so that
JoinTable is free (layers write channel slices of a shared wide buffer) and no
per-layer Python autograd overhead is paid.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from ._lib import GradSink, call, ptr, query, stream, workspace

BN_EPS = 1e-4       # SCN BatchNormalization eps (Appendix A.6)
BN_MOMENTUM = 0.1   # SCN "momentum 0.9" == torch-style 0.1
LEAK = 0.0          # scn.UNet leakiness=0 / BatchNormReLU


def _ws(nbytes, device):
    return workspace.get(max(int(nbytes), 256), device)


def _pow2_at_least(n):
    p = 1
    while p < n:
        p <<= 1
    return p


# --------------------------------------------------------------------------------------- geometry
class Geometry3D:
    """Active sets + rule tables of one batch on the device (bit-exact with oracle.scn3d.Geometry).

    Built in two phases: (A) all levels' hashes/active sets chained on the device with
    worst-case buffers, (B) ONE host sync to read the 7 row counts, then exact-size rule tables.
    """

    def __init__(self, coords: torch.Tensor, num_levels: int = 7, full_scale: int = 4096, device=None):
        if coords.dim() != 2 or coords.shape[1] != 4:
            raise RuntimeError(f"coords must be (N,4) [x,y,z,batch], got {tuple(coords.shape)}")
        device = torch.device(device if device is not None else "cuda")
        N = int(coords.shape[0])
        if N == 0:
            raise RuntimeError("empty point cloud")
        coords = coords.to(torch.int64)
        if coords.device.type == "cpu":
            coords = _lib.upload(coords, device)
        else:
            coords = coords.to(device).contiguous()
        self.device, self.n_points, self.num_levels = device, N, num_levels
        L = num_levels
        cap = _pow2_at_least(2 * N)
        i64 = dict(dtype=torch.int64, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        keys = [torch.empty(N, **i64) for _ in range(L)]
        tk = [torch.empty(cap, **i64) for _ in range(L)]
        tv = [torch.empty(cap, **i32) for _ in range(L)]
        item_row = [torch.empty(N, **i32) for _ in range(L)]  # [0]: point->row0 ; [l>0]: parent of level l-1 rows
        meta = torch.zeros(L + 1, **i32)                       # counts[0..L-1], status
        wsb = query("mopa_voxel_hash_workspace_bytes", N)
        ws = _ws(wsb, device)
        st = stream()
        call("mopa_voxel_hash_build", ptr(coords), N, ptr(tk[0]), ptr(tv[0]), cap, ptr(item_row[0]), ptr(keys[0]),
             ptr(meta), ptr(meta, L), ptr(ws), ws.numel(), st)
        for l in range(L - 1):
            call("mopa_coarsen_build", ptr(keys[l]), N, ptr(meta, l), ptr(tk[l + 1]), ptr(tv[l + 1]), cap,
                 ptr(item_row[l + 1]), ptr(keys[l + 1]), ptr(meta, l + 1), ptr(ws), ws.numel(), st)
        m = meta.cpu().tolist()  # the one host sync of the geometry build
        if m[L] != 0:
            raise RuntimeError("voxel coordinates out of range: need 0 <= x,y,z < 4096 and batch >= 0")
        self.num_active = m[:L]
        A = self.num_active
        self.point_row = item_row[0]
        self.row_keys = [keys[l][:A[l]] for l in range(L)]
        self.parent = [item_row[l + 1][:A[l]] for l in range(L - 1)]
        self.nbr27, self.ch, self.up = [], [], []
        for l in range(L):
            nbr = torch.empty(27, A[l], **i32)
            call("mopa_rulebook_subm", ptr(keys[l]), A[l], ptr(tk[l]), ptr(tv[l]), cap, full_scale >> l, ptr(nbr), st)
            self.nbr27.append(nbr)
        for l in range(L - 1):
            ch = torch.empty(8, A[l + 1], **i32)
            up = torch.empty(8, A[l], **i32)
            call("mopa_rulebook_updown", ptr(keys[l]), ptr(item_row[l + 1]), A[l], A[l + 1], ptr(ch), ptr(up), st)
            self.ch.append(ch)
            self.up.append(up)
        # grouped rulebooks (MFMA-ready 16-rule groups per 64-row tile) for ALL tables at once: tiles and groups are numbered
        # globally -- one count launch, one scan, ONE more host sync for the group total, one fill launch.  Built once per
        # geometry, used by every layer's fwd and bwd-data; a table's rulebook = its slice of the scan + the shared arrays.
        self._rb = {}
        tables = list(self.nbr27) + list(self.ch) + list(self.up)
        tile0, desc = [0], []
        for t in tables:
            K, Ao = t.shape
            desc += [t.data_ptr(), K, Ao, tile0[-1]]
            tile0.append(tile0[-1] + (Ao + 63) // 64)
        ntile = tile0[-1]
        desc_h = torch.tensor(desc, dtype=torch.int64)   # host array: travels to the kernels as an argument
        tg = torch.empty(ntile, **i32)
        gs = torch.empty(ntile + 1, **i32)
        call("mopa_rulebook_groups_count_batched", desc_h.data_ptr(), len(tables), ntile, ptr(tg), st)
        ws = _ws(query("mopa_scan_workspace_bytes", ntile), device)
        call("mopa_scan_exclusive_i32", ptr(tg), ptr(gs), ntile, ptr(gs, ntile), ptr(ws), ws.numel(), st)
        ng = max(int(gs[-1].item()), 1)  # second (and last) host sync
        go = torch.empty(ng, **i32)
        gi = torch.empty(ng * 16, **i32)
        gout = torch.empty(ng * 16, **i32)
        call("mopa_rulebook_groups_fill_batched", desc_h.data_ptr(), len(tables), ntile, ptr(gs), ptr(go), ptr(gi), ptr(gout), st)
        for i, t in enumerate(tables):
            self._rb[t.data_ptr()] = (gs[tile0[i]:tile0[i + 1] + 1], go, gi, gout)
        self.row_start = torch.empty(A[0] + 1, **i32)
        self.row_points = torch.empty(N, **i32)
        wsb = query("mopa_points_csr_workspace_bytes", A[0])
        ws = _ws(wsb, device)
        call("mopa_points_csr", ptr(self.point_row), N, A[0], ptr(self.row_start), ptr(self.row_points), ptr(ws),
             ws.numel(), st)

    def rulebook(self, table: torch.Tensor):
        return self._rb.get(table.data_ptr())

    def tensors(self):
        """Every device tensor this geometry owns."""
        out = [self.point_row, self.row_start, self.row_points]
        for lst in (self.row_keys, self.parent, self.nbr27, self.ch, self.up):
            out += list(lst)
        for rb in self._rb.values():
            out += list(rb)
        return out

    def record_stream(self, stream):
        """The geometry was built on another stream than the one that will use it: tell the caching allocator (its memory must
        not return to the building stream's pool while `stream` still reads it)."""
        seen = set()
        for t in self.tensors():
            if t.data_ptr() not in seen and t.numel():
                seen.add(t.data_ptr())
                t.record_stream(stream)

    @property
    def num_rules(self):
        return [int((n >= 0).sum().item()) for n in self.nbr27]


# --------------------------------------------------------------------------------------- kernels (thin wrappers)
class View:
    """A [rows, C] fp32 activation living in columns [col, col+C) of a wider row-major buffer."""

    __slots__ = ("t", "col", "C")

    def __init__(self, t: torch.Tensor, col: int = 0, C: int | None = None):
        self.t, self.col, self.C = t, col, (t.shape[1] - col if C is None else C)

    @property
    def rows(self):
        return self.t.shape[0]

    @property
    def ld(self):
        return self.t.shape[1]

    @property
    def p(self):
        return ptr(self.t, self.col)

    def dense(self):
        return self.t[:, self.col:self.col + self.C]


def new_view(rows, C, device, ld=None):
    return View(torch.empty(rows, ld or C, dtype=torch.float32, device=device), 0, C)


def spconv_fwd(nbr: torch.Tensor, x: View, w: torch.Tensor, out: View, w_flip: bool = False, rb=None,
               w_transposed: bool = False):
    """out = sum_o x[nbr[o]] @ Wc[o] with Wc = w ([K][Cin][Cout]) or, for backward-data (`w_transposed`), the
    per-offset transpose of the layer weight w ([K][Cout][Cin]).  `rb` = the table's grouped rulebook
    (Geometry3D.rulebook(nbr)) selects the prefetching kernels; without it the kernel compacts the dense table on the
    fly.  The weight is re-laid out per call as the chosen kernel wants it (packed / transposed): one tiny kernel."""
    K, A_out = nbr.shape
    cin, cout = x.C, out.C
    assert out.rows == A_out and w.shape == ((K, cout, cin) if w_transposed else (K, cin, cout)), (nbr.shape, cin, cout, w.shape)
    ntw = query("mopa_spconv_grouped_wants_packed", K, A_out, cin, cout) if rb is not None else 0
    if A_out * 8 * x.ld * 4 >= 1 << 32:   # the pipelined kernels use 32-bit byte offsets into the input rows
        ntw = 0
    packed = ntw > 0
    if packed:   # column groups of ntw 16-column tiles, MFMA-operand order (mopa_spconv_pack_weight)
        wk = _weight_form(w, ("pack", int(w_transposed), ntw))
    else:
        wk = _weight_form(w, ("transpose",)) if w_transposed else w
    spconv_launch(nbr, x, wk, out, w_flip, rb, packed)


_weight_cache = {}


def _weight_form(w: torch.Tensor, form: tuple) -> torch.Tensor:
    """Packed / transposed form of a conv weight, re-used until the weight changes (the source and the target half of an
    iteration share the weights).  Same validity rules as dense2d.relayout_cached: one live tensor object, one weight
    version (autograd counter + _lib.WEIGHTS_EPOCH), one stream."""
    import weakref
    key = (id(w), form, stream())
    tag = (_lib.WEIGHTS_EPOCH[0], w._version, w.data_ptr())
    hit = _weight_cache.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is w:
        return hit[1]
    K = w.shape[0]
    if form[0] == "pack":
        t = torch.empty(w.numel(), dtype=w.dtype, device=w.device)
        call("mopa_spconv_pack_weight", ptr(w), K, w.shape[1], w.shape[2], form[1], form[2], ptr(t), stream())
    else:
        t = spconv_transpose_weight(w)
    if len(_weight_cache) > 4096:
        _weight_cache.clear()
    _weight_cache[key] = (tag, t, weakref.ref(w))
    return t


def spconv_launch(nbr: torch.Tensor, x: View, wk: torch.Tensor, out: View, w_flip: bool, rb, packed: bool):
    """The convolution launch itself, on a weight already laid out for the kernel that runs (bench.py times this)."""
    K, A_out = nbr.shape
    cin, cout = x.C, out.C
    # measured on MI355X (profiles/bench_spconv.py): 27-offset tables run the pipelined kernels on packed weights (one wave
    # per tile at 16 channels, four waves per tile and column group above); the 8-offset down/up tables the dense-table
    # wave kernel, except on the shortest levels (< 200 tiles) where the 4-wave block kernel with LDS-staged weights wins.
    if packed:
        gs, go, gi, gout = rb
        call("mopa_spconv_fwd_grouped", ptr(gs), ptr(go), ptr(gi), ptr(gout), K, A_out, x.p, x.ld, cin, ptr(wk), cout,
             int(w_flip) | 2, out.p, out.ld, 0, 0, stream())
    elif rb is not None and (A_out + 63) // 64 < (1500 if K == 27 else 200):
        gs, go, gi, gout = rb
        ws = _ws(query("mopa_spconv_grouped_workspace_bytes", K, A_out, cout), wk.device)
        call("mopa_spconv_fwd_grouped", ptr(gs), ptr(go), ptr(gi), ptr(gout), K, A_out, x.p, x.ld, cin, ptr(wk), cout,
             int(w_flip), out.p, out.ld, ptr(ws), ws.numel(), stream())
    else:
        call("mopa_spconv_fwd", ptr(nbr), K, A_out, x.p, x.ld, cin, ptr(wk), cout, int(w_flip), out.p, out.ld, stream())


def spconv_transpose_weight(w: torch.Tensor) -> torch.Tensor:
    K, cin, cout = w.shape
    wt = torch.empty(K, cout, cin, dtype=w.dtype, device=w.device)
    call("mopa_spconv_transpose_weight", ptr(w), K, cin, cout, ptr(wt), stream())
    return wt


def spconv_bwd_weight(nbr: torch.Tensor, x: View, dout: View, dw: torch.Tensor, accumulate: bool = False):
    K, A_out = nbr.shape
    assert dw.shape == (K, x.C, dout.C) and dout.rows == A_out
    wsb = query("mopa_spconv_wgrad_workspace_bytes", K, A_out, x.C, dout.C)
    ws = _ws(wsb, dw.device)
    call("mopa_spconv_bwd_weight", ptr(nbr), K, A_out, x.p, x.ld, x.C, dout.p, dout.ld, dout.C, ptr(dw),
         int(accumulate), ptr(ws), ws.numel(), stream())


def bnrelu_fwd(x: View, y: View, gamma, beta, rmean, rvar, training: bool, stats: torch.Tensor):
    wsb = query("mopa_bnrelu_rows_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    call("mopa_bnrelu_rows_fwd", x.p, x.ld, y.p, y.ld, x.rows, x.C, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
         BN_MOMENTUM, BN_EPS, LEAK, int(training), ptr(stats), ptr(ws), ws.numel(), stream())


def bnrelu_bwd(dy: View, x: View, dx: View, stats, training: bool, dgamma, dbeta, acc_dx: bool, acc_params: bool = False):
    wsb = query("mopa_bnrelu_rows_bwd_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    call("mopa_bnrelu_rows_bwd", dy.p, dy.ld, x.p, x.ld, dx.p, dx.ld, x.rows, x.C, ptr(stats), LEAK, int(training),
         ptr(dgamma), ptr(dbeta), int(acc_params), int(acc_dx), ptr(ws), ws.numel(), stream())


# --------------------------------------------------------------------------------------- the network
def unet_param_names(num_planes=7, block_reps=1, prefix="sparseModel."):
    """Parameter/buffer names in scn.Sequential index naming, traversal order."""
    convs, bns = [prefix + "1"], []

    def U(pre, depth):
        idx = 0
        for _ in range(block_reps):
            bns.append(f"{pre}{idx}.0"); convs.append(f"{pre}{idx}.1"); idx += 1
        if depth < num_planes - 1:
            p = f"{pre}{idx}.1."
            bns.append(p + "0"); convs.append(p + "1")
            U(p + "2.", depth + 1)
            bns.append(p + "3"); convs.append(p + "4")
            idx += 2
            for _ in range(block_reps):
                bns.append(f"{pre}{idx}.0"); convs.append(f"{pre}{idx}.1"); idx += 1

    U(prefix + "2.", 0)
    bns.append(prefix + "3")
    return convs, bns


class SCNNetFunction(torch.autograd.Function):
    """InputLayer -> stem -> UNet -> BNReLU -> OutputLayer -> linear heads as one autograd node.

    inputs : feats (N,cin), then flat params in `spec.order`
    outputs: feats (N,m), seg_logit (N,C), seg_logit2 (N,C) (zeros-size if no dual head)
    """

    @staticmethod
    def forward(ctx, spec, geom: Geometry3D, training: bool, feats, *flat):
        ctx.set_materialize_grads(False)   # an output that no loss uses arrives as None in backward, not as a zero tensor
        dev = geom.device
        P = dict(zip(spec.order, flat))
        A, m, L, reps = geom.num_active, spec.m, spec.num_planes, spec.block_reps
        planes = [(i + 1) * m for i in range(L)]
        tape = []
        pre = spec.prefix

        def bn(name, x: View, y: View | None = None):
            y = y or new_view(x.rows, x.C, dev)
            stats = torch.empty(4, x.C, dtype=torch.float32, device=dev)
            bnrelu_fwd(x, y, P[name + ".weight"], P[name + ".bias"], P[name + ".running_mean"],
                       P[name + ".running_var"], training, stats)
            tape.append(("bn", name, x, y, stats))
            return y

        def conv(name, table, x: View, out: View, kind):
            spconv_fwd(table, x, P[name + ".weight"], out, rb=geom.rulebook(table))
            tape.append(("conv", name, table, x, out, kind))
            return out

        def U(pfx, l, x: View) -> View:
            idx = 0
            has_down = l < L - 1
            join = torch.empty(A[l], 2 * planes[l], dtype=torch.float32, device=dev) if has_down else None
            for rep in range(reps):
                last = rep == reps - 1
                out = View(join, 0, planes[l]) if (has_down and last) else new_view(A[l], planes[l], dev)
                x = conv(f"{pfx}{idx}.1", geom.nbr27[l], bn(f"{pfx}{idx}.0", x), out, ("subm", l))
                idx += 1
            if has_down:
                p = f"{pfx}{idx}.1."
                d = conv(p + "1", geom.ch[l], bn(p + "0", x), new_view(A[l + 1], planes[l + 1], dev), ("down", l))
                d = U(p + "2.", l + 1, d)
                conv(p + "4", geom.up[l], bn(p + "3", d), View(join, planes[l], planes[l]), ("up", l))
                tape.append(("join", l, x, View(join, planes[l], planes[l])))
                x = View(join, 0, 2 * planes[l])
                idx += 2
                for rep in range(reps):
                    x = conv(f"{pfx}{idx}.1", geom.nbr27[l], bn(f"{pfx}{idx}.0", x), new_view(A[l], planes[l], dev),
                             ("subm", l))
                    idx += 1
            return x

        feats = feats.contiguous().float()
        cin = spec.in_channels
        if feats.shape[1] != cin or feats.shape[0] < geom.n_points:
            raise RuntimeError(f"feats must be (>= {geom.n_points}, {cin}), got {tuple(feats.shape)}")
        x0 = new_view(A[0], cin, dev)
        call("mopa_input_layer_fwd", ptr(feats), cin, ptr(geom.row_start), ptr(geom.row_points), A[0], x0.p, x0.ld,
             stream())
        x = conv(pre + "1", geom.nbr27[0], x0, new_view(A[0], m, dev), ("subm", 0))
        x = U(pre + "2.", 0, x)
        y = bn(pre + "3", x)
        N, C = geom.n_points, spec.num_classes
        out_feats = torch.empty(N, m, dtype=torch.float32, device=dev)
        l1 = torch.empty(N, C, dtype=torch.float32, device=dev)
        l2 = torch.empty(N, C if spec.dual_head else 0, dtype=torch.float32, device=dev)
        w2 = P["linear2.weight"] if spec.dual_head else None
        b2 = P["linear2.bias"] if spec.dual_head else None
        call("mopa_output_layer_heads_fwd", y.p, y.ld, ptr(geom.point_row), N, m, C, ptr(P["linear.weight"]),
             ptr(P["linear.bias"]), ptr(w2), ptr(b2), ptr(out_feats), ptr(l1), ptr(l2) if spec.dual_head else None,
             stream())
        U = None   # the recursive closure refers to itself through its own cell: a cycle that pins `tape` until the cyclic GC runs
        ctx.spec, ctx.geom, ctx.training, ctx.tape = spec, geom, training, tape
        # a detached alias: the returned tensor itself gets this node as grad_fn, and keeping it on ctx would be a reference
        # cycle (node -> ctx -> output -> node) that only the cyclic GC frees -- ~2 GB of activations per step
        ctx.P, ctx.y_final, ctx.out_feats = P, y, out_feats.detach()
        ctx.feats_needs_grad = feats.requires_grad
        ctx.x0 = x0
        return out_feats, l1, l2

    @staticmethod
    def backward(ctx, dfeats, dl1, dl2):
        if dfeats is None and dl1 is None and dl2 is None:   # nothing flows back (e.g. only used as a detached KL target)
            return (None,) * (4 + len(ctx.spec.order))
        spec, geom, P, tape = ctx.spec, ctx.geom, ctx.P, ctx.tape
        dev = geom.device
        N, m, C = geom.n_points, spec.m, spec.num_classes
        A0 = geom.num_active[0]
        sink = GradSink(P, spec.order)   # gradients go straight into attached .grad buffers (accumulating)

        def cont(t):
            return None if t is None else t.contiguous().float()

        dfeats, dl1 = cont(dfeats), cont(dl1)
        dl2 = cont(dl2) if (spec.dual_head and dl2 is not None and dl2.numel()) else None
        dy = new_view(A0, m, dev)
        wsb = query("mopa_output_layer_heads_bwd_workspace_bytes", N, m, C)
        ws = _ws(wsb, dev)
        hnames = (["linear.weight", "linear.bias"] if dl1 is not None else []) + \
                 (["linear2.weight", "linear2.bias"] if dl2 is not None else [])
        hg, hacc = sink.take(*hnames)
        hg = dict(zip(hnames, hg))
        call("mopa_output_layer_heads_bwd", ptr(dfeats), ptr(dl1), ptr(dl2), ptr(ctx.out_feats),
             ptr(P["linear.weight"]), ptr(P["linear2.weight"]) if spec.dual_head else None, ptr(geom.row_start),
             ptr(geom.row_points), A0, N, m, C, dy.p, dy.ld, ptr(hg.get("linear.weight")), ptr(hg.get("linear.bias")),
             ptr(hg.get("linear2.weight")), ptr(hg.get("linear2.bias")), int(hacc), ptr(ws), ws.numel(), stream())

        # gradient w.r.t. activation buffers, keyed by (storage ptr, col, C)
        gmap = {}

        def key(v: View):
            return (v.t.data_ptr(), v.col, v.C)

        gmap[key(ctx.y_final)] = dy
        for rec in reversed(tape):
            kind = rec[0]
            if kind == "bn":
                _, name, x, y, stats = rec
                dyv = gmap.pop(key(y))
                k = key(x)
                if k in gmap:  # skip half of a join buffer: accumulate into the existing gradient
                    dx, acc = gmap[k], True
                else:
                    dx, acc = new_view(x.rows, x.C, dev), False
                    gmap[k] = dx
                (dg, db), pacc = sink.take(name + ".weight", name + ".bias")
                bnrelu_bwd(dyv, x, dx, stats, ctx.training, dg, db, acc, pacc)
            elif kind == "conv":
                _, name, table, x, out, ckind = rec
                dout = gmap.pop(key(out))
                w = P[name + ".weight"]
                (dw,), wacc = sink.take(name + ".weight")
                spconv_bwd_weight(table, x, dout, dw, accumulate=wacc)
                if name == spec.prefix + "1" and not ctx.feats_needs_grad:
                    continue
                dx = new_view(x.rows, x.C, dev)
                if ckind[0] == "subm":      # nbr[o][i]=j <=> nbr[26-o][j]=i : same table, flipped offsets
                    spconv_fwd(table, dout, w, dx, w_flip=True, rb=geom.rulebook(table), w_transposed=True)
                elif ckind[0] == "down":    # rules reversed = the up table of the same level
                    rt = geom.up[ckind[1]]
                    spconv_fwd(rt, dout, w, dx, rb=geom.rulebook(rt), w_transposed=True)
                else:                        # deconv: reversed rules = the children table
                    rt = geom.ch[ckind[1]]
                    spconv_fwd(rt, dout, w, dx, rb=geom.rulebook(rt), w_transposed=True)
                gmap[key(x)] = dx
            elif kind == "join":
                # dec-block BN produced d(join) for all 2P columns; expose its halves under the keys of the
                # two producers (skip conv output / deconv output), which are column views of the same buffer.
                _, l, skip, upv = rec
                full = gmap.pop((skip.t.data_ptr(), 0, 2 * skip.C))
                gmap[key(skip)] = View(full.t, 0, skip.C)
                gmap[key(upv)] = View(full.t, skip.C, skip.C)
        dfeat_in = None
        if ctx.feats_needs_grad:
            cin = spec.in_channels
            dx0 = gmap[key(ctx.x0)]
            dfeat_in = torch.zeros(N, cin, dtype=torch.float32, device=dev)
            call("mopa_input_layer_bwd", dx0.p, dx0.ld, ptr(geom.point_row), ptr(geom.row_start), N, cin,
                 ptr(dfeat_in), stream())
        return (None, None, None, dfeat_in) + sink.returned()
