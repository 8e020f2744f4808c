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

from . import _lib, syncbn
from ._lib import GradSink, call, ptr, query, stream, workspace

BN_EPS = 1e-4       # SCN BatchNormalization eps (Appendix A.6)
BN_MOMENTUM = 0.1   # SCN "momentum 0.9" == torch-style 0.1
LEAK = 0.0          # scn.UNet leakiness=0 / BatchNormReLU
BATCHED_REPACK = True   # one launch for all stale weight forms (the single-form path stays for never-built forms; tests flip the attribute)
RUN_PATH = os.environ.get("MOPA_SPCONV_RUN", "1") != "0"             # the offset-major convolution (csrc/sprun.hip); 0 = round-4 kernels only
RUN_MAX_ROWS = 220000   # 27-offset tables above this get no run-major rulebook


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
    worst-case buffers, (B) ONE host sync to read the 7 row counts, then exact-size rule tables (the grouped rulebooks are
    sized from a bound: no second round trip).
    """

    def __init__(self, coords: torch.Tensor, num_levels: int = 7, full_scale: int = 4096, device=None, group_points=None):
        """group_points (extension): an int N0 -- the first N0 points belong to a first group of scans (e.g. the source batch, the
        rest being the target batch, with its scan indices behind the source's) -- or up to two increasing point counts [N0, N1]
        (three groups).  Rows are numbered in first-seen order and a voxel belongs to one scan, so the groups' rows are consecutive
        ranges at every level: `self.split[l]` = the boundaries.  BatchNorm then runs per group on row ranges (statistics, running
        updates, gradients: as separate calls of the network would), everything else sees one batch."""
        if coords.dim() == 2 and coords.shape[1] == 3:
            # scn.InputLayer takes N x (dimension + 1) or N x dimension coordinates; without the last column every point belongs to
            # sample 0 -- the form of the reference's own call, mopa/models/xmuda_arch.py:171 (test_Net3DSeg)
            coords = torch.cat([coords, torch.zeros_like(coords[:, :1])], 1)
        if coords.dim() != 2 or coords.shape[1] != 4:
            raise RuntimeError(f"coords must be (N,4) [x,y,z,batch] or (N,3) [x,y,z], got {tuple(coords.shape)}")
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
        bounds = [] if group_points is None else ([int(group_points)] if np.isscalar(group_points) else [int(b) for b in group_points])
        if len(bounds) > 2 or any(not 0 < b < N for b in bounds) or any(a >= b for a, b in zip(bounds, bounds[1:])):
            raise ValueError(f"group_points={group_points}: need at most two increasing point counts inside (0, {N})")
        meta = torch.zeros((1 + len(bounds)) * L + 1, **i32)   # counts[0..L-1], status, then per boundary: rows in front of it [0..L-1]
        wsb = query("mopa_voxel_hash_workspace_bytes", N)
        ws = _ws(wsb, device)
        st = stream()
        call("mopa_voxel_hash_build", ptr(coords), N, ptr(tk[0]), ptr(tv[0]), cap, ptr(item_row[0]), ptr(keys[0]),
             ptr(meta), ptr(meta, L), ptr(ws), ws.numel(), st)
        for l in range(L - 1):
            call("mopa_coarsen_build", ptr(keys[l]), N, ptr(meta, l), ptr(tk[l + 1]), ptr(tv[l + 1]), cap,
                 ptr(item_row[l + 1]), ptr(keys[l + 1]), ptr(meta, l + 1), ptr(ws), ws.numel(), st)
        for k, b in enumerate(bounds):
            base = L + 1 + k * L
            call("mopa_group_split", ptr(item_row[0]), None, b, N, ptr(meta, base), st)
            for l in range(L - 1):   # parents of the rows in front of the boundary at level l
                call("mopa_group_split", ptr(item_row[l + 1]), ptr(meta, base + l), 0, N, ptr(meta, base + l + 1), st)
            # the groups must not share a voxel (scan indices disjoint and increasing from group to group: merge_domains_3d offsets
            # them): then every row of the points behind the boundary is a NEW row.  Checked on the device, read back with the rest.
            meta[L] += 2 * (item_row[0][b:N].min() < meta[base]).to(torch.int32)
        m = meta.cpu().tolist()  # the one host sync of the geometry build
        if m[L] & 1:
            raise RuntimeError("voxel coordinates out of range: need 0 <= x,y,z < 4096 and batch >= 0")
        if m[L] > 1:
            raise ValueError("group_points: a point behind a group boundary falls into a voxel of the group in front of it -- the groups' scan "
                             "indices (coords[:, 3]) must be disjoint and increasing from group to group (mopa_amd.step.merge_domains_3d)")
        self.num_active = m[:L]
        A = self.num_active
        # split[l]: the row boundaries between the groups at level l ([] for every level = None)
        self.split = [[m[L + 1 + k * L + l] for k in range(len(bounds))] for l in range(L)] if bounds else None
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
        # globally -- one count launch, one scan, one fill launch.  Built once per geometry, used by every layer's fwd and
        # bwd-data; a table's rulebook = its slice of the scan + the shared arrays.  The arrays are sized from a BOUND, not from
        # the scanned total (that was a second host round trip per geometry): every (tile, offset) pair wastes less than one
        # group, so groups <= rules / 16 + tiles * K <= K * (rows / 16 + tiles) -- ~0.08 * K * rows entries of 132 bytes, a few
        # times the exact size (tens of MB per geometry; the caching allocator re-uses the blocks step after step).
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
        ng = max(sum(t.shape[0] * ((t.shape[1] + 15) // 16 + (t.shape[1] + 63) // 64) for t in tables), 1)
        go = torch.empty(ng, **i32)
        gi = torch.empty(ng * 16, **i32)
        gout = torch.empty(ng * 16, **i32)
        call("mopa_rulebook_groups_fill_batched", desc_h.data_ptr(), len(tables), ntile, ptr(gs), ptr(go), ptr(gi), ptr(gout), st)
        for i, t in enumerate(tables):
            self._rb[t.data_ptr()] = (gs[tile0[i]:tile0[i + 1] + 1], go, gi, gout)
        # run-major rulebooks (csrc/sprun.hip: the rules of one filter offset as one contiguous run of slots) for the tables the
        # offset-major convolution can take -- every deconvolution table (one rule per output row: products go straight to the
        # output) and the 27-offset tables up to RUN_MAX_ROWS rows (above, the partial slab's bytes cost more than the fuller MFMA
        # groups return: mopa_spconv_run_wanted) -- all in three launches, sized from the bound K * rows (no host round trip)
        self._runs = {}
        rt = [t for t in self.up] + [t for t in self.nbr27 if t.shape[1] <= RUN_MAX_ROWS] if RUN_PATH else []
        for i in range(0, len(rt), 24):
            rows = []
            for t in rt[i:i + 24]:
                K, Ao = t.shape
                buf = torch.empty(query("mopa_rulebook_runs_bytes", K, Ao) // 4, **i32)
                self._runs[t.data_ptr()] = buf
                rows.append((t.data_ptr(), K, Ao, buf.data_ptr()))
            rdesc = np.asarray(rows, dtype=np.int64)
            call("mopa_rulebook_runs_build_batched", rdesc.ctypes.data, len(rows), st)
        self.row_start = torch.empty(A[0] + 1, **i32)
        self.row_points = torch.empty(N, **i32)
        wsb = query("mopa_points_csr_workspace_bytes", A[0])
        ws = _ws(wsb, device)
        call("mopa_points_csr", ptr(self.point_row), N, A[0], ptr(self.row_start), ptr(self.row_points), ptr(ws),
             ws.numel(), st)

    def desc(self) -> np.ndarray:
        """The geometry as the host table mopa_scn_forward / _backward read (layout: csrc/scn_exec.hip, geom_host)."""
        d = getattr(self, "_desc", None)
        if d is None:
            L = self.num_levels
            d = np.zeros(8 + 8 * (L + 1) + 8 + 24, np.int64)   # header, L + 1 level rows, tail (second group boundary per level), run rulebooks
            gs0, go, gi, gout = self._rb[self.nbr27[0].data_ptr()]
            d[0:8] = (L, self.n_points, self.point_row.data_ptr(), self.row_start.data_ptr(), self.row_points.data_ptr(),
                      go.data_ptr(), gi.data_ptr(), gout.data_ptr())
            for l in range(L):
                r = d[8 + 8 * l:16 + 8 * l]
                r[0], r[1], r[2] = self.num_active[l], self.nbr27[l].data_ptr(), self._rb[self.nbr27[l].data_ptr()][0].data_ptr()
                if l < L - 1:
                    r[3], r[4] = self.ch[l].data_ptr(), self._rb[self.ch[l].data_ptr()][0].data_ptr()
                    r[5], r[6] = self.up[l].data_ptr(), self._rb[self.up[l].data_ptr()][0].data_ptr()
                rbase = 8 + 8 * (L + 1) + 8 + 3 * l
                d[rbase] = self._run_ptr(self.nbr27[l])
                if l < L - 1:
                    d[rbase + 1], d[rbase + 2] = self._run_ptr(self.ch[l]), self._run_ptr(self.up[l])
                if self.split is not None:
                    r[7] = self.split[l][0]
                    if len(self.split[l]) > 1:
                        d[8 + 8 * (L + 1) + l] = self.split[l][1]
            self._desc = d
        return d

    def rule_table(self, kind: str, l: int) -> torch.Tensor:
        """subm: 27 offsets at level l; down / up: the stride-2 tables between levels l and l + 1; nin: the identity rule
        (the centre offset's row of the 27-offset table), for NetworkInNetwork as a one-offset convolution."""
        if kind == "subm":
            return self.nbr27[l]
        if kind == "down":
            return self.ch[l]
        if kind == "up":
            return self.up[l]
        return self.nbr27[l][13:14]

    def rulebook(self, table: torch.Tensor):
        return self._rb.get(table.data_ptr())

    def runs(self, table: torch.Tensor):
        """(run-major rulebook of the table, 1 if every output row has exactly one rule -- the deconvolution tables -- else 0),
        or None when none was built."""
        buf = self._runs.get(table.data_ptr())
        if buf is None:
            return None
        return buf, int(any(table is u for u in self.up))

    def _run_ptr(self, table):
        buf = self._runs.get(table.data_ptr())
        return 0 if buf is None else buf.data_ptr()

    def tensors(self):
        """Every device tensor this geometry owns."""
        out = [self.point_row, self.row_start, self.row_points]
        for lst in (self.row_keys, self.parent, self.nbr27, self.ch, self.up):
            out += list(lst)
        for rb in self._rb.values():
            out += list(rb)
        out += list(self._runs.values())
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
               w_transposed: bool = False, runs=None):
    """out = sum_o x[nbr[o]] @ Wc[o] with Wc = w ([K][Cin][Cout]) or, for backward-data (`w_transposed`), the
    per-offset transpose of the layer weight w ([K][Cout][Cin]).  `rb` = the table's grouped rulebook
    (Geometry3D.rulebook(nbr)) selects the prefetching kernels; without it the kernel compacts the dense table on the
    fly; `runs` = (the table's run-major rulebook, one-rule-per-row flag) (Geometry3D.runs(nbr)) lets the offset-major kernel take
    the shapes it wins on.  The weight is re-laid out per call as the chosen kernel wants it (packed / transposed): one tiny kernel."""
    K, A_out = nbr.shape
    cin, cout = x.C, out.C
    assert out.rows == A_out and w.shape == ((K, cout, cin) if w_transposed else (K, cin, cout)), (nbr.shape, cin, cout, w.shape)
    if runs is not None and A_out * 8 * x.ld * 4 < 1 << 32 and query("mopa_spconv_run_wanted", K, A_out, cin, cout, runs[1]):
        # offset-major: per-offset GEMM into a partial slab + ordered per-row sum (csrc/sprun.hip); same rule as scn_exec.hip
        wk = _weight_form(w, ("run", int(w_transposed), query("mopa_spconv_run_form", cin, cout)))
        spconv_launch_run(runs, K, x, wk, out, w_flip)
        return
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
_refreshed = {}   # stream -> WEIGHTS_EPOCH of the last batched refresh


def _refresh_stale_forms(st):
    """Re-lay out EVERY cached form whose weight changed since it was built -- all layers of all networks on this stream -- in one
    launch (mopa_spconv_pack_weights_batched) instead of one tiny kernel per layer and form as each is first used: a step of
    the 3D network re-packs ~50 forms (forward + backward-data of 25 convolutions) right after the optimizer step.  The cached
    destination buffers are re-used (same sizes; the launches that read them are ahead of this one on the same stream).  A form
    that was never built before (new network, another geometry-dependent plan) still takes the single-kernel path."""
    import numpy as np
    epoch = _lib.WEIGHTS_EPOCH[0]
    rows, hits = [], []
    for key, (tag, t, wref) in list(_weight_cache.items()):
        if key[2] != st:
            continue
        w = wref()
        if w is None:
            del _weight_cache[key]
            continue
        new_tag = (epoch, w._version, w.data_ptr())
        if tag == new_tag:
            continue
        form = key[1]
        flags = (form[1] | (form[2] << 8)) if form[0] == "pack" else (form[1] | (form[2] << 8) | 0x10000) if form[0] == "run" else 1
        rows.append((w.data_ptr(), t.data_ptr(), w.shape[0], w.shape[1], w.shape[2], flags))
        hits.append((key, new_tag, t, wref))
    for i in range(0, len(rows), 64):
        desc = np.asarray(rows[i:i + 64], dtype=np.int64)
        call("mopa_spconv_pack_weights_batched", desc.ctypes.data, len(desc), st)
    for key, new_tag, t, wref in hits:
        _weight_cache[key] = (new_tag, t, wref)


def _weight_form(w: torch.Tensor, form: tuple) -> torch.Tensor:
    """Packed / transposed form of a conv weight, re-used until the weight changes (the source and the target half of an
    iteration share the weights).  Same validity rules as dense2d.relayout_cached: one live tensor object, one weight
    version (autograd counter + _lib.WEIGHTS_EPOCH), one stream."""
    import weakref
    st = stream()
    key = (id(w), form, st)
    tag = (_lib.WEIGHTS_EPOCH[0], w._version, w.data_ptr())
    hit = _weight_cache.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is w:
        return hit[1]
    if hit is not None and hit[2]() is w and _refreshed.get(st) != tag[0] and BATCHED_REPACK:
        _refreshed[st] = tag[0]      # once per epoch and stream: everything stale goes in one launch
        _refresh_stale_forms(st)
        hit = _weight_cache.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1]
    K = w.shape[0]
    if form[0] == "pack":
        t = torch.empty(w.numel(), dtype=w.dtype, device=w.device)
        call("mopa_spconv_pack_weight", ptr(w), K, w.shape[1], w.shape[2], form[1], form[2], ptr(t), stream())
    elif form[0] == "run":
        t = torch.empty(w.numel(), dtype=w.dtype, device=w.device)
        call("mopa_spconv_run_pack_weight", ptr(w), K, w.shape[1], w.shape[2], form[1], ptr(t), stream())
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
    elif rb is not None and cin > 4 and (A_out + 63) // 64 < (1500 if K == 27 else 200):   # (<= 4 input channels: the stem kernel inside mopa_spconv_fwd)
        gs, go, gi, gout = rb
        ws = _ws(query("mopa_spconv_grouped_workspace_bytes", K, A_out, cout), wk.device)
        call("mopa_spconv_fwd_grouped", ptr(gs), ptr(go), ptr(gi), ptr(gout), K, A_out, x.p, x.ld, cin, ptr(wk), cout,
             int(w_flip), out.p, out.ld, ptr(ws), ws.numel(), stream())
    else:
        call("mopa_spconv_fwd", ptr(nbr), K, A_out, x.p, x.ld, cin, ptr(wk), cout, int(w_flip), out.p, out.ld, stream())


def spconv_launch_run(runs, K: int, x: View, wk: torch.Tensor, out: View, w_flip: bool):
    """The offset-major convolution launch (gather-GEMM + ordered reduce) on a weight in the run layout (bench.py times this)."""
    buf, one = runs
    ws = None if one else _ws(query("mopa_spconv_run_workspace_bytes", K, out.rows, out.C), wk.device)
    call("mopa_spconv_fwd_run", ptr(buf), K, out.rows, x.p, x.ld, x.C, ptr(wk), out.C, int(w_flip), out.p, out.ld, one,
         ptr(ws), 0 if ws is None else ws.numel(), stream())


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


def spconv_bwd_weight_of(geom, kind: str, l: int, x: View, dout: View, dw: torch.Tensor, accumulate: bool = False):
    """The weight gradient of the convolution (kind, level l): on the run-major rulebook where csrc/sprun.hip's dispatcher wants it
    (mopa_spconv_wgrad_run_wanted) -- the table's own, or for the stride-2 convolution the deconvolution table's with the two index
    lists swapped -- else on the dense table.  Mirrors csrc/scn_exec.hip::wgrad_plan_of (the two paths give the same bits)."""
    t = geom.rule_table(kind, l)
    if kind != "nin":
        rt, swap = (t, 0)
        r = geom.runs(rt)
        if r is None and kind == "down":
            rt, swap = geom.up[l], 1
            r = geom.runs(rt)
            if r is not None and not r[1]:
                r = None
        if r is not None:
            buf, one = r
            K, A = rt.shape
            if query("mopa_spconv_wgrad_run_wanted", K, A, x.C, dout.C, one):
                assert dw.shape == (K, x.C, dout.C)
                ws = _ws(query("mopa_spconv_wgrad_run_workspace_bytes", K, A, x.C, dout.C, one), dw.device)
                call("mopa_spconv_bwd_weight_run", ptr(buf), K, A, one, swap, x.p, x.ld, x.C, dout.p, dout.ld, dout.C, ptr(dw),
                     int(accumulate), ptr(ws), ws.numel(), stream())
                return
    spconv_bwd_weight(t, x, dout, dw, accumulate=accumulate)


def bn_row_groups(geom, level: int):
    """Row ranges BatchNorm treats as separate batches at `level`: [(0, A)] or, for a geometry built with group_points,
    [(0, s1), (s1, A)] / [(0, s1), (s1, s2), (s2, A)] (same rule as csrc/scn_exec.hip::bn_groups)."""
    A = geom.num_active[level]
    sp = list(geom.split[level]) if geom.split is not None else []
    s1 = sp[0] if sp else 0
    s2 = sp[1] if len(sp) > 1 else 0
    if s1 <= 0 or s1 >= A:
        return [(0, A)]
    if s2 <= s1 or s2 >= A:
        return [(0, s1), (s1, A)]
    return [(0, s1), (s1, s2), (s2, A)]


def rows_of(v: View, r0: int, r1: int) -> View:
    return v if (r0 == 0 and r1 == v.rows) else View(v.t[r0:r1], v.col, v.C)


def bnrelu_fwd(x: View, y: View, gamma, beta, rmean, rvar, training: bool, stats: torch.Tensor):
    """Returns None, or -- synchronised BatchNorm (mopa_amd.syncbn) in training mode -- the gathered moments for bnrelu_bwd."""
    if training and syncbn.active():
        return syncbn.fwd(x, y, gamma, beta, rmean, rvar, BN_MOMENTUM, BN_EPS, LEAK, 1, None, stats)
    wsb = query("mopa_bnrelu_rows_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    call("mopa_bnrelu_rows_fwd", x.p, x.ld, y.p, y.ld, x.rows, x.C, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
         BN_MOMENTUM, BN_EPS, LEAK, int(training), ptr(stats), ptr(ws), ws.numel(), stream())


def _splits(groups):
    s = [r1 for _, r1 in groups[:-1]] + [0, 0]
    return s[0], s[1]


def bnrelu_fwd_groups(x: View, y: View, gamma, beta, rmean, rvar, training: bool, stats: torch.Tensor, groups):
    """All row groups of one layer in one set of launches (3 kernels instead of 3 per group); stats: (len(groups), 4, C).  Bit-identical
    to bnrelu_fwd per row range (csrc/rows.hip::BnGroups) -- what the native executor issues too."""
    wsb = query("mopa_bnrelu_rows_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    s1, s2 = _splits(groups)
    call("mopa_bn_act_fwd_groups", x.p, x.ld, y.p, y.ld, x.rows, x.C, len(groups), s1, s2, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
         BN_MOMENTUM, BN_EPS, LEAK, 1, None, 0, int(training), ptr(stats), ptr(ws), ws.numel(), stream())


def bnrelu_bwd_groups(dy: View, x: View, dx: View, stats, training: bool, dgamma, dbeta, acc_dx: bool, acc_params: bool, groups):
    wsb = query("mopa_bnrelu_rows_bwd_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    s1, s2 = _splits(groups)
    call("mopa_bn_act_bwd_groups", dy.p, dy.ld, x.p, x.ld, dx.p, dx.ld, x.rows, x.C, len(groups), s1, s2, ptr(stats), LEAK, 1,
         None, 0, None, 0, 0, int(training), ptr(dgamma), ptr(dbeta), int(acc_params), int(acc_dx), ptr(ws), ws.numel(), stream())


def bnrelu_bwd(dy: View, x: View, dx: View, stats, training: bool, dgamma, dbeta, acc_dx: bool, acc_params: bool = False,
               gathered=None):
    if gathered is not None:   # the forward pass of this layer ran with global statistics
        return syncbn.bwd(dy, x, dx, stats, LEAK, 1, None, None, False, dgamma, dbeta, acc_params, acc_dx, gathered)
    wsb = query("mopa_bnrelu_rows_bwd_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    call("mopa_bnrelu_rows_bwd", dy.p, dy.ld, x.p, x.ld, dx.p, dx.ld, x.rows, x.C, ptr(stats), LEAK, int(training),
         ptr(dgamma), ptr(dbeta), int(acc_params), int(acc_dx), ptr(ws), ws.numel(), stream())


# --------------------------------------------------------------------------------------- the network
class Val:
    """A symbolic activation of the layer program: columns [col, col + C) of buffer `buf` at UNet level `level`."""

    __slots__ = ("buf", "col", "C", "level")

    def __init__(self, buf, col, C, level):
        self.buf, self.col, self.C, self.level = buf, col, C, level

    @property
    def key(self):
        return (self.buf, self.col, self.C)


class Program:
    """The layer sequence of ``UNetSCN`` as data: ``scn.Sequential(InputLayer, SubMConv, scn.UNet, BatchNormReLU, OutputLayer)``
    of ``mopa/models/scn_unet.py:25-30`` with ``scn.UNet`` unrolled (VGG blocks, or ResNet blocks for ``residual_blocks=True``)
    exactly as the reference's own ``UNetSCN_ED`` unrolls it (``scn_unet.py:38-219``).  Pinned against fixture G6
    (tests/test_scn_structure.py: names, channel counts, order, JoinTable operand order).

    ops: ("bn", name, src, dst) | ("conv", name, kind, level, src, dst) with kind subm / down / up / nin |
         ("add", a, b, dst).  JoinTable([skip, up]) costs nothing: both producers write halves of one buffer.
    Order inside a ResNet block: the shortcut's NetworkInNetwork is recorded AFTER the residual branch, so that in the
    reversed (backward) order it is the first writer of the block input's gradient and BatchNorm's backward accumulates."""

    def __init__(self, in_channels, m, num_planes, block_reps, residual_blocks, prefix):
        self.ops, self.bufs = [], []   # bufs: (level, width)
        planes = [(i + 1) * m for i in range(num_planes)]

        def val(level, C, buf=None, col=0):
            if buf is None:
                self.bufs.append((level, C))
                buf = len(self.bufs) - 1
            return Val(buf, col, C, level)

        def block(pre, idx, x, a, b, l, dst=None):
            if not residual_blocks:   # VGG style: Sequential(BNReLU(a), SubMConv(a -> b))
                y = val(l, a)
                self.ops.append(("bn", f"{pre}{idx}.0", x, y))
                out = dst or val(l, b)
                self.ops.append(("conv", f"{pre}{idx}.1", "subm", l, y, out))
                return out, idx + 1
            p = f"{pre}{idx}."       # ConcatTable(Identity | NiN(a -> b), Seq(BN, SubM(a -> b), BN, SubM(b -> b))), AddTable
            y1 = val(l, a); self.ops.append(("bn", p + "1.0", x, y1))
            c1 = val(l, b); self.ops.append(("conv", p + "1.1", "subm", l, y1, c1))
            y2 = val(l, b); self.ops.append(("bn", p + "1.2", c1, y2))
            c2 = val(l, b); self.ops.append(("conv", p + "1.3", "subm", l, y2, c2))
            sc = x
            if a != b:
                sc = val(l, b)
                self.ops.append(("conv", p + "0", "nin", l, x, sc))
            out = dst or val(l, b)
            self.ops.append(("add", sc, c2, out))
            return out, idx + 2

        def U(pre, l, x):
            P = planes[l]
            idx = 0
            has_down = l < num_planes - 1
            join = None
            if has_down:
                self.bufs.append((l, 2 * P))
                join = len(self.bufs) - 1
            for rep in range(block_reps):
                dst = Val(join, 0, P, l) if (has_down and rep == block_reps - 1) else None
                x, idx = block(pre, idx, x, P, P, l, dst)
            if has_down:
                p = f"{pre}{idx}.1."
                y = val(l, P); self.ops.append(("bn", p + "0", x, y))
                d = val(l + 1, planes[l + 1]); self.ops.append(("conv", p + "1", "down", l, y, d))
                d = U(p + "2.", l + 1, d)
                y = val(l + 1, planes[l + 1]); self.ops.append(("bn", p + "3", d, y))
                self.ops.append(("conv", p + "4", "up", l, y, Val(join, P, P, l)))
                x = Val(join, 0, 2 * P, l)     # JoinTable([skip, up])
                idx += 2
                for rep in range(block_reps):
                    x, idx = block(pre, idx, x, 2 * P if rep == 0 else P, P, l)
            return x

        self.x0 = val(0, in_channels)
        stem = val(0, m)
        self.ops.append(("conv", prefix + "1", "subm", 0, self.x0, stem))
        x = U(prefix + "2.", 0, stem)
        self.out = val(0, m)
        self.ops.append(("bn", prefix + "3", x, self.out))

    def backward_plan(self):
        """The backward schedule, resolved ONCE per program (it depends on the structure only): for every op in reverse order
        where its output gradient is read from and where its input gradient goes -- `(buffer, col, C)` slices of gradient
        buffers that mirror the activation buffers -- and whether that write accumulates (the slice already holds a
        contribution: the skip half of a join buffer, the input of a ResNet block).  AddTable hands its output gradient to
        both operands by aliasing.  The 3D step is host-paced, so none of this bookkeeping is redone per step."""
        if getattr(self, "_bwd", None) is not None:
            return self._bwd
        written = [[] for _ in self.bufs]
        alias = {}

        def ref(v):      # where the gradient of value v lives
            return alias.get(v.key, (v.buf, v.col, v.C))

        def write(v):
            if v.key in alias:
                return alias[v.key], True
            acc = any(c < v.col + v.C and v.col < c + n for c, n in written[v.buf])
            assert not acc or any(c <= v.col and v.col + v.C <= c + n for c, n in written[v.buf]), "partial overlap of gradient slices"
            written[v.buf].append((v.col, v.C))
            return (v.buf, v.col, v.C), acc

        plan = []
        out_ref, _ = write(self.out)
        for op in reversed(self.ops):
            if op[0] == "bn":
                _, name, src, dst = op
                dy = ref(dst)
                dx, acc = write(src)
                plan.append(("bn", name, src, dy, dx, acc))
            elif op[0] == "conv":
                _, name, kind, l, src, dst = op
                dout = ref(dst)
                if src is self.x0:     # the stem: its input gradient exists only when the caller wants d(feats)
                    plan.append(("conv", name, kind, l, src, dout, (src.buf, src.col, src.C)))
                    continue
                dx, acc = write(src)
                assert not acc, "a convolution's backward-data must be the first writer of its input gradient"
                plan.append(("conv", name, kind, l, src, dout, dx))
            else:
                _, a, b, dst = op
                g = ref(dst)
                alias[a.key] = g
                alias[b.key] = g
        self._bwd = (out_ref, plan)
        return self._bwd

    def native_tables(self):
        """The program as the host tables of the native executor (csrc/scn_exec.hip): built once per program.
        -> dict(prog int32 [n_ops][12], names [n_ops] (parameter prefix or None), stats_floats, plan int32 [n_steps][10],
        stem_step (index of the stem convolution's plan step), out (buf, col, C), x0 (buf, col, C), bufs [(level, width)])."""
        nt = getattr(self, "_native", None)
        if nt is not None:
            return nt
        kinds = {"subm": 0, "down": 1, "up": 2, "nin": 3}
        prog = np.zeros((len(self.ops), 12), np.int32)
        names, stat_off = [], 0
        for i, op in enumerate(self.ops):
            if op[0] == "bn":
                _, name, src, dst = op
                prog[i] = (0, 0, src.level, dst.level, src.buf, src.col, src.C, dst.buf, dst.col, dst.C, stat_off, 0)
                stat_off += 4 * src.C
                names.append(name)
            elif op[0] == "conv":
                _, name, kind, l, src, dst = op
                prog[i] = (1, kinds[kind], src.level, dst.level, src.buf, src.col, src.C, dst.buf, dst.col, dst.C, 0, 0)
                names.append(name)
            else:
                _, a, b, dst = op
                prog[i] = (2, 0, a.level, dst.level, a.buf, a.col, a.C, dst.buf, dst.col, dst.C, b.buf, b.col)
                names.append(None)
        index = {n: i for i, n in enumerate(names) if n is not None}
        out_ref, steps = self.backward_plan()
        plan = np.zeros((len(steps), 10), np.int32)
        stem_step = -1
        for j, st in enumerate(steps):
            if st[0] == "bn":
                _, name, src, dy, dx, acc = st
                plan[j] = (0, index[name], dy[0], dy[1], dy[2], dx[0], dx[1], dx[2], int(acc), 0)
            else:
                _, name, kind, l, src, dout, dx = st
                plan[j] = (1, index[name], dout[0], dout[1], dout[2], dx[0], dx[1], dx[2], 0, 0)
                if src is self.x0:
                    stem_step = j
        self._native = dict(prog=prog, names=names, stats_floats=stat_off, plan=plan, stem_step=stem_step, out=out_ref,
                            x0=(self.x0.buf, self.x0.col, self.x0.C), bufs=list(self.bufs))
        return self._native

    def layer_sequence(self):
        """The arithmetic layers in execution order, in the vocabulary of tests/golden/g6_scn_structure.json."""
        seq = [["InputLayer", self.x0.C, self.x0.C, None, 0]]
        kinds = {"subm": "SubmanifoldConvolution", "down": "Convolution", "up": "Deconvolution", "nin": "NetworkInNetwork"}
        for op in self.ops:
            if op[0] == "bn":
                seq.append(["BatchNormReLU", op[2].C, op[3].C, op[2].level, op[3].level])
            elif op[0] == "conv":
                seq.append([kinds[op[2]], op[4].C, op[5].C, op[4].level, op[5].level])
            else:
                seq.append(["AddTable", op[3].C, op[3].level])
        seq.append(["OutputLayer", self.out.C, self.out.C, 0, 0])
        return seq


_programs = {}


def program_for(spec) -> Program:
    key = (spec.in_channels, spec.m, spec.num_planes, spec.block_reps, bool(getattr(spec, "residual_blocks", False)), spec.prefix)
    if key not in _programs:
        _programs[key] = Program(*key)
    return _programs[key]


NATIVE = os.environ.get("MOPA_SCN_NATIVE", "1") != "0"   # A/B switch: the Python walk of the layer program (round 1-2)
_IO = dict(TRAINING=0, EPOCH=1, FEATS=2, CIN=3, X0BUF=4, X0COL=5, OUTBUF=6, OUTCOL=7, M=8, NCLS=9, W1=10, B1=11, W2=12, B2=13, OFEATS=14,
           L1=15, L2=16, STATS=17, MOMENTUM=18, EPS=19, LEAK=20, DFEATS_OUT=21, DL1=22, DL2=23, DFEATS_IN=24, DW1=25, DB1=26, DW2=27,
           DB2=28, HEADS_ACC=29, WSTREAM=30, WS2=31, WS2_BYTES=32, EV_READY=33, EV_DONE=34, N=35)

# The sparse weight gradients of the native backward pass on a second stream (csrc/scn_exec.hip, IO_WSTREAM): they need a layer's
# input and output gradient only and nothing in the pass waits for them; the backward-data / BatchNorm chain they ran in front of
# is a sequence of short launches that leaves most of the chip idle.  3D-only training 1423 -> 1495 scans/s (round 3), bit-identical
# results, neutral inside the joint steps (the chip is full there).  ON by default since round 5 (the product ships the faster
# step); the sparse-conv roofline figure of bench.py is bracketed in passes of the Python walk, which has no second stream, so the
# per-kernel figure is not measured beside a weight-gradient kernel.  MOPA_SCN_WGRAD_STREAM=0 switches it off.
SCN_WGRAD_STREAM = os.environ.get("MOPA_SCN_WGRAD_STREAM", "1") == "1"
_wgrad3 = {}   # (device index, consumer stream) -> (torch stream, "ready" event, "done" event)


def _wgrad3_side(dev):
    if not SCN_WGRAD_STREAM:
        return None
    from . import dense2d   # (same rule as the 2D branch: not when several ranks may share this device)
    if dense2d._shared_device_group() and os.environ.get("MOPA_SCN_WGRAD_STREAM_SHARED") != "1":
        return None
    key = (torch.device(dev).index, stream())
    ent = _wgrad3.get(key)
    if ent is None:
        st = torch.cuda.Stream(device=dev)
        evs = (torch.cuda.Event(), torch.cuda.Event())
        for e in evs:
            e.record(st)    # (creates the hipEvent_t: torch makes it at the first record)
        ent = _wgrad3[key] = (st, evs[0], evs[1])
    return ent


def _f64_bits(x: float) -> int:
    return int(np.float64(x).view(np.int64))


class NativeState:
    """What the native executor (csrc/scn_exec.hip) needs beside the program tables, per network instance: the parameter
    pointer table, the derived weight forms (buffers + the caller-owned record of what they hold) and the weight epoch.
    Lives on the module's parameter cache (dropped with it whenever the parameter objects may have changed)."""

    def __init__(self, prog: Program, P: dict, device):
        self.nt = prog.native_tables()
        n = len(self.nt["names"])
        self.params = np.zeros((n, 4), np.int64)
        self.forms = np.zeros((n, 2, 3), np.int64)
        self.forms[:, :, 1] = -1
        self._form_bufs = []
        for i, name in enumerate(self.nt["names"]):
            if name is not None and self.nt["prog"][i, 0] == 1:
                w = P[name + ".weight"]
                for f in range(2):
                    t = torch.empty(w.numel(), dtype=torch.float32, device=device)
                    self._form_bufs.append(t)
                    self.forms[i, f, 0] = t.data_ptr()
        self.epoch, self._tag = 0, None
        self.grads = np.zeros((n, 3), np.int64)
        self._grad_key = None

    def refresh(self, P: dict):
        """Parameter addresses (FlatAdam re-points .data at its flat buffer) and the weight epoch of this pass."""
        names, prog = self.nt["names"], self.nt["prog"]
        versions = []
        for i, name in enumerate(names):
            if name is None:
                continue
            if prog[i, 0] == 0:
                self.params[i] = (P[name + ".weight"].data_ptr(), P[name + ".bias"].data_ptr(),
                                  P[name + ".running_mean"].data_ptr(), P[name + ".running_var"].data_ptr())
            else:
                w = P[name + ".weight"]
                self.params[i, 0] = w.data_ptr()
                versions.append(w._version)
        tag = (_lib.WEIGHTS_EPOCH[0], tuple(versions), int(self.params[:, 0].sum()))
        if tag != self._tag:
            self._tag = tag
            self.epoch += 1

    def buffers(self, A, device, with_stats: bool, groups: int = 1):
        """One arena for the pass: [nbufs][2] (pointer, row stride) + the tensor that owns the memory (+ stats pointer; `groups`
        slots of 4 C floats per BatchNorm)."""
        sizes = [(A[level] * width * 4 + 255) // 256 * 256 for level, width in self.nt["bufs"]]
        extra = (self.nt["stats_floats"] * groups * 4 + 255) // 256 * 256 if with_stats else 0
        arena = torch.empty(sum(sizes) + extra, dtype=torch.uint8, device=device)
        base = arena.data_ptr()
        bufs = np.zeros((len(sizes), 2), np.int64)
        off = 0
        for b, (sz, (level, width)) in enumerate(zip(sizes, self.nt["bufs"])):
            bufs[b] = (base + off, width)
            off += sz
        return bufs, arena, base + off


def _native_forward(ctx, spec, geom, training, feats, flat, P, prog):
    dev = geom.device
    holder = spec.native_holder
    nat = getattr(holder, "native", None)
    if nat is None or nat.nt is not prog.native_tables():
        nat = holder.native = NativeState(prog, P, dev)
    nat.refresh(P)
    nt = nat.nt
    A, m, C, N = geom.num_active, spec.m, spec.num_classes, geom.n_points
    gd = geom.desc()
    bufs, arena, stats_ptr = nat.buffers(A, dev, True, 1 + len(geom.split[0]) if geom.split is not None else 1)
    out_feats = torch.empty(N, m, dtype=torch.float32, device=dev)
    l1 = torch.empty(N, C, dtype=torch.float32, device=dev)
    l2 = torch.empty(N, C if spec.dual_head else 0, dtype=torch.float32, device=dev)
    io = np.zeros(_IO["N"], np.int64)
    x0, out = nt["x0"], nt["out"]
    io[[_IO["TRAINING"], _IO["EPOCH"], _IO["FEATS"], _IO["CIN"], _IO["X0BUF"], _IO["X0COL"], _IO["OUTBUF"], _IO["OUTCOL"], _IO["M"],
        _IO["NCLS"]]] = (int(training), nat.epoch, feats.data_ptr(), spec.in_channels, x0[0], x0[1], out[0], out[1], m, C)
    io[_IO["W1"]], io[_IO["B1"]] = P["linear.weight"].data_ptr(), P["linear.bias"].data_ptr()
    if spec.dual_head:
        io[_IO["W2"]], io[_IO["B2"]], io[_IO["L2"]] = P["linear2.weight"].data_ptr(), P["linear2.bias"].data_ptr(), l2.data_ptr()
    io[_IO["OFEATS"]], io[_IO["L1"]], io[_IO["STATS"]] = out_feats.data_ptr(), l1.data_ptr(), stats_ptr
    io[_IO["MOMENTUM"]], io[_IO["EPS"]], io[_IO["LEAK"]] = _f64_bits(BN_MOMENTUM), _f64_bits(BN_EPS), _f64_bits(LEAK)
    key = (id(nt), C, m)
    wsb = geom._ws_bytes.get(key) if hasattr(geom, "_ws_bytes") else None
    if wsb is None:
        if not hasattr(geom, "_ws_bytes"):
            geom._ws_bytes = {}
        wsb = geom._ws_bytes[key] = int(_lib.load().mopa_scn_workspace_bytes(nt["prog"].ctypes.data, len(nt["prog"]), gd.ctypes.data, C, m))
    ws = _ws(wsb, dev)
    call("mopa_scn_forward", nt["prog"].ctypes.data, len(nt["prog"]), nat.params.ctypes.data, nat.forms.ctypes.data, gd.ctypes.data,
         bufs.ctypes.data, io.ctypes.data, ptr(ws), ws.numel(), stream())
    ctx.native = (nat, bufs, arena, io, wsb)
    return out_feats, l1, l2


def _native_backward(ctx, dfeats, dl1, dl2):
    spec, geom, P, prog = ctx.spec, ctx.geom, ctx.P, ctx.prog
    nat, bufs, arena, io_f, wsb = ctx.native
    nt = nat.nt
    dev = geom.device
    N, m, C = geom.n_points, spec.m, spec.num_classes
    A = geom.num_active
    sink = GradSink(P, spec.order, defer_hooks=True)   # (the pointer tables below take every gradient before the one native call)

    def cont(t):
        return None if t is None else t.contiguous().float()

    dfeats, dl1 = cont(dfeats), cont(dl1)
    dl2 = cont(dl2) if (spec.dual_head and dl2 is not None and dl2.numel()) else None
    gbufs, garena, _ = nat.buffers(A, dev, False)
    nat.refresh(P)
    io = io_f.copy()
    io[_IO["EPOCH"]] = nat.epoch
    io[_IO["DFEATS_OUT"]], io[_IO["DL1"]], io[_IO["DL2"]] = ptr(dfeats) or 0, ptr(dl1) or 0, ptr(dl2) or 0
    hnames = (["linear.weight", "linear.bias"] if dl1 is not None else []) + (["linear2.weight", "linear2.bias"] if dl2 is not None else [])
    hg, hacc = sink.take(*hnames)
    hg = dict(zip(hnames, hg))
    for key, name in (("DW1", "linear.weight"), ("DB1", "linear.bias"), ("DW2", "linear2.weight"), ("DB2", "linear2.bias")):
        io[_IO[key]] = ptr(hg.get(name)) or 0
    io[_IO["HEADS_ACC"]] = int(hacc)
    grads = nat.grads
    prog_t = nt["prog"]
    for i, name in enumerate(nt["names"]):
        if name is None:
            continue
        if prog_t[i, 0] == 0:
            (dg, db), acc = sink.take(name + ".weight", name + ".bias")
            grads[i] = (dg.data_ptr(), db.data_ptr(), int(acc))
        else:
            (dw,), acc = sink.take(name + ".weight")
            grads[i] = (dw.data_ptr(), 0, int(acc))
    plan = nt["plan"]
    dfeat_in = None
    if ctx.feats_needs_grad:
        dfeat_in = torch.zeros(N, spec.in_channels, dtype=torch.float32, device=dev)
        io[_IO["DFEATS_IN"]] = dfeat_in.data_ptr()
    elif nt["stem_step"] >= 0:
        plan = plan.copy()
        plan[nt["stem_step"], 9] = 1      # no gradient w.r.t. the input features: the stem's backward-data is skipped
    ws = _ws(wsb, dev)
    side = _wgrad3_side(dev)
    if side is not None:
        wst, ev_ready, ev_done = side
        with torch.cuda.stream(wst):
            ws2 = _ws(wsb, dev)    # the second stream's own scratch (keyed by stream)
        io[_IO["WSTREAM"]], io[_IO["WS2"]], io[_IO["WS2_BYTES"]] = wst.cuda_stream, ws2.data_ptr(), ws2.numel()
        io[_IO["EV_READY"]], io[_IO["EV_DONE"]] = ev_ready.cuda_event, ev_done.cuda_event
    call("mopa_scn_backward", prog_t.ctypes.data, len(prog_t), plan.ctypes.data, len(plan), nat.params.ctypes.data, nat.forms.ctypes.data,
         grads.ctypes.data, geom.desc().ctypes.data, bufs.ctypes.data, gbufs.ctypes.data, io.ctypes.data, ptr(ws), ws.numel(), stream())
    return (None, None, None, dfeat_in) + sink.returned()


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
        A, m = geom.num_active, spec.m
        prog = program_for(spec)
        bufs = [None] * len(prog.bufs)
        views = {}

        def view(v: Val) -> View:
            vw = views.get(v.key)
            if vw is None:
                if bufs[v.buf] is None:
                    level, width = prog.bufs[v.buf]
                    bufs[v.buf] = torch.empty(A[level], width, dtype=torch.float32, device=dev)
                vw = views[v.key] = View(bufs[v.buf], v.col, v.C)
            return vw

        def table(kind, l):
            return geom.rule_table(kind, l)

        feats = feats.contiguous().float()
        cin = spec.in_channels
        if feats.shape[1] != cin or feats.shape[0] < geom.n_points:
            raise RuntimeError(f"feats must be (>= {geom.n_points}, {cin}), got {tuple(feats.shape)}")
        # the whole pass as ONE C-ABI call (csrc/scn_exec.hip); synchronised BatchNorm has collectives between its kernels and
        # keeps the per-layer walk below
        if NATIVE and getattr(spec, "native_holder", None) is not None and not (training and syncbn.active()):
            out_feats, l1, l2 = _native_forward(ctx, spec, geom, training, feats, flat, P, prog)
            ctx.spec, ctx.geom, ctx.training, ctx.prog = spec, geom, training, prog
            ctx.P, ctx.out_feats = P, out_feats.detach()
            ctx.feats_needs_grad = feats.requires_grad
            return out_feats, l1, l2
        ctx.native = None
        x0 = view(prog.x0)
        call("mopa_input_layer_fwd", ptr(feats), cin, ptr(geom.row_start), ptr(geom.row_points), A[0], x0.p, x0.ld,
             stream())
        stats = {}
        for op in prog.ops:
            if op[0] == "bn":
                _, name, src, dst = op
                groups = bn_row_groups(geom, src.level)
                st = torch.empty(len(groups), 4, src.C, dtype=torch.float32, device=dev)
                if training and syncbn.active():   # collectives between the kernels of a layer: one call per group
                    gathered = [bnrelu_fwd(rows_of(view(src), r0, r1), rows_of(view(dst), r0, r1), P[name + ".weight"], P[name + ".bias"],
                                           P[name + ".running_mean"], P[name + ".running_var"], training, st[g])
                                for g, (r0, r1) in enumerate(groups)]
                else:
                    bnrelu_fwd_groups(view(src), view(dst), P[name + ".weight"], P[name + ".bias"], P[name + ".running_mean"],
                                      P[name + ".running_var"], training, st, groups)
                    gathered = [None]
                stats[name] = st
                if gathered[0] is not None:
                    stats[name + "/moments"] = gathered
            elif op[0] == "conv":
                _, name, kind, l, src, dst = op
                w = P[name + ".weight"]
                t = table(kind, l)
                if kind == "nin":   # NetworkInNetwork == a one-offset convolution on the identity rule (the centre offset's row)
                    w = w.view(1, w.shape[0], w.shape[1])
                spconv_fwd(t, view(src), w, view(dst), rb=geom.rulebook(t), runs=geom.runs(t))
            else:
                _, a, b, dst = op
                va, vb, vd = view(a), view(b), view(dst)
                call("mopa_rows_add", va.p, va.ld, vb.p, vb.ld, vd.p, vd.ld, vd.rows, vd.C, stream())
        y = view(prog.out)
        N, C = geom.n_points, spec.num_classes
        out_feats = torch.empty(N, m, dtype=torch.float32, device=dev)
        l1 = torch.empty(N, C, dtype=torch.float32, device=dev)
        l2 = torch.empty(N, C if spec.dual_head else 0, dtype=torch.float32, device=dev)
        w2 = P["linear2.weight"] if spec.dual_head else None
        b2 = P["linear2.bias"] if spec.dual_head else None
        call("mopa_output_layer_heads_fwd", y.p, y.ld, ptr(geom.point_row), N, m, C, ptr(P["linear.weight"]),
             ptr(P["linear.bias"]), ptr(w2), ptr(b2), ptr(out_feats), ptr(l1), ptr(l2) if spec.dual_head else None,
             stream())
        ctx.spec, ctx.geom, ctx.training = spec, geom, training
        ctx.prog, ctx.views, ctx.stats = prog, views, stats
        # a detached alias: the returned tensor itself gets this node as grad_fn, and keeping it on ctx would be a reference
        # cycle (node -> ctx -> output -> node) that only the cyclic GC frees -- ~2 GB of activations per step
        ctx.P, ctx.out_feats = P, out_feats.detach()
        ctx.feats_needs_grad = feats.requires_grad
        return out_feats, l1, l2

    @staticmethod
    def backward(ctx, dfeats, dl1, dl2):
        if dfeats is None and dl1 is None and dl2 is None:   # nothing flows back (e.g. only used as a detached KL target)
            return (None,) * (4 + len(ctx.spec.order))
        if ctx.native is not None:
            return _native_backward(ctx, dfeats, dl1, dl2)
        spec, geom, P, prog, views = ctx.spec, ctx.geom, ctx.P, ctx.prog, ctx.views
        dev = geom.device
        N, m, C = geom.n_points, spec.m, spec.num_classes
        A = geom.num_active
        A0 = A[0]
        sink = GradSink(P, spec.order)   # gradients go straight into attached .grad buffers (accumulating)

        def cont(t):
            return None if t is None else t.contiguous().float()

        out_ref, plan = prog.backward_plan()
        gbufs = [None] * len(prog.bufs)
        gviews = {}

        def gview(ref) -> View:   # (buffer, col, C) slice of the gradient buffer mirroring activation buffer `buffer`
            vw = gviews.get(ref)
            if vw is None:
                b = ref[0]
                if gbufs[b] is None:
                    level, width = prog.bufs[b]
                    gbufs[b] = torch.empty(A[level], width, dtype=torch.float32, device=dev)
                vw = gviews[ref] = View(gbufs[b], ref[1], ref[2])
            return vw

        dfeats, dl1 = cont(dfeats), cont(dl1)
        dl2 = cont(dl2) if (spec.dual_head and dl2 is not None and dl2.numel()) else None
        dy = gview(out_ref)
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

        for step in plan:
            if step[0] == "bn":
                _, name, src, dy_ref, dx_ref, acc = step
                (dg, db), pacc = sink.take(name + ".weight", name + ".bias")
                moments = ctx.stats.get(name + "/moments")
                groups = bn_row_groups(geom, src.level)
                if moments is None:
                    bnrelu_bwd_groups(gview(dy_ref), views[src.key], gview(dx_ref), ctx.stats[name], ctx.training, dg, db, acc, pacc, groups)
                else:
                    for g, (r0, r1) in enumerate(groups):   # (the groups' parameter gradients add up)
                        bnrelu_bwd(rows_of(gview(dy_ref), r0, r1), rows_of(views[src.key], r0, r1), rows_of(gview(dx_ref), r0, r1),
                                   ctx.stats[name][g], ctx.training, dg, db, acc, pacc or g > 0, gathered=moments[g])
            else:
                _, name, kind, l, src, dout_ref, dx_ref = step
                dout = gview(dout_ref)
                w = P[name + ".weight"]
                (dw,), wacc = sink.take(name + ".weight")
                t = geom.rule_table(kind, l)
                if kind == "nin":
                    w, dw = w.view(1, w.shape[0], w.shape[1]), dw.view(1, dw.shape[0], dw.shape[1])
                spconv_bwd_weight_of(geom, kind, l, views[src.key], dout, dw, accumulate=wacc)
                if src is prog.x0 and not ctx.feats_needs_grad:
                    continue
                dx = gview(dx_ref)
                if kind == "subm":        # nbr[o][i]=j <=> nbr[26-o][j]=i : same table, flipped offsets
                    spconv_fwd(t, dout, w, dx, w_flip=True, rb=geom.rulebook(t), w_transposed=True, runs=geom.runs(t))
                elif kind == "nin":
                    spconv_fwd(t, dout, w, dx, rb=geom.rulebook(t), w_transposed=True)
                else:                     # rules reversed: conv <-> deconv swap tables
                    rt = geom.up[l] if kind == "down" else geom.ch[l]
                    spconv_fwd(rt, dout, w, dx, rb=geom.rulebook(rt), w_transposed=True, runs=geom.runs(rt))
        dfeat_in = None
        if ctx.feats_needs_grad:
            cin = spec.in_channels
            dx0 = gview((prog.x0.buf, prog.x0.col, prog.x0.C))
            dfeat_in = torch.zeros(N, cin, dtype=torch.float32, device=dev)
            call("mopa_input_layer_bwd", dx0.p, dx0.ld, ptr(geom.point_row), ptr(geom.row_start), N, cin,
                 ptr(dfeat_in), stream())
        return (None, None, None, dfeat_in) + sink.returned()
