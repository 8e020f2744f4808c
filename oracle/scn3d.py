"""Oracle: SparseConvNet UNet (3D branch) restated on numpy + torch-CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **Parity unpinned**: the
reference delegates this arithmetic to the un-vendored ``sparseconvnet``
package; this file restates its published semantics:

* call sites followed: ``mopa/models/scn_unet.py:9-34`` (layer stack,
  hyper-parameters), ``mopa/models/scn_unet.py:38-219`` (unrolled wiring of
  BN->conv blocks / JoinTable order), ``mopa/config/xmuda.py:217-224``
  (m=16, block_reps=1, residual_blocks=False, full_scale=4096, num_planes=7),
  ``mopa/models/xmuda_arch.py:114-126`` (Net3DSeg heads).
* A.2 InputLayer(mode=4): first-seen row order, mean of duplicate points.
* A.3 OutputLayer: every point receives its voxel row.
* A.4 SubmanifoldConvolution 3^3: offset index o=(dx+1)*9+(dy+1)*3+(dz+1).
* A.5 Convolution / Deconvolution 2^3 stride 2: o=(x&1)*4+(y&1)*2+(z&1).
* A.6 BatchNormReLU: eps 1e-4, biased var to normalise, unbiased var into the
  running estimate, running = 0.9*running + 0.1*batch.
* A.7 scn.UNet wiring and state_dict names.

Row order is canonical here (and in the HIP path): level-0 rows in first-seen
point order; level l+1 rows in first-seen order of the level-l rows' parents.
Rule tables are dense ``nbr[K, A_out]`` int32 arrays (-1 = no rule).
"""
from __future__ import annotations

import numpy as np
import torch

BN_EPS = 1e-4
BN_MOMENTUM = 0.1  # SCN "momentum 0.9" == torch momentum 0.1 (Appendix A.6)


# --------------------------------------------------------------------------- #
# integer part: keys, active sets, rule tables (bit-exact contract)
# --------------------------------------------------------------------------- #
def pack_keys(coords: np.ndarray) -> np.ndarray:
    """coords (N,4) int64 [x,y,z,b] -> uint64 key b<<36 | x<<24 | y<<12 | z."""
    c = np.asarray(coords, dtype=np.int64)
    assert c.ndim == 2 and c.shape[1] == 4
    assert (c[:, :3] >= 0).all() and (c[:, :3] < 4096).all() and (c[:, 3] >= 0).all()
    return ((c[:, 3] << 36) | (c[:, 0] << 24) | (c[:, 1] << 12) | c[:, 2]).astype(np.uint64)


def unpack_keys(keys: np.ndarray) -> np.ndarray:
    k = keys.astype(np.int64)
    return np.stack([(k >> 24) & 4095, (k >> 12) & 4095, k & 4095, k >> 36], 1)


def first_seen_unique(keys: np.ndarray):
    """Unique keys in order of first occurrence + inverse map (A.2 row numbering)."""
    uniq, first_idx, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first_idx, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    return uniq[order], rank[inv.reshape(-1)]


class _Lookup:
    """key -> row via sorted search (the CPU stand-in for the device hash)."""

    def __init__(self, row_keys: np.ndarray):
        self.order = np.argsort(row_keys, kind="stable")
        self.sorted = row_keys[self.order]

    def __call__(self, q: np.ndarray) -> np.ndarray:
        pos = np.searchsorted(self.sorted, q)
        pos_c = np.minimum(pos, self.sorted.size - 1)
        hit = self.sorted[pos_c] == q
        return np.where(hit, self.order[pos_c], -1).astype(np.int32)


class Geometry:
    """Active sets and rule tables of one batch for all UNet levels."""

    def __init__(self, coords: np.ndarray, num_levels: int = 7, full_scale: int = 4096):
        coords = np.asarray(coords, dtype=np.int64)
        if coords.ndim == 2 and coords.shape[1] == 3:   # scn.InputLayer: N x dimension coordinates = every point in sample 0 (A.2;
            coords = np.concatenate([coords, np.zeros((len(coords), 1), np.int64)], 1)   # mopa/models/xmuda_arch.py:171 calls it so)
        self.n_points = coords.shape[0]
        self.num_levels = num_levels
        keys = pack_keys(coords)
        row_keys, self.point_row = first_seen_unique(keys)
        self.point_row = self.point_row.astype(np.int32)
        self.row_keys = [row_keys]
        self.nbr27, self.parent, self.octant, self.ch, self.up = [], [], [], [], []
        for l in range(num_levels):
            rk = self.row_keys[l]
            xyzb = unpack_keys(rk)
            size = full_scale >> l
            look = _Lookup(rk)
            # --- submanifold 3x3x3 table (A.4)
            nbr = np.full((27, rk.size), -1, np.int32)
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        o = (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)
                        n = xyzb.copy()
                        n[:, 0] += dx
                        n[:, 1] += dy
                        n[:, 2] += dz
                        ok = ((n[:, :3] >= 0) & (n[:, :3] < size)).all(1)
                        q = pack_keys(np.where(ok[:, None], n, xyzb))
                        r = look(q)
                        nbr[o] = np.where(ok, r, -1)
            self.nbr27.append(nbr)
            if l == num_levels - 1:
                break
            # --- stride-2 tables (A.5)
            pc = xyzb.copy()
            pc[:, :3] >>= 1
            ckeys, parent = first_seen_unique(pack_keys(pc))
            octant = ((xyzb[:, 0] & 1) * 4 + (xyzb[:, 1] & 1) * 2 + (xyzb[:, 2] & 1)).astype(np.uint8)
            parent = parent.astype(np.int32)
            ch = np.full((8, ckeys.size), -1, np.int32)
            ch[octant, parent] = np.arange(rk.size, dtype=np.int32)
            up = np.full((8, rk.size), -1, np.int32)
            up[octant, np.arange(rk.size)] = parent
            self.row_keys.append(ckeys)
            self.parent.append(parent)
            self.octant.append(octant)
            self.ch.append(ch)
            self.up.append(up)

    @property
    def num_active(self):
        return [k.size for k in self.row_keys]

    @property
    def num_rules(self):
        return [int((n >= 0).sum()) for n in self.nbr27]


# --------------------------------------------------------------------------- #
# floating-point part (torch ops so autograd supplies the backward oracle)
# --------------------------------------------------------------------------- #
def input_layer(point_row, feats: torch.Tensor, num_rows: int) -> torch.Tensor:
    """InputLayer mode 4 (A.2): per-voxel mean of the point features."""
    idx = torch.as_tensor(np.asarray(point_row), dtype=torch.int64)
    feats = feats[: idx.numel()]  # extra feature rows are ignored (Appendix B.8)
    s = torch.zeros(num_rows, feats.shape[1], dtype=feats.dtype).index_add_(0, idx, feats)
    cnt = torch.zeros(num_rows, dtype=feats.dtype).index_add_(0, idx, torch.ones(idx.numel(), dtype=feats.dtype))
    return s / cnt[:, None]


def output_layer(point_row, x: torch.Tensor) -> torch.Tensor:
    """OutputLayer (A.3): copy the voxel row to each of its points."""
    return x[torch.as_tensor(np.asarray(point_row), dtype=torch.int64)]


def sparse_conv(x: torch.Tensor, nbr, weight: torch.Tensor, num_out: int | None = None) -> torch.Tensor:
    """out[i] = sum_o x[nbr[o, i]] @ weight[o]  over rules with nbr >= 0.

    One function covers SubmanifoldConvolution (nbr27), Convolution k2s2 (ch
    table) and Deconvolution k2s2 (up table) -- A.4/A.5.  Offsets are applied in
    increasing o (this is also the HIP kernel's accumulation order).
    """
    nbr = np.asarray(nbr)
    K, A_out = nbr.shape
    if num_out is not None:
        assert num_out == A_out
    out = torch.zeros(A_out, weight.shape[2], dtype=x.dtype)
    for o in range(K):
        sel = np.nonzero(nbr[o] >= 0)[0]
        if sel.size == 0:
            continue
        rows_out = torch.from_numpy(sel.astype(np.int64))
        rows_in = torch.from_numpy(nbr[o][sel].astype(np.int64))
        out = out.index_add(0, rows_out, x[rows_in] @ weight[o])
    return out


def bn_relu(x, weight, bias, running_mean, running_var, training: bool, leak: float = 0.0,
            update_running: bool = True):
    """BatchNormReLU over active rows (A.6).  Updates running stats in place."""
    if training:
        n = x.shape[0]
        mean = x.mean(0)
        var = ((x - mean) ** 2).mean(0)  # biased
        if update_running:
            with torch.no_grad():
                unbiased = var * (n / max(n - 1, 1))
                running_mean.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach().to(running_mean.dtype))
                running_var.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unbiased.detach().to(running_var.dtype))
    else:
        mean, var = running_mean.to(x.dtype), running_var.to(x.dtype)
    y = (x - mean) * torch.rsqrt(var + BN_EPS) * weight + bias
    return torch.where(y > 0, y, y * leak)


# --------------------------------------------------------------------------- #
# parameter naming (A.7) and the network
# --------------------------------------------------------------------------- #
def fold_state_dict(sd: dict) -> dict:
    """SparseConvNet checkpoints store conv weights as (volume, 1, nIn, nOut) (grouped-convolution releases) or
    (volume, nIn, nOut); this oracle computes on the 3-D form.  Returns a dict with the 4-D tensors reshaped (views)."""
    return {k: (v.reshape(v.shape[0], v.shape[2], v.shape[3]) if (hasattr(v, "dim") and v.dim() == 4 and v.shape[1] == 1
                                                                 and "sparseModel" in k) else v) for k, v in sd.items()}


def unet_param_shapes(in_channels=1, m=16, num_planes=7, block_reps=1, prefix="sparseModel.", residual_blocks=False):
    """Ordered {name: shape} for UNetSCN (scn_unet.py:25-30 + scn.UNet A.7; VGG or ResNet style blocks)."""
    planes = [(i + 1) * m for i in range(num_planes)]
    out = {}

    def bn(name, c):
        out[name + ".weight"] = (c,)
        out[name + ".bias"] = (c,)
        out[name + ".running_mean"] = (c,)
        out[name + ".running_var"] = (c,)

    def block(pre, idx, a, b):
        if not residual_blocks:
            bn(f"{pre}{idx}.0", a)
            out[f"{pre}{idx}.1.weight"] = (27, a, b)
            return idx + 1
        if a != b:
            out[f"{pre}{idx}.0.weight"] = (a, b)   # NetworkInNetwork shortcut
        bn(f"{pre}{idx}.1.0", a)
        out[f"{pre}{idx}.1.1.weight"] = (27, a, b)
        bn(f"{pre}{idx}.1.2", b)
        out[f"{pre}{idx}.1.3.weight"] = (27, b, b)
        return idx + 2   # ConcatTable, AddTable

    def U(pre, pl):
        idx = 0
        for _ in range(block_reps):
            idx = block(pre, idx, pl[0], pl[0])
        if len(pl) > 1:
            p = f"{pre}{idx}.1."
            bn(p + "0", pl[0])
            out[p + "1.weight"] = (8, pl[0], pl[1])
            U(p + "2.", pl[1:])
            bn(p + "3", pl[1])
            out[p + "4.weight"] = (8, pl[1], pl[0])
            idx += 2  # ConcatTable, JoinTable
            for i in range(block_reps):
                idx = block(pre, idx, pl[0] * (2 if i == 0 else 1), pl[0])

    out[prefix + "1.weight"] = (27, in_channels, m)
    U(prefix + "2.", planes)
    bn(prefix + "3", m)
    return out


def unet_forward(params: dict, geom: Geometry, feats: torch.Tensor, *, m=16, num_planes=7,
                 block_reps=1, training=True, prefix="sparseModel.", taps: dict | None = None,
                 trace: list | None = None, residual_blocks=False):
    """UNetSCN.forward (scn_unet.py:32-34): returns per-point features (N, m).

    `trace` (optional list) receives the executed layer sequence as tuples in the vocabulary of
    tests/golden/g6_scn_structure.json["layer_sequence"] (structure pin, tests/test_scn_structure.py)."""
    planes = [(i + 1) * m for i in range(num_planes)]
    P = lambda n: params[prefix + n]  # noqa: E731

    def rec(*t):
        if trace is not None:
            trace.append(list(t))

    def bn(name, x, l=0):
        rec("BatchNormReLU", x.shape[1], x.shape[1], l, l)
        return bn_relu(x, P(name + ".weight"), P(name + ".bias"), P(name + ".running_mean"),
                       P(name + ".running_var"), training)

    def block(pre, idx, x, l):
        """scn.UNet block (A.7): VGG = BN -> SubM; ResNet = (Identity | NiN)(x) + SubM(BN(SubM(BN(x))))."""
        if not residual_blocks:
            w = P(f"{pre}{idx}.1.weight")
            x = sparse_conv(bn(f"{pre}{idx}.0", x, l), geom.nbr27[l], w)
            rec("SubmanifoldConvolution", w.shape[1], w.shape[2], l, l)
            return x, idx + 1
        w1, w2 = P(f"{pre}{idx}.1.1.weight"), P(f"{pre}{idx}.1.3.weight")
        sc = x
        if w1.shape[1] != w1.shape[2]:
            wn = P(f"{pre}{idx}.0.weight")
            rec("NetworkInNetwork", wn.shape[0], wn.shape[1], l, l)
            sc = x @ wn
        y = sparse_conv(bn(f"{pre}{idx}.1.0", x, l), geom.nbr27[l], w1)
        rec("SubmanifoldConvolution", w1.shape[1], w1.shape[2], l, l)
        y = sparse_conv(bn(f"{pre}{idx}.1.2", y, l), geom.nbr27[l], w2)
        rec("SubmanifoldConvolution", w2.shape[1], w2.shape[2], l, l)
        rec("AddTable", w2.shape[2], l)
        return sc + y, idx + 2

    def U(pre, l, x):
        idx = 0
        for _ in range(block_reps):
            x, idx = block(pre, idx, x, l)
        if l < num_planes - 1:
            p = f"{pre}{idx}.1."
            y = sparse_conv(bn(p + "0", x, l), geom.ch[l], P(p + "1.weight"))
            rec("Convolution", x.shape[1], y.shape[1], l, l + 1)
            y = U(p + "2.", l + 1, y)
            y = bn(p + "3", y, l + 1)
            rec("Deconvolution", y.shape[1], P(p + "4.weight").shape[2], l + 1, l)
            y = sparse_conv(y, geom.up[l], P(p + "4.weight"))
            rec("JoinTable", [x.shape[1], y.shape[1]], ["AddTable" if residual_blocks else "SubmanifoldConvolution", "Deconvolution"], l)
            x = torch.cat([x, y], 1)  # JoinTable([skip, up])
            idx += 2
            for _ in range(block_reps):
                x, idx = block(pre, idx, x, l)
        if taps is not None:
            taps[f"level{l}"] = x
        return x

    x = input_layer(geom.point_row, feats, geom.num_active[0])
    rec("InputLayer", x.shape[1], x.shape[1], None, 0)
    rec("SubmanifoldConvolution", x.shape[1], m, 0, 0)
    x = sparse_conv(x, geom.nbr27[0], P("1.weight"))
    if taps is not None:
        taps["stem"] = x
    x = U("2.", 0, x)
    x = bn("3", x)
    rec("OutputLayer", m, m, 0, 0)
    return output_layer(geom.point_row, x)


def net3dseg_forward(params: dict, geom: Geometry, feats, *, dual_head=True, training=True, **kw):
    """Net3DSeg.forward (xmuda_arch.py:114-126); params use its state_dict names."""
    f = unet_forward(params, geom, feats, training=training, prefix="net_3d.sparseModel.", **kw)
    out = {"feats": f, "seg_logit": f @ params["linear.weight"].t() + params["linear.bias"]}
    if dual_head:
        out["seg_logit2"] = f @ params["linear2.weight"].t() + params["linear2.bias"]
    return out


# --------------------------------------------------------------------------- #
# dense equivalents used to anchor the restatement
# --------------------------------------------------------------------------- #
def to_dense(x: torch.Tensor, row_keys: np.ndarray, size: int) -> torch.Tensor:
    c = unpack_keys(row_keys)
    nb = int(c[:, 3].max()) + 1
    d = torch.zeros(nb, x.shape[1], size, size, size, dtype=x.dtype)
    d[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]] = x
    return d


def from_dense(d: torch.Tensor, row_keys: np.ndarray) -> torch.Tensor:
    c = unpack_keys(row_keys)
    return d[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]]
