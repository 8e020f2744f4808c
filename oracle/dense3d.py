"""Second, independent oracle of the 3D branch: the SCN UNet as DENSE torch ops on a small voxel grid.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  ``oracle/scn3d.py`` restates SparseConvNet with hashes, canonical row orders
and rule tables -- the same data structures the HIP path uses, written by the same author.  This file shares none of that:
no keys, no row numbering, no rule tables.  The network is executed as ``F.conv3d`` / ``F.conv_transpose3d`` / masked
batch-norm on a ``(B, C, S, S, S)`` grid, DRIVEN BY THE REFERENCE'S OWN LAYER GRAPH: the symbolic forward trace of
``mopa/models/scn_unet.py:9-34`` (``UNetSCN``) recorded under the stand-in ``sparseconvnet`` of ``oracle/scn_recorder.py`` and
committed as ``tests/golden/g6_scn_structure.json["UNetSCN"]["trace"]`` (entries: op, parameter name, src / dst value ids,
channel counts, levels).  The per-layer semantics are SURVEY.md Appendix A.2-A.6:

* InputLayer mode 4: mean of the features of the points that fall into a voxel; the voxel becomes active.
* SubmanifoldConvolution 3^3: dense cross-correlation with padding 1, evaluated at the (unchanged) active sites;
  ``W[o]``, ``o = (dx+1)*9 + (dy+1)*3 + (dz+1)``  ==  ``W.view(3,3,3,Cin,Cout)[dx+1,dy+1,dz+1]``.
* Convolution 2^3 stride 2: dense strided cross-correlation; a coarse site is active iff one of its 8 children is;
  ``o = (x&1)*4 + (y&1)*2 + (z&1)``.
* Deconvolution 2^3 stride 2: dense transposed convolution, evaluated at the FINE active set of that level (cached skip grid).
* BatchNormReLU: statistics over the active sites only, eps 1e-4, biased variance; ReLU; inactive sites stay zero.
* JoinTable: channel concatenation ``[skip, up]``.  OutputLayer: every point reads its voxel.

This does not pin SparseConvNet's numbers (its source is absent, parity stays "unpinned"); it removes "one author, one
restatement" from the evidence: two unrelated formulations must agree, forward and backward, to fp64 round-off.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-4
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6_scn_structure.json")


def load_trace(variant: str = "UNetSCN"):
    with open(GOLDEN) as f:
        return json.load(f)[variant]["trace"]


def _subm_weight(w):      # (27, Cin, Cout) -> (Cout, Cin, 3, 3, 3)
    return w.reshape(3, 3, 3, w.shape[1], w.shape[2]).permute(4, 3, 0, 1, 2)


def _down_weight(w):      # (8, Cin, Cout) -> (Cout, Cin, 2, 2, 2)
    return w.reshape(2, 2, 2, w.shape[1], w.shape[2]).permute(4, 3, 0, 1, 2)


def _up_weight(w):        # (8, Cin, Cout) -> conv_transpose3d's (Cin, Cout, 2, 2, 2)
    return w.reshape(2, 2, 2, w.shape[1], w.shape[2]).permute(3, 4, 0, 1, 2)


def masked_bn_relu(x, mask, gamma, beta):
    """Training-mode BatchNormReLU over the active sites of a dense (B,C,S,S,S) grid; `mask` (B,1,S,S,S) in {0,1}."""
    n = mask.sum()
    mean = (x * mask).sum(dim=(0, 2, 3, 4), keepdim=True) / n
    var = (((x - mean) ** 2) * mask).sum(dim=(0, 2, 3, 4), keepdim=True) / n
    y = (x - mean) * torch.rsqrt(var + BN_EPS) * gamma.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)
    return torch.relu(y) * mask, mean.flatten(), var.flatten(), n


def unet_dense(trace, params: dict, coords: np.ndarray, feats: torch.Tensor, size: int, prefix: str = "", stats: dict | None = None):
    """Run the traced UNetSCN on a dense grid.  coords (N,4) int [x,y,z,b] with 0 <= x,y,z < size; feats (N,Cin); params by the
    trace's parameter names (+ prefix), conv weights (volume, Cin, Cout).  Returns per-point features (N, m).
    `stats` (optional dict) receives name -> (batch mean, biased batch variance, active-site count) of every BatchNorm."""
    c = torch.as_tensor(np.asarray(coords), dtype=torch.int64)
    nb = int(c[:, 3].max()) + 1
    dt = feats.dtype
    # active sets of all levels: level 0 from the points, level l+1 = "any child active"
    m0 = torch.zeros(nb, 1, size, size, size, dtype=dt)
    m0[c[:, 3], 0, c[:, 0], c[:, 1], c[:, 2]] = 1.0
    masks = [m0]
    while masks[-1].shape[-1] > 1:
        masks.append(F.max_pool3d(masks[-1], 2))
    vals = {}
    for t in trace:
        op = t["op"]
        if op == "InputLayer":
            cin = feats.shape[1]
            flat = (c[:, 3] * size + c[:, 0]) * size * size + c[:, 1] * size + c[:, 2]
            s = torch.zeros(nb * size ** 3, cin, dtype=dt).index_add(0, flat, feats[: c.shape[0]])
            cnt = torch.zeros(nb * size ** 3, dtype=dt).index_add(0, flat, torch.ones(c.shape[0], dtype=dt))
            x = s / cnt.clamp(min=1.0)[:, None]
            vals[t["dst"]] = x.view(nb, size, size, size, cin).permute(0, 4, 1, 2, 3)
        elif op == "SubmanifoldConvolution":
            w = params[prefix + t["name"] + ".weight"]
            vals[t["dst"]] = F.conv3d(vals[t["src"]], _subm_weight(w), padding=1) * masks[t["level_out"]]
        elif op == "Convolution":
            w = params[prefix + t["name"] + ".weight"]
            vals[t["dst"]] = F.conv3d(vals[t["src"]], _down_weight(w), stride=2) * masks[t["level_out"]]
        elif op == "Deconvolution":
            w = params[prefix + t["name"] + ".weight"]
            vals[t["dst"]] = F.conv_transpose3d(vals[t["src"]], _up_weight(w), stride=2) * masks[t["level_out"]]
        elif op in ("BatchNormReLU", "BatchNormLeakyReLU"):
            name = prefix + t["name"]
            y, mean, var, n = masked_bn_relu(vals[t["src"]], masks[t["level_out"]], params[name + ".weight"], params[name + ".bias"])
            vals[t["dst"]] = y
            if stats is not None:
                stats[name] = (mean.detach(), var.detach(), float(n))
        elif op == "JoinTable":
            vals[t["dst"]] = torch.cat([vals[s] for s in t["srcs"]], 1)
        elif op == "OutputLayer":
            x = vals[t["src"]]
            return x[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]]
        else:
            raise NotImplementedError(f"dense oracle: layer {op} (only the VGG-style UNetSCN of the shipped configs is traced here)")
    raise RuntimeError("trace has no OutputLayer")


def net3dseg_dense(params: dict, coords, feats, size: int, dual_head: bool = True, stats: dict | None = None):
    """Net3DSeg.forward (mopa/models/xmuda_arch.py:114-126) on the dense UNet; params under Net3DSeg's state_dict names."""
    f = unet_dense(load_trace("UNetSCN"), params, coords, feats, size, prefix="net_3d.", stats=stats)
    out = {"feats": f, "seg_logit": f @ params["linear.weight"].t() + params["linear.bias"]}
    if dual_head:
        out["seg_logit2"] = f @ params["linear2.weight"].t() + params["linear2.bias"]
    return out


def dense_case(seed=11, n=700, size=64, batch=2):
    """Points of a few 'surfaces' inside a 64^3 field (7 UNet levels: 64 -> 1), duplicates included; feats of 1 channel."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u, v = rng.integers(0, 40, n), rng.integers(0, 40, n)
    plane = rng.integers(0, 3, n)
    x = np.where(plane == 0, u + 8, np.where(plane == 1, 20 + (u // 3), u + 12))
    y = np.where(plane == 0, v + 10, np.where(plane == 1, v + 4, 30 + (v // 4)))
    z = np.where(plane == 0, 12 + (u + v) // 8, np.where(plane == 1, u + 6, v + 9))
    c = np.stack([x, y, z, rng.integers(0, batch, n)], 1).astype(np.int64)
    assert c[:, :3].min() >= 0 and c[:, :3].max() < size
    return c, torch.from_numpy(rng.random((n, 1)) + 0.5)
