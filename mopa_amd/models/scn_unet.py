"""UNetSCN: the SparseConvNet 3D backbone, MI355X-native.

Drop-in for ``mopa/models/scn_unet.py:9-34`` (same constructor arguments and
``out_channels`` attribute).  Parameters live in nested containers whose names
reproduce scn.Sequential's index naming, e.g.
``sparseModel.2.1.1.1.weight``, so ``state_dict()`` keys line up with checkpoints.
Weights are ``(K, Cin, Cout)`` with K = 27 (submanifold) or 8 (stride-2 conv/deconv).
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from .. import sparse3d

DIMENSION = 3

# Layout of sparse-conv weights in state_dict(): "4d" = (filter_volume, 1, nIn, nOut), what current SparseConvNet (grouped
# convolutions; `install.sh:1` installs HEAD) stores and strict-loads; "3d" = (filter_volume, nIn, nOut), the older releases
# that xMUDA's published checkpoints were written with.  Loading accepts both whatever this says.
CHECKPOINT_LAYOUT = os.environ.get("MOPA_SCN_CKPT_LAYOUT", "4d")


class _Slot(nn.Module):
    """Index-named container (stands in for scn.Sequential / ConcatTable nodes)."""

    def put(self, idx, child):
        self.add_module(str(idx), child)
        return child


class _SparseConvParams(nn.Module):
    def __init__(self, K, n_in, n_out):
        super().__init__()
        # SCN init: normal(0, sqrt(2 / (nIn * filter_volume)))  (Appendix A.4/A.5)
        self.weight = nn.Parameter(torch.randn(K, n_in, n_out) * math.sqrt(2.0 / (n_in * K)))

    def checkpoint_shape(self):
        K, n_in, n_out = self.weight.shape
        return (K, 1, n_in, n_out) if CHECKPOINT_LAYOUT == "4d" else (K, n_in, n_out)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        w = destination[prefix + "weight"]
        destination[prefix + "weight"] = w.reshape(self.checkpoint_shape())   # a view: same storage, SCN's shape

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # SparseConvNet checkpoints store (filter_volume, nIn, nOut) or, in other releases, (filter_volume, 1, nIn, nOut)
        # (SURVEY.md A.4/A.7); both map onto this module's (K, Cin, Cout) without touching the values.
        key = prefix + "weight"
        w = state_dict.get(key)
        if w is not None and w.dim() == 4 and w.shape[1] == 1 and tuple(w.shape[0:1] + w.shape[2:]) == tuple(self.weight.shape):
            state_dict[key] = w.reshape(self.weight.shape)
            super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
            state_dict[key] = w   # leave the caller's dict as it was
            return
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class _NiNParams(nn.Module):
    """scn.NetworkInNetwork(nIn, nOut, bias=False): weight (nIn, nOut), init normal(0, sqrt(2 / nIn))."""

    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n_in, n_out) * math.sqrt(2.0 / n_in))


class _BNParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))


class UNetSCN(nn.Module):
    def __init__(self, in_channels, m=16, block_reps=1, residual_blocks=False, full_scale=4096, num_planes=7,
                 pretrained=False):
        super().__init__()
        # The reference accepts any m (scn_unet.py:11,23) and ships 16.  Constructing is always possible for m % 4 == 0 (rows are moved
        # as float4; state_dict names / shapes: tests/test_scn_structure.py); RUNNING needs widths the kernels hold -- the sparse
        # weight-gradient kernel at most 112 output channels (the widest level: m * num_planes), the widest convolution input (the
        # join [skip | up] in front of a decoder block: 2 * m * (num_planes - 1) channels) at most 224 channels in the MFMA-tiled
        # kernels (every channel count a multiple of 16), 192 otherwise: m = 4, 8, 12, 16 with the shipped 7 levels.
        if m <= 0 or m % 4:
            raise NotImplementedError(f"UNetSCN(m={m}): m must be a positive multiple of 4 (the reference ships m = 16)")
        self.not_runnable = self.why_not_runnable(in_channels, m, num_planes)
        self.in_channels, self.out_channels = in_channels, m
        self.m, self.block_reps, self.full_scale, self.num_planes = m, block_reps, full_scale, num_planes
        self.residual_blocks = bool(residual_blocks)
        planes = [(i + 1) * m for i in range(num_planes)]
        sm = self.sparseModel = _Slot()
        sm.put(1, _SparseConvParams(27, in_channels, m))

        def block(node, idx, a, b):
            """scn.UNet's block(m, a, b): VGG = Sequential(BN, SubM); ResNet = ConcatTable(Identity | NiN,
            Sequential(BN, SubM, BN, SubM)) + AddTable -- returns the next free index."""
            if not residual_blocks:
                blk = node.put(idx, _Slot())
                blk.put(0, _BNParams(a))
                blk.put(1, _SparseConvParams(27, a, b))
                return idx + 1
            ct = node.put(idx, _Slot())
            if a != b:
                ct.put(0, _NiNParams(a, b))
            seq = ct.put(1, _Slot())
            seq.put(0, _BNParams(a)); seq.put(1, _SparseConvParams(27, a, b))
            seq.put(2, _BNParams(b)); seq.put(3, _SparseConvParams(27, b, b))
            return idx + 2   # ConcatTable, AddTable

        def U(node, pl):
            idx = 0
            for _ in range(block_reps):
                idx = block(node, idx, pl[0], pl[0])
            if len(pl) > 1:
                seq = node.put(idx, _Slot()).put(1, _Slot())
                seq.put(0, _BNParams(pl[0]))
                seq.put(1, _SparseConvParams(8, pl[0], pl[1]))
                U(seq.put(2, _Slot()), pl[1:])
                seq.put(3, _BNParams(pl[1]))
                seq.put(4, _SparseConvParams(8, pl[1], pl[0]))
                idx += 2
                for i in range(block_reps):
                    idx = block(node, idx, pl[0] * (2 if i == 0 else 1), pl[0])

        U(sm.put(2, _Slot()), planes)
        sm.put(3, _BNParams(m))

    @staticmethod
    def why_not_runnable(in_channels, m, num_planes):
        """None, or the reason the HIP kernels cannot run this width (checked at the first forward, not at construction)."""
        widest_in = max(2 * m * (num_planes - 1), m * num_planes, in_channels)
        limit = 224 if m % 16 == 0 else 192
        if m > 64 or m * num_planes > 112 or widest_in > limit:
            return (f"UNetSCN(m={m}, num_planes={num_planes}): the sparse-conv kernels need m <= 64, m * num_planes <= 112 and "
                    f"2 * m * (num_planes - 1) <= {limit} (m = 4, 8, 12, 16 with the shipped 7 levels; the reference ships m = 16)")
        return None

    def geometry(self, locs, group_points=None) -> sparse3d.Geometry3D:
        """group_points: see Geometry3D (two groups of scans in one batch, BatchNorm per group)."""
        if self.not_runnable:
            raise NotImplementedError(self.not_runnable)
        dev = next(self.parameters()).device
        return sparse3d.Geometry3D(locs, self.num_planes, self.full_scale, dev, group_points=group_points)


def checkpoint_shapes(model: nn.Module):
    """Per trainable parameter of `model` (in ``parameters()`` order): the shape it has in ``state_dict()`` if that differs
    from the live parameter (sparse-conv weights in SparseConvNet's 4-D layout), else ``None``.  For
    ``FlatAdam(..., checkpoint_shapes=...)``."""
    special = {}
    for mod in model.modules():
        if isinstance(mod, _SparseConvParams):
            special[id(mod.weight)] = tuple(mod.checkpoint_shape())
    return [special.get(id(p)) if special.get(id(p)) != tuple(p.shape) else None for p in model.parameters() if p.requires_grad]
