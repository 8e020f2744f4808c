"""Net2DSeg / Net3DSeg: the two segmentation networks behind the model factory.

Drop-in for ``mopa/models/xmuda_arch.py:22-126``: same constructor arguments,
``forward(data_batch: dict) -> dict`` with the same keys, same ``state_dict`` names
(``net_2d.*`` / ``net_3d.sparseModel.*``, ``linear``, ``linear2``).  The arithmetic
runs in libmopa_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import sparse3d
from .scn_unet import UNetSCN


class _Spec:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _require_cuda(module: nn.Module):
    dev = next(module.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("mopa_amd models run on an MI355X only: call .cuda() first (no CPU fallback)")
    return dev


class Net3DSeg(nn.Module):
    """3D branch (xmuda_arch.py:82-126): SCN UNet + linear head(s) on per-point features."""

    def __init__(self, num_classes, dual_head, backbone_3d, backbone_3d_kwargs, da_method=None, pretrained=False):
        super().__init__()
        self.backbone_3d = backbone_3d
        if backbone_3d != "SCN":
            raise NotImplementedError("3D backbone {} not supported".format(backbone_3d))
        self.net_3d = UNetSCN(**dict(backbone_3d_kwargs))
        m = self.net_3d.out_channels
        self.num_classes = num_classes
        self.linear = nn.Linear(m, num_classes)
        self.dual_head = dual_head
        if dual_head:
            self.linear2 = nn.Linear(m, num_classes)
        self.da_method = da_method
        if da_method == "MCD":
            raise NotImplementedError("da_method='MCD' (linear3) is not on the shipped hot path")
        self._order = None

    def _spec(self):
        if self._order is None:
            self._order = [k for k, _ in self.named_parameters()] + [k for k, _ in self.named_buffers()]
        n = self.net_3d
        return _Spec(order=self._order, prefix="net_3d.sparseModel.", in_channels=n.in_channels, m=n.m,
                     num_planes=n.num_planes, block_reps=n.block_reps, num_classes=self.num_classes,
                     dual_head=bool(self.dual_head))

    def forward(self, data_batch):
        dev = _require_cuda(self)
        locs, feats = data_batch["x"][0], data_batch["x"][1]
        geom = data_batch.get("geometry_3d") if isinstance(data_batch, dict) else None
        if geom is None:
            geom = self.net_3d.geometry(locs)
        feats = feats.to(dev, non_blocking=True)
        spec = self._spec()
        tensors = dict(self.named_parameters())
        tensors.update(dict(self.named_buffers()))
        flat = [tensors[k] for k in spec.order]
        f, l1, l2 = sparse3d.SCNNetFunction.apply(spec, geom, self.training, feats, *flat)
        preds = {"feats": f, "seg_logit": l1}
        if self.dual_head:
            preds["seg_logit2"] = l2
        return preds
