"""Net2DSeg / Net3DSeg: the two segmentation networks behind the model factory.

Drop-in for ``mopa/models/xmuda_arch.py:22-126``: same constructor arguments,
``forward(data_batch: dict) -> dict`` with the same keys, same ``state_dict`` names
(``net_2d.*`` / ``net_3d.sparseModel.*``, ``linear``, ``linear2``).  The arithmetic
runs in libmopa_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

import numpy as np

from .. import dense2d, sparse3d
from .resnet34_unet import UNetResNet34
from .scn_unet import UNetSCN


class _Spec:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _teacher_scope_enter():
    """Forward passes under ``torch.no_grad()`` never trust cached weight layouts, and leave none behind.

    The caches of re-laid-out conv weights (dense2d / sparse3d) are validated by ``(WEIGHTS_EPOCH, tensor._version,
    data_ptr)``.  ``torch_ema``'s ``average_parameters()`` / ``copy_to`` / ``restore`` -- what the reference uses for its
    teacher forward, ``train_xmuda_mopa.py:221-226,264-280`` -- write through ``param.data.copy_`` which changes none of
    the three, and that forward runs under ``torch.no_grad()``.  Bumping the epoch on entry and exit of every no-grad
    forward makes the teacher pass read the live (EMA) weights and the next student pass rebuild from the restored ones;
    cost: one weight re-layout (~0.1 ms) per no-grad forward.  Any other raw in-place write (e.g. ``dist.broadcast(p.data)``
    after a first forward) needs ``mopa_amd.invalidate_weight_caches()``."""
    if not torch.is_grad_enabled():
        from .._lib import WEIGHTS_EPOCH
        WEIGHTS_EPOCH[0] += 1
        return True
    return False


def _teacher_scope_exit(entered):
    if entered:
        from .._lib import WEIGHTS_EPOCH
        WEIGHTS_EPOCH[0] += 1


class _FlatCache:
    """`[tensor for name in spec.order]` of a module, resolved once: walking named_parameters() / named_buffers() costs ~0.3 ms per
    forward (25,000 generator steps for the 2D network), which matters for the host-paced 3D step.  Parameters and buffers keep
    their identity under load_state_dict / optimizer updates; `Module._apply` (.cuda(), .to(), .float()) replaces buffer objects
    and assigning a sub-module or parameter to the network (`model.linear = nn.Linear(64, 11)` for another label set) replaces
    parameter objects: the owning module drops the cache in both cases (`_CachedParams`).  Anything that swaps tensor OBJECTS
    deeper inside (`load_state_dict(..., assign=True)`, `model.net_2d.layer4.to(...)`) is caught by the per-forward identity check
    below when it keeps the module objects; replacing a whole sub-module (`model.net_2d.layer4 = ...`) still needs
    `model.refresh_parameters()` (the cached owner dictionaries belong to the old module)."""

    def __init__(self):
        self.order, self.flat, self._owners = None, None, None

    def get(self, module):
        if self.flat is not None:
            # cheap per-forward validation: every cached tensor must still be the object its owning sub-module holds --
            # two dict lookups per entry (~20-50 us per forward); a stale entry (sub-module .to() / load_state_dict(assign=True) /
            # a replaced layer) would otherwise let the kernels update orphaned running statistics silently.  Stale -> rebuilt.
            for (params, bufs, attr), t in zip(self._owners, self.flat):
                if params.get(attr) is not t and bufs.get(attr) is not t:
                    self.flat = None
                    self.__dict__.pop("native", None)     # the native executor's pointer tables belong to the old objects
                    self.__dict__.pop("graphs2d", None)   # so do the recorded HIP graphs of the 2D backbone
                    break
        if self.flat is None:
            self.order = [k for k, _ in module.named_parameters() if not k.startswith("linear3.")] + [k for k, _ in module.named_buffers()]
            tensors = dict(module.named_parameters())
            tensors.update(dict(module.named_buffers()))
            self.flat = [tensors[k] for k in self.order]
            self._owners = []
            for k in self.order:
                prefix, _, attr = k.rpartition(".")
                owner = module.get_submodule(prefix) if prefix else module
                self._owners.append((owner._parameters, owner._buffers, attr))
        return self.order, self.flat


class _CachedParams:
    """Mixin of the two networks: owns `_cache` and drops it whenever the set of tensor objects may have changed."""

    def _apply(self, fn, *a, **k):
        self.__dict__["_cache"] = _FlatCache()
        return super()._apply(fn, *a, **k)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)) and "_cache" in self.__dict__:
            self.__dict__["_cache"] = _FlatCache()
        super().__setattr__(name, value)

    def refresh_parameters(self):
        """Re-resolve the parameter / buffer objects on the next forward (after replacing tensors inside a sub-module)."""
        self.__dict__["_cache"] = _FlatCache()


def _require_cuda(module: nn.Module):
    dev = next(module.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("mopa_amd models run on an MI355X only: call .cuda() first (no CPU fallback)")
    return dev


class Net2DSeg(_CachedParams, nn.Module):
    """2D branch (xmuda_arch.py:22-79): UNetResNet34, full-image head, integer point gather, point head(s)."""

    def __init__(self, num_classes, dual_head, backbone_2d, backbone_2d_kwargs, output_all=False):
        super().__init__()
        if backbone_2d != "UNetResNet34":
            raise NotImplementedError("2D backbone {} not supported".format(backbone_2d))
        self.net_2d = UNetResNet34(**dict(backbone_2d_kwargs))
        feat_channels = 64
        self.num_classes = num_classes
        self.linear = nn.Linear(feat_channels, num_classes)
        self.output_all = output_all
        self.dual_head = dual_head
        if dual_head:
            self.linear2 = nn.Linear(feat_channels, num_classes)
        self._cache = _FlatCache()
        self._calls = 0

    @staticmethod
    def pack_indices(img_indices, H, W, device):
        """list of B (N_b,2) [row v, col u] arrays -> int32 pixel-row ids into the /16-padded NHWC feature map."""
        Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
        rows = []
        for b, idx in enumerate(img_indices):
            idx = idx.cpu().numpy() if torch.is_tensor(idx) else np.asarray(idx)
            idx = idx.astype(np.int64).reshape(-1, 2)
            if idx.size and (idx.min() < 0 or idx[:, 0].max() >= H or idx[:, 1].max() >= W):
                raise IndexError(f"img_indices of image {b} outside the {H}x{W} image")
            rows.append((b * Hp + idx[:, 0]) * Wp + idx[:, 1])
        flat = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        t = torch.from_numpy(flat.astype(np.int32))
        if torch.device(device).type != "cuda" or t.numel() == 0:
            return t.to(device)
        from .._lib import upload
        return upload(t, device)

    def forward(self, data_batch):
        dev = _require_cuda(self)
        img = data_batch["img"].to(dev, non_blocking=True)
        if img.dim() != 4 or img.shape[1] != 3:
            raise RuntimeError(f"img must be (B,3,H,W), got {tuple(img.shape)}")
        H, W = img.shape[2], img.shape[3]
        pix = data_batch.get("point_pix_2d")
        if pix is None:
            if len(data_batch["img_indices"]) != img.shape[0]:
                raise IndexError("img_indices must hold one array per image")
            pix = self.pack_indices(data_batch["img_indices"], H, W, dev)
        order, flat = self._cache.get(self)
        # "bn_groups": G (extension, default 1): the batch is G consecutive equal groups of images that would otherwise be G calls
        # (source batch, then target batch): BatchNorm statistics / running-statistics updates / dropout masks per group in that
        # order -- same result as the G calls, one pass over G times the rows (dense2d._backbone_forward)
        groups = int(data_batch.get("bn_groups", 1))
        if groups < 1 or img.shape[0] % groups:
            raise ValueError(f"bn_groups={groups} does not divide the batch of {img.shape[0]} images")
        if groups > 3:   # the grouped BatchNorm kernels hold three row ranges per launch (csrc/rows.hip::BN_MAX_GROUPS)
            raise ValueError(f"bn_groups={groups}: at most 3 groups per pass (source, target and one more batch); call the network once per "
                             "group, or in passes of up to three groups")
        # graphs: where dense2d keeps the recorded HIP graphs of the backbone (dropped with the cache when tensor objects change)
        spec = _Spec(order=order, num_classes=self.num_classes, dual_head=bool(self.dual_head), graphs=self._cache,
                     grad_enabled=torch.is_grad_enabled(), groups=groups)
        seeds = []
        for _ in range(groups):   # one dropout seed per call that this pass stands for
            self._calls += 1
            seeds.append((torch.initial_seed() * 1000003 + self._calls) & 0x7FFFFFFFFFFF)
        seed = seeds[0] if groups == 1 else tuple(seeds)
        scope = _teacher_scope_enter()
        with torch.cuda.device(dev):   # kernels launch on the current stream of the device the tensors live on
            feats, l1, l2, pred_all = dense2d.Net2DFunction.apply(spec, img, pix, self.training, float(self.net_2d.dropout.p),
                                                                 seed, *flat)
        _teacher_scope_exit(scope)
        preds = {"feats": feats}
        if self.output_all:
            preds["seg_logit_all"] = pred_all
        if self.dual_head:
            preds["seg_logit2"] = l2
        preds["seg_logit"] = l1
        return preds


class Net3DSeg(_CachedParams, nn.Module):
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
        if da_method == "MCD":   # a third head that the reference creates but never uses in forward (xmuda_arch.py:110-126)
            self.linear3 = nn.Linear(m, num_classes)
        self._cache = _FlatCache()

    def _spec(self):
        order, _ = self._cache.get(self)
        n = self.net_3d
        return _Spec(order=order, prefix="net_3d.sparseModel.", in_channels=n.in_channels, m=n.m,
                     num_planes=n.num_planes, block_reps=n.block_reps, residual_blocks=n.residual_blocks,
                     num_classes=self.num_classes,
                     dual_head=bool(self.dual_head),
                     native_holder=self._cache)   # the native executor's per-network state lives (and dies) with the parameter cache

    def forward(self, data_batch):
        dev = _require_cuda(self)
        locs, feats = data_batch["x"][0], data_batch["x"][1]
        geom = data_batch.get("geometry_3d") if isinstance(data_batch, dict) else None
        # "bn_group_points": N0 or [N0, N1] (extension, like Net2DSeg's "bn_groups"): the first N0 points are the scans of a first batch
        # (source), the following ones those of a second (target; its scan indices behind the first's) and, with N1, a third (the
        # VGI batch) that the reference sends through the network in separate calls (train_xmuda_mopa.py:343,427,558): BatchNorm
        # statistics, running updates and gradients per group in that order, everything else on the joint batch
        # (sparse3d.Geometry3D.split).  A geometry passed in must have been built the same way.
        gp = data_batch.get("bn_group_points") if isinstance(data_batch, dict) else None
        if self.net_3d.not_runnable:
            raise NotImplementedError(self.net_3d.not_runnable)
        if geom is None:
            with torch.cuda.device(dev):
                geom = self.net_3d.geometry(locs, group_points=gp)
        elif gp is not None and geom.split is None:
            raise ValueError("bn_group_points given, but geometry_3d was built without group_points")
        feats = feats.to(dev, non_blocking=True)
        spec = self._spec()
        flat = self._cache.get(self)[1]
        scope = _teacher_scope_enter()
        with torch.cuda.device(dev):
            f, l1, l2 = sparse3d.SCNNetFunction.apply(spec, geom, self.training, feats, *flat)
        _teacher_scope_exit(scope)
        preds = {"feats": f, "seg_logit": l1}
        if self.dual_head:
            preds["seg_logit2"] = l2
        return preds
