"""Pseudo-label update of the MoPA phase on the device -- mirrors the reference's call sites
(``mopa/train/train_xmuda_mopa.py:221-226,264-335,587-591``) without their host round trips:

* ``refine_pseudo_labels(probs, pseudo_label)``  == ``mopa/data/utils/refine_pseudo_labels.py:5-22`` on device tensors;
* ``pseudo_labels(logit_2d, logit_3d, xm)``      == softmax -> (entropy-weighted fusion) -> max/argmax -> refine;
* ``FlatEMA(optimizer, decay)``                  == ``torch_ema.ExponentialMovingAverage`` over a ``FlatAdam`` buffer
  (``update()``, ``average_parameters()`` context manager).
"""
from __future__ import annotations

import contextlib

import torch

from ._lib import WEIGHTS_EPOCH, call, ptr, query, stream, workspace


def _f32(t):
    return t.contiguous().float()


def fuse(logit_a: torch.Tensor, logit_b: torch.Tensor | None = None):
    """-> (max prob (N,) fp32, label (N,) int64) of softmax(logit_a), or of the entropy-weighted fusion with logit_b."""
    a = _f32(logit_a)
    b = _f32(logit_b) if logit_b is not None else None
    n, c = a.shape
    assert b is None or b.shape == a.shape
    maxp = torch.empty(n, dtype=torch.float32, device=a.device)
    label = torch.empty(n, dtype=torch.int64, device=a.device)
    if n:
        call("mopa_pseudo_fuse", ptr(a), ptr(b), n, c, ptr(maxp), ptr(label), stream())
    return maxp, label


def refine_pseudo_labels(probs: torch.Tensor, pseudo_label: torch.Tensor, ignore_label: int = -100, num_classes: int | None = None):
    """Per class, labels whose probability is below min(median, 0.9) become ``ignore_label`` (device tensors in and out).
    ``num_classes`` defaults to 32 (the kernel's limit); labels outside [0, num_classes) pass through."""
    p = _f32(probs)
    lab = pseudo_label.contiguous().to(torch.int64)
    n = p.numel()
    c = 32 if num_classes is None else int(num_classes)
    out = torch.empty_like(lab)
    if n:
        ws = workspace.get(query("mopa_refine_pseudo_labels_workspace_bytes", c), p.device)
        call("mopa_refine_pseudo_labels", ptr(p), ptr(lab), n, c, int(ignore_label), ptr(out), ptr(ws), ws.numel(), stream())
    return out


def pseudo_labels(logit_2d: torch.Tensor, logit_3d: torch.Tensor, xm: bool, ignore_label: int = -100):
    """-> (ps_label_2d, ps_label_3d), both (N,) int64 on the device (train_xmuda_mopa.py:281-313)."""
    c = logit_2d.shape[1]
    if xm:
        maxp, lab = fuse(logit_2d, logit_3d)
        out = refine_pseudo_labels(maxp, lab, ignore_label, c)
        return out, out.clone()
    m2, l2 = fuse(logit_2d)
    m3, l3 = fuse(logit_3d)
    return refine_pseudo_labels(m2, l2, ignore_label, c), refine_pseudo_labels(m3, l3, ignore_label, c)


class FlatEMA:
    """Exponential moving average of a network's weights, kept as one flat buffer beside ``FlatAdam``'s."""

    def __init__(self, optimizer, decay: float, use_num_updates: bool = True):
        self.opt, self.decay = optimizer, float(decay)
        self.num_updates = 0 if use_num_updates else None
        self.shadow = optimizer.flat.clone()

    def state_dict(self):
        """Same keys as ``torch_ema.ExponentialMovingAverage.state_dict()`` (per-parameter shadow tensors), so EMA
        checkpoints written by the reference's ``CheckpointerV2(..., ema=...)`` style code load here and vice versa.  The
        shadow tensors take the optimizer's checkpoint shapes (SparseConvNet's 4-D conv-weight layout when ``FlatAdam`` was built
        with ``checkpoint_shapes``), i.e. the shapes ``model.state_dict()`` saves -- a reference-side ``copy_to`` then pairs every
        shadow tensor with a parameter of its own shape."""
        shapes = getattr(self.opt, "_ckpt_shapes", [None] * len(self.opt.params))
        shadow = [self.shadow[off:off + n].view(shp or p.shape).clone()
                  for (off, n), p, shp in zip(self.opt._slices, self.opt.params, shapes)]
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": shadow, "collected_params": None}

    def load_state_dict(self, sd):
        self.decay = float(sd["decay"])
        self.num_updates = sd["num_updates"]
        sp = sd["shadow_params"]
        if len(sp) != len(self.opt.params):
            raise ValueError("FlatEMA.load_state_dict: %d shadow tensors for %d parameters" % (len(sp), len(self.opt.params)))
        for (off, n), t in zip(self.opt._slices, sp):
            self.shadow[off:off + n].copy_(t.reshape(-1))

    def update(self):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        call("mopa_ema_update", ptr(self.shadow), ptr(self.opt.flat), self.shadow.numel(), decay, stream())

    @contextlib.contextmanager
    def average_parameters(self):
        """Run the body with the averaged weights in place (the parameters are views of the flat buffer)."""
        saved = self.opt.flat.clone()
        self.opt.flat.copy_(self.shadow)
        WEIGHTS_EPOCH[0] += 1   # the parameters are views of the flat buffer: their own version counters did not move
        try:
            yield
        finally:
            self.opt.flat.copy_(saved)
            WEIGHTS_EPOCH[0] += 1
