"""Losses of the hot path as autograd nodes over libmopa_hip.so.

``mask_cons_loss`` keeps the reference signature (``mopa/common/utils/loss.py:241-283``);
``xm_kl`` / ``seg_ce`` / ``softmax_lastdim`` package the inline ``torch.nn.functional`` expressions of
``mopa/train/train_xmuda_mopa.py:354-363,389-398,472-473`` as single fused kernels.
Loss values, normalisers and the upstream gradient stay on the device (no host sync).
"""
from __future__ import annotations

from typing import List

import torch

from ..._lib import call, ptr, query, stream, workspace

import os as _os

VALIDATE_LABELS = _os.environ.get("MOPA_VALIDATE_LABELS", "0") == "1"   # host-syncing range checks (see seg_ce)


def _ws(n, dev):
    return workspace.get(max(int(n), 256), dev)


def _f32c(t):
    return t.contiguous().float()


class _SoftmaxKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit_p, logit_q):
        a, b = _f32c(logit_p), _f32c(logit_q.detach())
        N, C = a.shape
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        ws = _ws(query("mopa_loss_workspace_bytes", N), a.device)
        call("mopa_softmax_kl_fwd", ptr(a), ptr(b), N, C, ptr(loss), ptr(ws), ws.numel(), stream())
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a)
        call("mopa_softmax_kl_bwd", ptr(a), ptr(b), a.shape[0], a.shape[1], ptr(_f32c(g)), ptr(da), stream())
        return da, None


def xm_kl(logit_p: torch.Tensor, logit_q: torch.Tensor) -> torch.Tensor:
    """F.kl_div(log_softmax(p), softmax(q.detach()), 'none').sum(1).mean()  (train_xmuda_mopa.py:389-398)."""
    # the target is detached BEFORE it enters the graph: as an input of the autograd node it would still schedule the other
    # network's backward (with zero gradients) -- a whole redundant backward pass per loss
    return _SoftmaxKL.apply(logit_p, logit_q.detach())


class _WeightedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weight, ignore_index):
        z = _f32c(logits)
        y = labels.to(device=z.device, dtype=torch.int64).contiguous()
        w = None if weight is None else _f32c(weight.to(z.device))
        N, C = z.shape
        out = torch.empty(2, dtype=torch.float32, device=z.device)  # loss, normaliser
        status = torch.zeros(1, dtype=torch.int32, device=z.device)
        ws = _ws(query("mopa_loss_workspace_bytes", N), z.device)
        call("mopa_wce_fwd", ptr(z), ptr(y), ptr(w), N, C, ignore_index, ptr(out), ptr(out, 1), ptr(status), ptr(ws),
             ws.numel(), stream())
        ctx.save_for_backward(z, y, out)
        ctx.w, ctx.ignore_index, ctx.status = w, ignore_index, status
        if VALIDATE_LABELS and int(status.item()) != 0:   # a host sync: debugging / validation runs only
            raise IndexError(f"seg_ce: a label is outside [0, {C}) and is not ignore_index {ignore_index} "
                             "(F.cross_entropy raises 'Target out of bounds' here)")
        return out[0]

    @staticmethod
    def backward(ctx, g):
        z, y, out = ctx.saved_tensors
        dz = torch.empty_like(z)
        call("mopa_wce_bwd", ptr(z), ptr(y), ptr(ctx.w), z.shape[0], z.shape[1], ctx.ignore_index, ptr(out, 1),
             ptr(_f32c(g)), ptr(dz), stream())
        return dz, None, None, None


def seg_ce(logits, labels, weight=None, ignore_index: int = -100) -> torch.Tensor:
    """F.cross_entropy(logits, labels, weight=weight) with torch's default ignore_index (train_xmuda_mopa.py:354-363).

    Rows with label == ignore_index are skipped in-kernel, so the boolean-mask compaction the reference does
    for pseudo labels (:452-465, a device sync) is unnecessary: pass the full tensors.

    Deviation from ``F.cross_entropy``: a label outside ``[0, C)`` that is not ``ignore_index`` (e.g. 255, or a class-count
    mismatch) is dropped from numerator and normaliser instead of raising -- raising needs a device sync per call.  The kernel
    records the condition; set ``mopa_amd.common.utils.loss.VALIDATE_LABELS = True`` (or ``MOPA_VALIDATE_LABELS=1``) to check
    it after every call, e.g. for the first iterations of a new dataset.  The same switch validates SAM mask ids (> 255) in
    ``mask_cons_loss``.
    """
    if logits.shape[0] == 0:
        return logits.sum() * float("nan")
    return _WeightedCE.apply(logits, labels, weight, ignore_index)


class _Softmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        zc = _f32c(z)
        C = zc.shape[-1]
        p = torch.empty_like(zc)
        call("mopa_softmax_fwd", ptr(zc), zc.numel() // C, C, ptr(p), stream())
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        p, = ctx.saved_tensors
        C = p.shape[-1]
        dz = torch.empty_like(p)
        call("mopa_softmax_bwd", ptr(p), ptr(_f32c(dp)), p.numel() // C, C, ptr(dz), stream())
        return dz


def softmax_lastdim(z: torch.Tensor) -> torch.Tensor:
    """F.softmax(z, dim=-1) (caller of mask_cons_loss: train_xmuda_mopa.py:473)."""
    return _Softmax.apply(z)


class _MaskCons(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, masks, min_entropy):
        p = _f32c(probs)
        B, C = p.shape[0], p.shape[-1]
        HW = p.numel() // (B * C)
        k_norm = p.shape[1]  # quirk: the caller passes (B,H,W,C), so the normaliser is log2(H)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        state = torch.empty(query("mopa_mask_cons_state_floats", B, C), dtype=torch.float32, device=p.device)
        ws = _ws(query("mopa_mask_cons_workspace_bytes", B, HW, C), p.device)
        call("mopa_mask_cons_fwd", ptr(p), ptr(masks), B, HW, C, k_norm, int(min_entropy), ptr(loss), ptr(state),
             ptr(ws), ws.numel(), stream())
        ctx.save_for_backward(p, masks, state)
        ctx.args = (B, HW, C, k_norm, int(min_entropy))
        return loss

    @staticmethod
    def backward(ctx, g):
        p, masks, state = ctx.saved_tensors
        B, HW, C, k_norm, me = ctx.args
        dp = torch.empty_like(p)
        call("mopa_mask_cons_bwd", ptr(p), ptr(masks), B, HW, C, k_norm, me, ptr(state), ptr(_f32c(g)), ptr(dp), stream())
        return dp, None, None


def mask_cons_loss(all_logits: torch.Tensor, sam_mask_ls: List[torch.Tensor], min_entropy: bool = False):
    """Intra-mask consistency loss, same call as ``mopa/common/utils/loss.py:241``.

    ``all_logits``: per-pixel class probabilities as the caller passes them, (B,H,W,C)
    (``train_xmuda_mopa.py:473-478``); ``sam_mask_ls``: B tensors (H,W) of integer mask ids, negative = ignore,
    valid ids in [0,255] (uint8 SAM files).  Returns the mean over images of the mean over mask ids.
    """
    if len(sam_mask_ls) == 0:
        return 0
    dev = all_logits.device
    masks = torch.stack([torch.as_tensor(m).to(device=dev, dtype=torch.int32) for m in sam_mask_ls]).contiguous()
    if VALIDATE_LABELS and int(masks.max().item()) > 255:
        raise IndexError("mask_cons_loss: mask id > 255 (ids come from uint8 SAM files; larger ids would be treated as ignore)")
    if masks.shape[0] != all_logits.shape[0] or masks.numel() * all_logits.shape[-1] != all_logits.numel():
        raise RuntimeError(f"mask_cons_loss: probs {tuple(all_logits.shape)} vs masks {tuple(masks.shape)}")
    return _MaskCons.apply(all_logits, masks, bool(min_entropy))
