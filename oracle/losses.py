"""Oracle: loss functions of the hot path restated with torch-CPU ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED: fixtures G2/G3/G5
(``tests/golden``) were produced by the imported reference functions
(``oracle/gen_golden.py``).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def xm_kl(logit_p: torch.Tensor, logit_q: torch.Tensor) -> torch.Tensor:
    """Cross-modal KL, ``train_xmuda_mopa.py:389-398,440-445``.

    KL(q || p) summed over classes, mean over points; q (the other modality's
    head-1 logits) is detached by the caller.
    """
    return F.kl_div(F.log_softmax(logit_p, 1), F.softmax(logit_q.detach(), 1), reduction="none").sum(1).mean()


def seg_ce(logit: torch.Tensor, label: torch.Tensor, weight=None) -> torch.Tensor:
    """Weighted CE with ignore -100, ``train_xmuda_mopa.py:354-363,456-465``."""
    return F.cross_entropy(logit, label.long(), weight=weight, ignore_index=-100)


def mask_cons_loss(probs: torch.Tensor, sam_mask_ls, min_entropy: bool = True):
    """``mopa/common/utils/loss.py:241-283`` as a closed form per (image, mask id).

    probs is (B,H,W,C) (caller ``train_xmuda_mopa.py:472-478``), hence the
    entropy normaliser is log2(probs.shape[1]) = log2(H) (Appendix B.2).
    """
    K = probs.shape[1]
    per_img = []
    for b, masks in enumerate(sam_mask_ls):
        masks = torch.as_tensor(masks)
        ids = [int(i) for i in torch.unique(masks).tolist() if i >= 0]
        if not ids:
            per_img.append(0)
            continue
        tot = 0
        for m in ids:
            P = probs[b][masks == m]
            mu = P.mean(0, keepdim=True)
            l = ((P - mu) ** 2).mean()
            if min_entropy:
                l = l - (mu[0] * torch.log2(mu[0] + 1e-30)).sum() / math.log2(K)
            tot = tot + l
        per_img.append(tot / len(ids))
    if not per_img:
        return 0
    return sum(per_img) / len(per_img)


def seg_iou_update(mat, seg_logit, seg_label, num_classes, ignore_index=-100):
    """``SegIoU.update_dict`` (``mopa/models/metric.py:37-53``)."""
    pred = seg_logit.argmax(1)
    keep = seg_label != ignore_index
    inds = num_classes * seg_label[keep] + pred[keep]
    add = torch.bincount(inds, minlength=num_classes ** 2).reshape(num_classes, num_classes)
    return add if mat is None else mat + add
