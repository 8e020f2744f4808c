"""Oracle: Net2DSeg / UNetResNet34 (2D branch) restated with torch-CPU functional ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED: checked against
the imported reference modules by ``oracle/gen_golden.py`` (fixture G1 in
``tests/golden/g1_net2dseg_*.npz``).

Follows ``mopa/models/resnet34_unet.py:83-191`` (encoder = torchvision
ResNet34 with a stride-1 conv1 :93-94, decoder :104-110, forward :131-191) and
``mopa/models/xmuda_arch.py:49-79`` (full-image head :58-60, integer point
gather :62-65, point heads :73-77).  Parameter names are the reference's
``state_dict`` keys.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = [("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 6, 2), ("layer4", 512, 3, 2)]
BN2D_EPS = 1e-5
BN2D_MOMENTUM = 0.1


def param_shapes(num_classes=5, dual_head=True):
    """Ordered {state_dict key: shape} of Net2DSeg(UNetResNet34); buffers included."""
    out = {}

    def bn(name, c):
        out[name + ".weight"] = (c,)
        out[name + ".bias"] = (c,)
        out[name + ".running_mean"] = (c,)
        out[name + ".running_var"] = (c,)
        out[name + ".num_batches_tracked"] = ()

    p = "net_2d."
    out[p + "conv1.weight"] = (64, 3, 7, 7)
    bn(p + "bn1", 64)
    cin = 64
    for name, c, n, stride in LAYERS:
        for b in range(n):
            q = f"{p}{name}.{b}."
            out[q + "conv1.weight"] = (c, cin if b == 0 else c, 3, 3)
            bn(q + "bn1", c)
            out[q + "conv2.weight"] = (c, c, 3, 3)
            bn(q + "bn2", c)
            if b == 0 and (stride != 1 or cin != c):
                out[q + "downsample.0.weight"] = (c, cin, 1, 1)
                bn(q + "downsample.1", c)
        cin = c
    # decoder (resnet34_unet.py:104-110, dec_stage :115-129)
    for stage, (c_out_enc, c_in_enc) in {"5": (512, 256), "4": (256, 128), "3": (128, 64), "2": (64, 64)}.items():
        out[f"{p}dec_t_conv_stage{stage}.0.weight"] = (c_out_enc, c_in_enc, 2, 2)
        out[f"{p}dec_t_conv_stage{stage}.0.bias"] = (c_in_enc,)
        bn(f"{p}dec_t_conv_stage{stage}.1", c_in_enc)
    for stage, c in {"4": 256, "3": 128, "2": 64}.items():
        out[f"{p}dec_conv_stage{stage}.0.weight"] = (c, 2 * c, 3, 3)
        out[f"{p}dec_conv_stage{stage}.0.bias"] = (c,)
        bn(f"{p}dec_conv_stage{stage}.1", c)
    out[p + "dec_conv_stage1.weight"] = (64, 128, 3, 3)
    out[p + "dec_conv_stage1.bias"] = (64,)
    out["linear.weight"] = (num_classes, 64)
    out["linear.bias"] = (num_classes,)
    if dual_head:
        out["linear2.weight"] = (num_classes, 64)
        out["linear2.bias"] = (num_classes,)
    return out


def _bn(P, name, x, training, relu=True):
    y = F.batch_norm(x, P[name + ".running_mean"], P[name + ".running_var"], P[name + ".weight"],
                     P[name + ".bias"], training, BN2D_MOMENTUM, BN2D_EPS)
    return F.relu(y) if relu else y


def unet_resnet34_forward(P: dict, img: torch.Tensor, *, training=True, dropout_p=0.4,
                          dropout_masks=None, prefix="net_2d.", taps=None):
    """UNetResNet34.forward (resnet34_unet.py:131-191).

    ``dropout_masks``: optional pair of keep-masks (already scaled by 1/(1-p))
    for the two dropout sites (:154,:159), so train-mode parity can be checked.
    """
    p = prefix
    h, w = img.shape[2], img.shape[3]
    pad_h = (h + 15) // 16 * 16 - h
    pad_w = (w + 15) // 16 * 16 - w
    x = F.pad(img, [0, pad_w, 0, pad_h]) if (pad_h or pad_w) else img

    def drop(x, i):
        if dropout_masks is not None:
            return x * dropout_masks[i]
        return F.dropout(x, dropout_p, training)

    skips = []
    x = _bn(P, p + "bn1", F.conv2d(x, P[p + "conv1.weight"], None, 1, 3), training)
    skips.append(x)
    x = F.max_pool2d(x, 3, 2, 1)
    for name, c, n, stride in LAYERS:
        for b in range(n):
            q = f"{p}{name}.{b}."
            s = stride if b == 0 else 1
            idt = x
            y = _bn(P, q + "bn1", F.conv2d(x, P[q + "conv1.weight"], None, s, 1), training)
            y = _bn(P, q + "bn2", F.conv2d(y, P[q + "conv2.weight"], None, 1, 1), training, relu=False)
            if (q + "downsample.0.weight") in P:
                idt = _bn(P, q + "downsample.1", F.conv2d(x, P[q + "downsample.0.weight"], None, s, 0),
                          training, relu=False)
            x = F.relu(y + idt)
        if name == "layer3":
            x = drop(x, 0)
        if name == "layer4":
            x = drop(x, 1)
        else:
            skips.append(x)
    for stage, skip in (("5", skips[3]), ("4", skips[2]), ("3", skips[1]), ("2", skips[0])):
        t = f"{p}dec_t_conv_stage{stage}."
        x = _bn(P, t + "1", F.conv_transpose2d(x, P[t + "0.weight"], P[t + "0.bias"], 2), training)
        x = torch.cat([skip, x], 1)
        if taps is not None:
            x.retain_grad() if x.requires_grad else None
            taps["join" + stage] = x
        nxt = str(int(stage) - 1)
        if nxt == "1":
            x = F.conv2d(x, P[p + "dec_conv_stage1.weight"], P[p + "dec_conv_stage1.bias"], 1, 1)
        else:
            c = f"{p}dec_conv_stage{nxt}."
            x = _bn(P, c + "1", F.conv2d(x, P[c + "0.weight"], P[c + "0.bias"], 1, 1), training)
    if pad_h or pad_w:
        x = x[:, :, :h, :w]
    return x


def net2dseg_forward(P: dict, img, img_indices, *, dual_head=True, **kw):
    """Net2DSeg.forward (xmuda_arch.py:49-79) with output_all=True (build.py:10)."""
    x = unet_resnet34_forward(P, img, **kw)
    nhwc = x.permute(0, 2, 3, 1)
    out = {"seg_logit_all": F.linear(nhwc, P["linear.weight"], P["linear.bias"])}
    feats = []
    for i in range(x.shape[0]):
        idx = torch.as_tensor(np.asarray(img_indices[i]), dtype=torch.int64)
        feats.append(nhwc[i][idx[:, 0], idx[:, 1]])
    feats = torch.cat(feats, 0)
    out["feats"] = feats
    if dual_head:
        out["seg_logit2"] = F.linear(feats, P["linear2.weight"], P["linear2.bias"])
    out["seg_logit"] = F.linear(feats, P["linear.weight"], P["linear.bias"])
    return out
