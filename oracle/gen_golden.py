"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY.  Run from the repo root:  python oracle/gen_golden.py
Needs /root/reference (absent on the GPU box -- the committed .npz/.json files
are what travels).  The reference's Python is imported, never copied.

Missing third-party modules are stubbed in ``sys.modules``:
* ``torchvision.models.resnet.resnet34``: a plain-torch ResNet34 with
  torchvision's attribute / parameter names (public architecture: BasicBlock
  [3,4,6,3], conv1 7x7 s2, maxpool 3x3 s2).  The reference code under test is
  ``UNetResNet34`` + ``Net2DSeg`` which wrap it.
* ``sparseconvnet``: the *recording* stand-in of oracle/scn_recorder.py -- generic
  containers + leaf layers that log constructor arguments and a symbolic forward
  trace (G6: structure pin).  No arithmetic flows through it; the 3D oracle's
  numbers stay "parity unpinned".
Weights come from oracle.params.det_tensor(name) so only I/O is stored.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


# ----------------------------------------------------------------------------- stubs
def _install_stubs():
    class BasicBlock(nn.Module):
        def __init__(self, cin, c, stride):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, c, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(c)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(c)
            self.downsample = None
            if stride != 1 or cin != c:
                self.downsample = nn.Sequential(nn.Conv2d(cin, c, 1, stride, bias=False), nn.BatchNorm2d(c))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            y = self.relu(self.bn1(self.conv1(x)))
            y = self.bn2(self.conv2(y))
            return self.relu(y + idt)

    class ResNet34(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            cin = 64
            for i, (c, n, s) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
                blocks = [BasicBlock(cin, c, s)] + [BasicBlock(c, c, 1) for _ in range(n - 1)]
                setattr(self, f"layer{i}", nn.Sequential(*blocks))
                cin = c

    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")
    tvr.resnet34 = lambda pretrained=False: ResNet34()
    tv.models, tvm.resnet = tvm, tvr
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.resnet": tvr,
                        "sparseconvnet": _scn_recorder().as_module(),
                        "openpyxl": types.ModuleType("openpyxl")})  # metric_logger.py:11 (xlsx export, unused here)
    sys.path.insert(0, REF)


def _scn_recorder():
    from oracle import scn_recorder
    return scn_recorder


def _np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


# ----------------------------------------------------------------------------- G1
def gen_g1():
    from mopa.models.xmuda_arch import Net2DSeg  # reference
    from oracle.params import det_tensor

    def build(C=5):
        net = Net2DSeg(num_classes=C, dual_head=True, backbone_2d="UNetResNet34",
                       backbone_2d_kwargs={"pretrained": False}, output_all=True)
        sd = net.state_dict()
        net.load_state_dict({k: det_tensor(k, v.shape) for k, v in sd.items()})
        return net

    cases = {"pad_eval": ((2, 3, 30, 46), False), "nopad_eval": ((1, 3, 32, 48), False),
             "pad_train": ((2, 3, 30, 46), True)}
    for name, (shape, train) in cases.items():
        rng = np.random.Generator(np.random.PCG64(len(name)))
        img = torch.from_numpy(rng.random(shape, dtype=np.float32))
        idx = [np.stack([rng.integers(0, shape[2], 200), rng.integers(0, shape[3], 200)], 1).astype(np.int64)
               for _ in range(shape[0])]
        net = build()
        net.train(train)
        net.net_2d.dropout.p = 0.0  # train-mode parity without RNG (hard part 5)
        img.requires_grad_(True)
        out = net({"img": img, "img_indices": idx})
        save = {"img": img, **{f"idx{i}": a for i, a in enumerate(idx)}}
        save.update({"out_" + k: v for k, v in out.items()})
        if train:
            g = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape), dtype=np.float32)) for k, v in out.items()}
            save.update({"gin_" + k: v for k, v in g.items()})
            sum((out[k] * g[k]).sum() for k in out).backward()
            save["grad_img"] = img.grad
            norms = {}
            for k, p in net.named_parameters():
                norms[k] = [float(p.grad.double().sum()), float(p.grad.double().norm())]
            for k in ("net_2d.conv1.weight", "linear.weight", "linear2.bias", "net_2d.dec_conv_stage1.bias",
                      "net_2d.layer2.0.downsample.0.weight", "net_2d.dec_t_conv_stage2.0.weight",
                      "net_2d.layer4.2.bn2.weight", "net_2d.bn1.bias"):
                save["pgrad_" + k] = dict(net.named_parameters())[k].grad
            for k in ("net_2d.bn1.running_mean", "net_2d.bn1.running_var", "net_2d.layer3.5.bn2.running_var",
                      "net_2d.dec_conv_stage2.1.running_mean"):
                save["buf_" + k] = net.state_dict()[k]
            with open(os.path.join(OUT, f"g1_net2dseg_{name}_gradnorms.json"), "w") as f:
                json.dump(norms, f, indent=0)
        np.savez_compressed(os.path.join(OUT, f"g1_net2dseg_{name}.npz"), **_np(save))
        print("G1", name, {k: tuple(v.shape) for k, v in out.items()})


def gen_g1b():
    """G1b: a WELL-CONDITIONED train-mode case of the reference Net2DSeg -- 2 x (3, 64, 96): layer4 / decoder stage 5 live on a
    4 x 6 map, 48 samples per channel, so one BN-ReLU mask flip no longer moves bottleneck gradients by percent and the GPU
    test can bound EVERY parameter gradient at 1 %.  Inputs are regenerated from the seed; stored:
    the point outputs, a strided sample of seg_logit_all, gradient norms of all parameters, slices of the bottleneck grads."""
    from mopa.models.xmuda_arch import Net2DSeg  # reference
    from oracle.params import det_tensor

    rng = np.random.Generator(np.random.PCG64(64096))
    shape = (2, 3, 64, 96)
    img = torch.from_numpy(rng.random(shape, dtype=np.float32))
    idx = [np.stack([rng.integers(0, 64, 200), rng.integers(0, 96, 200)], 1).astype(np.int64) for _ in range(2)]
    net = Net2DSeg(num_classes=5, dual_head=True, backbone_2d="UNetResNet34", backbone_2d_kwargs={"pretrained": False}, output_all=True)
    net.load_state_dict({k: det_tensor(k, v.shape) for k, v in net.state_dict().items()})
    net.train()
    net.net_2d.dropout.p = 0.0
    out = net({"img": img, "img_indices": idx})
    g = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape), dtype=np.float32)) for k, v in out.items()}
    sum((out[k] * g[k]).sum() for k in out).backward()
    save = {"out_feats_s4": out["feats"][::4], "out_seg_logit": out["seg_logit"], "out_seg_logit2": out["seg_logit2"],
            "out_seg_logit_all_s4": out["seg_logit_all"][:, ::4, ::4]}
    named = dict(net.named_parameters())
    for k in ("net_2d.layer4.2.conv2.weight", "net_2d.layer4.0.downsample.0.weight", "net_2d.dec_t_conv_stage5.0.weight",
              "net_2d.dec_conv_stage4.0.weight", "net_2d.layer3.5.conv2.weight"):
        save["pgrad_" + k] = named[k].grad[:1].clone()          # first slice along dim 0
    for k in ("net_2d.conv1.weight", "net_2d.layer4.2.bn2.weight", "net_2d.dec_t_conv_stage5.1.bias", "linear.weight"):
        save["pgrad_" + k] = named[k].grad
    norms = {k: [float(p.grad.double().sum()), float(p.grad.double().norm())] for k, p in named.items()}
    with open(os.path.join(OUT, "g1b_net2dseg_64x96_train_gradnorms.json"), "w") as f:
        json.dump(norms, f, indent=0)
    np.savez_compressed(os.path.join(OUT, "g1b_net2dseg_64x96_train.npz"), **_np(save))
    print("G1b", {k: tuple(v.shape) for k, v in out.items()})


def gen_g1c():
    """G1c: the reference's OWN call shape, scaled down: mopa/models/xmuda_arch.py:129-162 (test_Net2DSeg) builds
    Net2DSeg(11 classes, dual head) and calls it with a (B, N / B, 2) index TENSOR on 2 x 3 x 225 x 400 -- the nuScenes resize of
    config/xmuda.py:98 (225 -> pad 240, layer4 at 15 x 25).  Here the same aspect at 45 x 80 (pad 48 x 80: layer4 at 3 x 5), 11
    classes, the index tensor form, train mode (dropout p = 0) and eval mode.  Stored: point outputs, a strided sample of
    seg_logit_all, gradient norms of all parameters (train)."""
    from mopa.models.xmuda_arch import Net2DSeg  # reference
    from oracle.params import det_tensor

    for train in (True, False):
        rng = np.random.Generator(np.random.PCG64(4580 + int(train)))
        shape = (2, 3, 45, 80)
        img = torch.from_numpy(rng.random(shape, dtype=np.float32))
        idx = torch.from_numpy(np.stack([rng.integers(0, 45, (2, 250)), rng.integers(0, 80, (2, 250))], 2).astype(np.int64))   # (B, N / B, 2)
        net = Net2DSeg(num_classes=11, dual_head=True, backbone_2d="UNetResNet34", backbone_2d_kwargs={"pretrained": False}, output_all=True)
        net.load_state_dict({k: det_tensor(k, v.shape) for k, v in net.state_dict().items()})
        net.train(train)
        net.net_2d.dropout.p = 0.0
        out = net({"img": img, "img_indices": idx})
        save = {"out_feats_s4": out["feats"][::4], "out_seg_logit": out["seg_logit"], "out_seg_logit2": out["seg_logit2"],
                "out_seg_logit_all_s4": out["seg_logit_all"][:, ::4, ::4]}
        tag = "train" if train else "eval"
        if train:
            g = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape), dtype=np.float32)) for k, v in out.items()}
            sum((out[k] * g[k]).sum() for k in out).backward()
            named = dict(net.named_parameters())
            for k in ("net_2d.conv1.weight", "linear.weight", "linear2.weight", "net_2d.dec_conv_stage1.bias"):
                save["pgrad_" + k] = named[k].grad
            for k in ("net_2d.bn1.running_mean", "net_2d.layer4.2.bn2.running_var"):
                save["buf_" + k] = net.state_dict()[k]
            norms = {k: [float(p.grad.double().sum()), float(p.grad.double().norm())] for k, p in named.items()}
            with open(os.path.join(OUT, "g1c_net2dseg_45x80_c11_train_gradnorms.json"), "w") as f:
                json.dump(norms, f, indent=0)
        np.savez_compressed(os.path.join(OUT, f"g1c_net2dseg_45x80_c11_{tag}.npz"), **_np(save))
        print("G1c", tag, {k: tuple(v.shape) for k, v in out.items()})


# ----------------------------------------------------------------------------- G2
def gen_g2():
    from mopa.common.utils.loss import mask_cons_loss  # reference

    rng = np.random.Generator(np.random.PCG64(22))
    B, H, W, C = 3, 30, 46, 5
    logits = torch.from_numpy(rng.standard_normal((B, H, W, C), dtype=np.float32) * 2).requires_grad_(True)
    masks = []
    for b in range(B):
        m = rng.integers(0, 7, (H, W)).astype(np.int32)
        m[:10] = -100
        m[rng.random((H, W)) < 0.1] = -100
        if b == 2:
            m[:] = -100  # an image with no valid id still counts in the mean (loss.py:278-281)
        masks.append(torch.from_numpy(m))
    probs = F.softmax(logits, dim=3)  # caller: train_xmuda_mopa.py:473
    loss = mask_cons_loss(probs, masks, True)
    loss.backward()
    loss_noent = mask_cons_loss(F.softmax(logits.detach(), dim=3), masks, False)
    np.savez_compressed(os.path.join(OUT, "g2_mask_cons.npz"), logits=logits.detach().numpy(),
                        masks=np.stack([m.numpy() for m in masks]), loss=loss.detach().numpy(),
                        loss_noent=np.asarray(float(loss_noent)), grad_logits=logits.grad.numpy())
    print("G2", float(loss), float(loss_noent))


# ----------------------------------------------------------------------------- G3
def gen_g3():
    rng = np.random.Generator(np.random.PCG64(33))
    N, C = 400, 5
    a = torch.from_numpy(rng.standard_normal((N, C), dtype=np.float32) * 3).requires_grad_(True)
    b = torch.from_numpy(rng.standard_normal((N, C), dtype=np.float32) * 3)
    lab = torch.from_numpy(rng.integers(0, C, N).astype(np.int64))
    lab[rng.random(N) < 0.15] = -100
    w = torch.tensor([2.68678412, 4.36182969, 5.47896839, 3.89026883, 1.0])
    # same expressions as train_xmuda_mopa.py:389-393 and :354-358
    kl = F.kl_div(F.log_softmax(a, dim=1), F.softmax(b.detach(), dim=1), reduction="none").sum(1).mean()
    ga_kl, = torch.autograd.grad(kl, a)
    ce = F.cross_entropy(a, lab, weight=w)
    ga_ce, = torch.autograd.grad(ce, a)
    ce_now = F.cross_entropy(a, lab)
    ga_ce_now, = torch.autograd.grad(ce_now, a)
    np.savez_compressed(os.path.join(OUT, "g3_kl_ce.npz"), a=a.detach().numpy(), b=b.numpy(), label=lab.numpy(),
                        weight=w.numpy(), kl=kl.detach().numpy(), grad_kl=ga_kl.numpy(), ce=ce.detach().numpy(),
                        grad_ce=ga_ce.numpy(), ce_noweight=ce_now.detach().numpy(), grad_ce_noweight=ga_ce_now.numpy())
    print("G3", float(kl), float(ce))


# ----------------------------------------------------------------------------- G4
def gen_g4():
    """Voxeliser pins.  Besides the reference's outputs we record the reference's own rotated points and the three
    uniform draws it used for the translation (replayed from the same seed), so that a device voxeliser can be
    checked bit-exactly on the stages after the rotation."""
    from mopa.data.utils.augmentation_3d import augment_and_scale_3d  # reference

    save = {}
    for k in range(3):
        rng = np.random.Generator(np.random.PCG64(40 + k))
        pts = (rng.standard_normal((500, 3)) * np.array([20, 20, 1.5])).astype(np.float32)
        kw = dict(noisy_rot=0.1 * (k > 0), flip_y=0.5 * (k > 0), rot_z=6.2831 * (k > 1), transl=k > 0)
        np.random.seed(k)
        coords, aug = augment_and_scale_3d(pts, 20, 4096, **kw)
        # replay the draws of augmentation_3d.py:26-59 in order to capture the translation's rand(3)
        np.random.seed(k)
        if kw["noisy_rot"] > 0:
            np.random.randn(3, 3)
        if kw["flip_y"] > 0:
            np.random.randint(0, 2)
        if kw["rot_z"] > 0:
            np.random.rand()
        u = np.random.rand(3) if kw["transl"] else np.zeros(3)
        ci = coords.astype(np.int64)  # nuscenes_dataloader.py:419
        keep = (ci.min(1) >= 0) & (ci.max(1) < 4096)  # :422-424
        save[f"points{k}"], save[f"coords{k}"], save[f"keep{k}"] = pts, ci, keep
        save[f"aug_points{k}"], save[f"u{k}"], save[f"transl{k}"] = np.asarray(aug, np.float32), u, np.asarray(kw["transl"])
    np.savez_compressed(os.path.join(OUT, "g4_voxelize.npz"), **save)
    print("G4 ok")


# ----------------------------------------------------------------------------- G5
def gen_g5():
    from mopa.data.utils.refine_pseudo_labels import refine_pseudo_labels  # reference
    from mopa.models.losses import prob_2_entropy  # reference
    from mopa.models.metric import SegIoU  # reference

    rng = np.random.Generator(np.random.PCG64(55))
    N, C = 300, 5
    logit = torch.from_numpy(rng.standard_normal((N, C), dtype=np.float32) * 2)
    prob = F.softmax(logit, 1)
    maxp, lab = prob.max(1)
    refined = refine_pseudo_labels(maxp.numpy().copy(), lab.numpy().copy())
    ent = prob_2_entropy(prob)
    gt = torch.from_numpy(rng.integers(0, C, N).astype(np.int64))
    gt[rng.random(N) < 0.2] = -100
    m = SegIoU(C, name="iou")
    m.update_dict({"seg_logit": logit}, {"seg_label": gt})
    m.update_dict({"seg_logit": logit.flip(0)}, {"seg_label": gt})
    np.savez_compressed(os.path.join(OUT, "g5_misc.npz"), logit=logit.numpy(), refined=refined, entropy=ent.numpy(),
                        gt=gt.numpy(), iou_mat=m.mat.numpy(), iou=m.iou.numpy())
    print("G5 ok")


# ----------------------------------------------------------------------------- G6
def gen_g6():
    """Structure pin of the 3D branch: the reference's own constructors (scn_unet.py:9-34, :38-219, xmuda_arch.py:82-126)
    run under the recording ``sparseconvnet`` stand-in; what they build and the order they execute it in is committed as
    JSON.  See oracle/scn_recorder.py for what this does and does not pin."""
    from mopa.models.scn_unet import UNetSCN, UNetSCN_ED  # reference
    from mopa.models.xmuda_arch import Net3DSeg  # reference
    rec = _scn_recorder()

    def shapes(mod):   # a LIST: module order is part of the pin (optimizer state is indexed by parameter order)
        return [[k, list(v.shape)] for k, v in mod.state_dict().items()]

    def run(mod, cin):
        rec.set_root(mod)
        out = mod([torch.zeros(4, 4, dtype=torch.int64), torch.ones(4, cin)])
        assert isinstance(out, rec.Sym)
        return [dict(t) for t in rec.TRACE], out.C

    def arith(trace):   # the arithmetic layer sequence with BatchNormLeakyReLU(leak 0) == BatchNormReLU folded
        seq = []
        for t in trace:
            op = {"BatchNormLeakyReLU": "BatchNormReLU"}.get(t["op"], t["op"])
            if op == "JoinTable":
                seq.append((op, tuple(t["parts"]), tuple(t["part_ops"]), t["level_out"]))
            elif op == "AddTable":
                seq.append((op, t["cout"], t["level_out"]))
            else:
                seq.append((op, t["cin"], t["cout"], t["level_in"], t["level_out"]))
        return seq

    doc = {"_about": "generated by oracle/gen_golden.py::gen_g6 from /root/reference under oracle/scn_recorder.py; "
                     "see that file's docstring for provenance"}
    kw = dict(in_channels=1, m=16, block_reps=1, residual_blocks=False, full_scale=4096, num_planes=7)  # config/xmuda.py:217-224
    variants = {"UNetSCN": kw, "UNetSCN_residual": dict(kw, residual_blocks=True),
                "UNetSCN_reps2": dict(kw, block_reps=2), "UNetSCN_residual_reps2": dict(kw, residual_blocks=True, block_reps=2),
                "UNetSCN_m32_planes5": dict(kw, m=32, num_planes=5)}
    for name, k in variants.items():
        rec.LOG.clear()
        net = UNetSCN(**k)
        ctor = [dict(c) for c in rec.LOG]
        trace, cout = run(net, k["in_channels"])
        doc[name] = {"kwargs": k, "out_channels": net.out_channels, "ctor_calls": ctor, "state_dict": shapes(net),
                     "trace": trace}
        assert cout == net.out_channels
    rec.LOG.clear()
    ed = UNetSCN_ED(1, m=16)
    ctor = [dict(c) for c in rec.LOG]
    trace, cout = run(ed, 1)
    doc["UNetSCN_ED"] = {"kwargs": {"in_channels": 1, "m": 16}, "ctor_calls": ctor, "state_dict": shapes(ed), "trace": trace}
    # the reference's unrolled network and scn.UNet (restated) must be the same arithmetic, layer for layer
    a, b = arith(doc["UNetSCN"]["trace"]), arith(trace)
    assert a == b, [(x, y) for x, y in zip(a, b) if x != y][:3]
    doc["UNetSCN_equals_UNetSCN_ED"] = True
    doc["layer_sequence"] = [list(map(lambda v: list(v) if isinstance(v, tuple) else v, t)) for t in a]
    for name, k in {"Net3DSeg_dual": dict(dual_head=True), "Net3DSeg_single": dict(dual_head=False),
                    "Net3DSeg_MCD": dict(dual_head=True, da_method="MCD")}.items():
        net = Net3DSeg(num_classes=5, backbone_3d="SCN", backbone_3d_kwargs=kw, **k)
        doc[name] = {"kwargs": k, "state_dict": shapes(net)}
    with open(os.path.join(OUT, "g6_scn_structure.json"), "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
    print("G6", len(doc["UNetSCN"]["state_dict"]), "tensors;", len(doc["layer_sequence"]), "ops; ED == UNet:", a == b)


# ----------------------------------------------------------------------------- G8
def vgi_fv_objects(k: int):
    """The objects of G8b: those of vgi_case(k) with the first one mirrored behind the sensor (mean x < 0)."""
    objs = [o.copy() for o in vgi_case(k)["objs"]]
    objs[0][:, 0] -= 2.0 * objs[0][:, 0].mean() + 3.0
    return objs


def vgi_case(k: int):
    """Synthetic VGI inputs of case k (shared by the generator and the tests; nothing reference-derived): a nuScenes-shape
    scan, a ground mask, two object clusters, a pinhole projection, the front axis."""
    from mopa_amd import synth
    front = "y" if k % 2 == 0 else "x"
    pts = synth.lidar_points(300 + k)
    if front == "y":   # nuScenes lidar frame: y forward; the generator's sweep is symmetric, so swap axes for variety
        pts = pts[:, [1, 0, 2]].copy()
    rng = np.random.Generator(np.random.PCG64(800 + k))
    ori_pc = np.concatenate([pts, rng.random((len(pts), 1), dtype=np.float32)], 1)   # (N, 4): xyz + intensity
    g_mask = pts[:, 2] < -1.7
    objs, labs = [], []
    shapes = (((4.2, 1.8, 1.5), (6.0, 9.0, -1.0)), ((0.8, 0.8, 1.7), (-3.0, 5.0, -0.9)))
    if k == 2:   # the object with the LARGEST extent (list index 1) fits nowhere: the first anchor attempt fails and the reference's
        # ignore_idx_ls bookkeeping (positions in extent order used as list indices, mixmatch_ss.py:123-196) decides what is inserted
        shapes = (shapes[0], ((22.0, 20.0, 1.5), (1.0, 14.0, -1.0)), shapes[1])
    for j, (size, c) in enumerate(shapes):
        n = 400 + 150 * j
        o = (rng.random((n, 3)) - 0.5) * np.array(size) + np.array(c)
        if front == "x":
            o = o[:, [1, 0, 2]]
        objs.append(np.concatenate([o, rng.random((n, 1))], 1).astype(np.float32))
        labs.append(np.full(n, 1 + j, np.int64))
    fx, fy, cx, cy = 1266.0, 1266.0, 800.0, 450.0
    if front == "y":
        proj = np.array([[fx, cx, 0, 0], [0, cy, -fy, 0], [0, 1, 0, 0]], np.float64)
    else:
        proj = np.array([[cx, -fx, 0, 0], [cy, 0, -fy, 0], [1, 0, 0, 0]], np.float64)
    label = rng.integers(0, 5, len(pts)).astype(np.int64)
    return dict(ori_pc=ori_pc, label=label, g_mask=g_mask, objs=objs, obj_labels=labs, proj=proj, image_size=(1600, 900), front=front)


def gen_g8():
    """VGI pins: the reference's check_overlap / point_mixmatch(ground) / range_projection / post_process on synthetic
    inputs.  Environment shims (none of them changes the reference's arithmetic): torchsparse.sparse_quantize restated from
    v1.4.0 (oracle/vgi.py), ``Tensor.cuda()`` -> identity (no GPU in the build container; the reference runs conv3d / where
    there), ``np.bool8`` (removed in numpy 2) -> ``np.bool_``, the visualisation module stubbed."""
    from oracle import vgi as ovgi
    ts = types.ModuleType("torchsparse")
    ts.SparseTensor = object
    tsu, tsq, tsc = types.ModuleType("torchsparse.utils"), types.ModuleType("torchsparse.utils.quantize"), types.ModuleType("torchsparse.utils.collate")
    tsq.sparse_quantize = ovgi.sparse_quantize
    tsc.sparse_collate = lambda *a, **k: None
    vis = types.ModuleType("mopa.data.utils.visualize")
    vis.debug_visualizer = vis.draw_range_image_labels = lambda *a, **k: None
    sys.modules.update({"torchsparse": ts, "torchsparse.utils": tsu, "torchsparse.utils.quantize": tsq,
                        "torchsparse.utils.collate": tsc, "pypatchworkpp": types.ModuleType("pypatchworkpp"),
                        "mopa.data.utils.visualize": vis})
    if not hasattr(np, "bool8"):
        np.bool8 = np.bool_
    torch.Tensor.cuda = lambda self, *a, **k: self
    from mopa.data import mixmatch_ss as ref  # reference
    from mopa.data.utils.augmentation_3d import range_projection  # reference

    save = {}
    for k in range(3):
        c = vgi_case(k)
        # (a) candidate centres of the larger object
        vc = ref.check_overlap(c["ori_pc"], c["objs"][0][:, :3], voxel_size=0.5, search_range=[25.0, 25.0], z_min=-2.0,
                               z_max=None, front_axis=c["front"])
        g = ovgi.overlap_grid(c["ori_pc"], c["objs"][0][:, :3], 0.5, (25.0, 25.0), -2.0, None, c["front"])
        assert vc is not None and np.array_equal(vc, ovgi.check_overlap(c["ori_pc"], c["objs"][0][:, :3], 0.5, (25.0, 25.0), -2.0, None, c["front"]))
        save[f"free_bits{k}"], save[f"free_shape{k}"] = np.packbits(g["free"].reshape(-1)), np.array(g["free"].shape)
        save[f"n_centers{k}"], save[f"centers_head{k}"], save[f"centers_sum{k}"] = np.array(len(vc)), vc[:64], vc.sum(0)
        # (b) the whole ground-mode insertion (overlap test -> filters -> ground cells -> placement matrices)
        np.random.seed(100 + k)
        cat_pc, cat_label, obj_mask, obj_ps_mask = ref.point_mixmatch(
            c["ori_pc"], c["label"], [o.copy() for o in c["objs"]], c["obj_labels"], insert_mode="ground", search_voxel_size=0.5,
            search_range=[25.0, 25.0], search_z_min=-2.0, proj_matrix=c["proj"], image_size=c["image_size"],
            g_indices=c["g_mask"], front_axis=c["front"])
        assert obj_mask.any(), "fixture case must insert"
        n0 = len(c["ori_pc"])
        save[f"obj_xyz{k}"] = cat_pc[n0:]                  # the transformed object points (float64)
        save[f"cat_label_tail{k}"] = cat_label[n0:]
        # (c) range-image occlusion culling on the concatenated cloud
        pres = range_projection(np.concatenate((cat_pc[:, :3], np.ones((cat_pc.shape[0], 1))), axis=1), 0.05235, -0.43633, 1024, 64,
                                crop=False, obj_mask=obj_mask)["pres_idx"]
        save[f"pres_bits{k}"], save[f"n_cat{k}"] = np.packbits(pres), np.array(len(pres))
        # (d) post_process: culling + augment_and_scale_3d + int cast + field filter + collate
        np.random.seed(200 + k)
        aug = {"noisy_rot": 0.1, "flip_y": 0.5, "rot_z": 6.2831, "transl": True}
        cat_input, cat_ps, om, _ = ref.post_process([cat_pc], [cat_label], [obj_mask], 20, 4096, aug, scan_pth_ls=["g8"], use_proj=True, backbone="SCN")
        locs = cat_input["x"][0].numpy()
        key = (locs[:, 0] << 24) | (locs[:, 1] << 12) | locs[:, 2]
        save[f"locs_n{k}"], save[f"locs_head{k}"] = np.array(len(locs)), locs[:128]
        save[f"locs_keysum{k}"], save[f"locs_keyxor{k}"] = np.array(int(key.sum())), np.array(int(np.bitwise_xor.reduce(key)))
        save[f"ps_sum{k}"], save[f"om_sum{k}"] = np.array(int(cat_ps.sum())), np.array(int(om.sum()))
        print("G8 case", k, c["front"], "centres", len(vc), "inserted", int(obj_mask.sum()), "kept", int(pres.sum()), "of", len(pres),
              "locs", len(locs))
    np.savez_compressed(os.path.join(OUT, "g8_vgi.npz"), **save)
    # G8b: insert_mode="fv" (mixmatch_ss.py:83-105) -- one object behind the sensor (rotated to the front), one in front (kept)
    save = {}
    for k in range(2):
        c = vgi_case(k)
        objs = [o.copy() for o in c["objs"]]
        objs[0][:, 0] -= 2.0 * objs[0][:, 0].mean() + 3.0          # mean x < 0: takes the rotation branch
        cat_pc, cat_label, obj_mask, _ = ref.point_mixmatch(c["ori_pc"], c["label"], objs, c["obj_labels"], z_disc=-0.324, insert_mode="fv")
        n0 = len(c["ori_pc"])
        assert obj_mask.sum() == sum(len(o) for o in objs) and cat_pc.dtype == np.float32
        save[f"obj_xyz{k}"], save[f"cat_label_tail{k}"] = cat_pc[n0:], cat_label[n0:]
        save[f"objs_after{k}_0"], save[f"objs_after{k}_1"] = objs[0], objs[1]    # the reference edits its inputs in place
        print("G8b case", k, "fv: inserted", int(obj_mask.sum()), "mean x before/after", float(c["objs"][0][:, 0].mean() * -1 - 3.0), float(cat_pc[n0:n0 + len(objs[0]), 0].mean()))
    np.savez_compressed(os.path.join(OUT, "g8b_vgi_fv.npz"), **save)


# ----------------------------------------------------------------------------- G7
def gen_g7():
    """Integer pins of the synthetic scan (our own generator; not reference-derived)."""
    from mopa_amd import synth
    from oracle import scn3d

    pins = {}
    for seed in (0, 1):
        s = synth.make_scan(seed)
        c = np.concatenate([s["coords"], np.zeros((len(s["coords"]), 1), np.int64)], 1)
        g = scn3d.Geometry(c)
        pins[str(seed)] = {"n_points": int(c.shape[0]), "extent": s["coords"].max(0).tolist(),
                           "active": g.num_active, "rules": g.num_rules}
    with open(os.path.join(OUT, "g7_synth_pins.json"), "w") as f:
        json.dump(pins, f, indent=1)
    print("G7", pins["0"])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _install_stubs()
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
    else:
        gen_g1(), gen_g1b(), gen_g1c(), gen_g2(), gen_g3(), gen_g4(), gen_g5(), gen_g6(), gen_g7(), gen_g8()
