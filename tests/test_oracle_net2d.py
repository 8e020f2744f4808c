"""Oracle 2D branch / losses vs the golden vectors produced by the imported reference."""
import json
import os

import numpy as np
import torch

from oracle import losses, net2d
from oracle.params import det_state


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _run(g, training):
    B = g["img"].shape[0]
    P = det_state(net2d.param_shapes(5, True))
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in P.items()}
    img = torch.from_numpy(g["img"]).requires_grad_(True)
    idx = [g[f"idx{i}"] for i in range(B)]
    out = net2d.net2dseg_forward(P, img, idx, training=training, dropout_p=0.0)
    return P, img, out


def test_g1_eval(golden_dir):
    for name in ("pad_eval", "nopad_eval"):
        g = _load(golden_dir, f"g1_net2dseg_{name}.npz")
        _, _, out = _run(g, False)
        for k in ("feats", "seg_logit", "seg_logit2", "seg_logit_all"):
            np.testing.assert_allclose(out[k].detach().numpy(), g["out_" + k], rtol=1e-4, atol=1e-5)


def test_g1_train_and_grads(golden_dir):
    g = _load(golden_dir, "g1_net2dseg_pad_train.npz")
    P, img, out = _run(g, True)
    for k in ("feats", "seg_logit", "seg_logit2", "seg_logit_all"):
        np.testing.assert_allclose(out[k].detach().numpy(), g["out_" + k], rtol=1e-4, atol=2e-5)
    sum((out[k] * torch.from_numpy(g["gin_" + k])).sum() for k in out).backward()
    np.testing.assert_allclose(img.grad.numpy(), g["grad_img"], rtol=2e-3, atol=2e-4)
    for k, v in g.items():
        if k.startswith("pgrad_"):
            ref = v
            np.testing.assert_allclose(P[k[6:]].grad.numpy(), ref, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(ref).max()))
        if k.startswith("buf_"):
            np.testing.assert_allclose(P[k[4:]].numpy(), v, rtol=1e-5, atol=1e-6)
    norms = json.load(open(os.path.join(golden_dir, "g1_net2dseg_pad_train_gradnorms.json")))
    for k, (s, n) in norms.items():
        assert abs(float(P[k].grad.double().norm()) - n) <= 2e-3 * n + 1e-5, k


def _g1c_inputs(train):
    """Inputs of fixture G1c, regenerated from the seed (oracle/gen_golden.py::gen_g1c): the index TENSOR form (B, N / B, 2) of the
    reference's own call, mopa/models/xmuda_arch.py:129-162."""
    rng = np.random.Generator(np.random.PCG64(4580 + int(train)))
    img = torch.from_numpy(rng.random((2, 3, 45, 80), dtype=np.float32))
    idx = torch.from_numpy(np.stack([rng.integers(0, 45, (2, 250)), rng.integers(0, 80, (2, 250))], 2).astype(np.int64))
    return rng, img, idx


def test_g1c_reference_call_shape_11_classes(golden_dir):
    """The oracle against the reference's outputs at the reference's own call shape scaled down (45 x 80 = the 225 x 400 aspect, pad
    48 x 80), 11 classes, train + eval, and the reference's parameter-gradient norms."""
    for train in (True, False):
        rng, img, idx = _g1c_inputs(train)
        g = _load(golden_dir, f"g1c_net2dseg_45x80_c11_{'train' if train else 'eval'}.npz")
        P = det_state(net2d.param_shapes(11, True))
        P = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in P.items()}
        out = net2d.net2dseg_forward(P, img, idx, training=train, dropout_p=0.0)
        np.testing.assert_allclose(out["feats"][::4].detach().numpy(), g["out_feats_s4"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(out["seg_logit"].detach().numpy(), g["out_seg_logit"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(out["seg_logit2"].detach().numpy(), g["out_seg_logit2"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(out["seg_logit_all"][:, ::4, ::4].detach().numpy(), g["out_seg_logit_all_s4"], rtol=1e-4, atol=2e-5)
        if not train:
            continue
        gin = {k: torch.from_numpy(rng.standard_normal(tuple(out[k].shape), dtype=np.float32)) for k in ("feats", "seg_logit_all", "seg_logit2", "seg_logit")}
        sum((out[k] * gin[k]).sum() for k in gin).backward()
        for k, v in g.items():
            if k.startswith("pgrad_"):
                np.testing.assert_allclose(P[k[6:]].grad.numpy(), v, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(v).max()))
            if k.startswith("buf_"):
                np.testing.assert_allclose(P[k[4:]].numpy(), v, rtol=1e-5, atol=1e-6)
        norms = json.load(open(os.path.join(golden_dir, "g1c_net2dseg_45x80_c11_train_gradnorms.json")))
        for k, (s_, n) in norms.items():
            assert abs(float(P[k].grad.double().norm()) - n) <= 5e-3 * n + 1e-5, k


def test_g2_mask_cons(golden_dir):
    g = _load(golden_dir, "g2_mask_cons.npz")
    logits = torch.from_numpy(g["logits"]).requires_grad_(True)
    masks = [torch.from_numpy(m) for m in g["masks"]]
    loss = losses.mask_cons_loss(torch.softmax(logits, 3), masks, True)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5)
    np.testing.assert_allclose(logits.grad.numpy(), g["grad_logits"], rtol=1e-4, atol=1e-8)
    l2 = losses.mask_cons_loss(torch.softmax(logits.detach(), 3), masks, False)
    np.testing.assert_allclose(float(l2), float(g["loss_noent"]), rtol=1e-5)


def test_g3_kl_ce(golden_dir):
    g = _load(golden_dir, "g3_kl_ce.npz")
    a = torch.from_numpy(g["a"]).requires_grad_(True)
    b, lab, w = torch.from_numpy(g["b"]), torch.from_numpy(g["label"]), torch.from_numpy(g["weight"])
    kl = losses.xm_kl(a, b)
    np.testing.assert_allclose(float(kl), float(g["kl"]), rtol=1e-6)
    np.testing.assert_allclose(torch.autograd.grad(kl, a)[0].numpy(), g["grad_kl"], rtol=1e-5, atol=1e-9)
    ce = losses.seg_ce(a, lab, w)
    np.testing.assert_allclose(float(ce), float(g["ce"]), rtol=1e-6)
    np.testing.assert_allclose(torch.autograd.grad(ce, a)[0].numpy(), g["grad_ce"], rtol=1e-5, atol=1e-9)
    ce2 = losses.seg_ce(a, lab, None)
    np.testing.assert_allclose(float(ce2), float(g["ce_noweight"]), rtol=1e-6)


def test_g5_segiou(golden_dir):
    g = _load(golden_dir, "g5_misc.npz")
    logit, gt = torch.from_numpy(g["logit"]), torch.from_numpy(g["gt"])
    mat = losses.seg_iou_update(None, logit, gt, 5)
    mat = losses.seg_iou_update(mat, logit.flip(0), gt, 5)
    assert (mat.numpy() == g["iou_mat"]).all()


def test_g4_voxel_coords(golden_dir):
    from oracle.voxelize import voxel_coords
    g = _load(golden_dir, "g4_voxelize.npz")
    for k in range(3):
        ci, keep = voxel_coords(g[f"aug_points{k}"], 20, 4096, g[f"u{k}"] if bool(g[f"transl{k}"]) else None)
        assert np.array_equal(ci, g[f"coords{k}"]) and np.array_equal(keep, g[f"keep{k}"])
