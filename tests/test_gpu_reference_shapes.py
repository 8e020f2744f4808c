"""The reference's OWN call shapes on the GPU (round-4 review, "untested shapes").

* `test_Net2DSeg` of mopa/models/xmuda_arch.py:129-162: Net2DSeg(11 classes, dual head) on 2 x 3 x 225 x 400 (the nuScenes resize of
  mopa/config/xmuda.py:98: pad to 240 x 400, layer4 at 15 x 25) with a (B, N / B, 2) index TENSOR of 2000 points;
* `test_Net3DSeg` (:165-216): Net3DSeg(11 classes) on 2000 uniformly random (N, 3) coordinates in 4096^3 -- isolated voxels, every
  27-offset rule table is the centre offset only, the deep levels are as long as level 0, BatchNorm over 2000 rows;
* fixture G1c: the reference's outputs at the same aspect scaled down (45 x 80, 11 classes, index tensor; oracle/gen_golden.py::gen_g1c);
* a joint 2D + 3D training step at 225 x 400 through the size-independent properties of tests/test_gpu_configs.py.
Tolerances: SURVEY 8c (per-layer rtol 1e-4 / atol 1e-5, end-to-end logits 1e-3 of their scale) unless a line says otherwise.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import net2d, scn3d
from oracle.params import det_tensor

pytestmark = pytest.mark.gpu


def _build_2d(C):
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    model, _ = build_model_2d(default_cfg(C, True))
    model.load_state_dict({k: det_tensor(k, v.shape) for k, v in model.state_dict().items()})
    model.net_2d.dropout.p = 0.0
    return model.cuda()


def _oracle_params(C, dtype):
    P = {k: (det_tensor(k, v).to(dtype) if "num_batches" not in k else det_tensor(k, v)) for k, v in net2d.param_shapes(C, True).items()}
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    return P


def _close(got, want, rtol, atol_of_scale):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    want = want.detach().cpu().numpy() if torch.is_tensor(want) else want
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol_of_scale * scale)


@pytest.mark.parametrize("train", [True, False])
def test_net2dseg_fixture_g1c_45x80_eleven_classes_index_tensor(golden_dir, train):
    """HIP path against the reference's own fp32 outputs (fixture G1c) in the reference's call form; in train mode every parameter
    gradient against the fp64 oracle with the reference's fp32 gradient norm as the yardstick."""
    rng = np.random.Generator(np.random.PCG64(4580 + int(train)))
    img = torch.from_numpy(rng.random((2, 3, 45, 80), dtype=np.float32))
    idx = torch.from_numpy(np.stack([rng.integers(0, 45, (2, 250)), rng.integers(0, 80, (2, 250))], 2).astype(np.int64))
    g = dict(np.load(os.path.join(golden_dir, f"g1c_net2dseg_45x80_c11_{'train' if train else 'eval'}.npz")))
    model = _build_2d(11).train(train)
    out = model({"img": img.cuda(), "img_indices": idx.cuda()})     # (the reference's test moves both to the device)
    assert out["seg_logit"].shape == (500, 11) and out["seg_logit_all"].shape == (2, 45, 80, 11)
    _close(out["feats"][::4], g["out_feats_s4"], 1e-3, 2e-4)
    _close(out["seg_logit"], g["out_seg_logit"], 1e-3, 2e-4)
    _close(out["seg_logit2"], g["out_seg_logit2"], 1e-3, 2e-4)
    _close(out["seg_logit_all"][:, ::4, ::4], g["out_seg_logit_all_s4"], 1e-3, 2e-4)
    if not train:
        return
    gin = {k: torch.from_numpy(rng.standard_normal(tuple(out[k].shape), dtype=np.float32)) for k in ("feats", "seg_logit_all", "seg_logit2", "seg_logit")}
    sum((out[k] * gin[k].cuda()).sum() for k in gin).backward()
    P = _oracle_params(11, torch.float64)
    ref = net2d.net2dseg_forward(P, img.double(), idx, training=True, dropout_p=0.0)
    sum((ref[k] * gin[k].double()).sum() for k in gin).backward()
    named = dict(model.named_parameters())
    norms = json.load(open(os.path.join(golden_dir, "g1c_net2dseg_45x80_c11_train_gradnorms.json")))
    gmax = max(float(P[k].grad.norm()) for k in norms)
    for k, (_, n) in norms.items():
        truth = P[k].grad
        tn = float(truth.norm())
        err = float((named[k].grad.double().cpu() - truth).norm())
        if tn <= 1e-6 * gmax:   # a conv bias in front of a BatchNorm: its true gradient is exactly zero
            assert err <= 1e-4 * gmax, (k, err)
            continue
        # layer4 lives on a 3 x 5 map here (30 samples per channel): the reference's own fp32 gradient is the yardstick, 1 % the floor
        assert err <= max(3.0 * abs(n - tn) + 0.01 * tn, 0.02 * tn), (k, err, tn, n)
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("buf_"):
            _close(sd[k[4:]], v, 1e-4, 1e-5)


@pytest.mark.parametrize("train,f4_roles", [(True, None), (False, None), (True, ("fwd",))])
def test_net2dseg_reference_call_shape_225x400_eleven_classes(train, f4_roles, monkeypatch):
    """mopa/models/xmuda_arch.py:129-162 as written: B = 2, 225 x 400, 2000 points as a (2, 1000, 2) tensor, 11 classes, dual head.
    Outputs against the fp64 oracle at 1e-3 of their scale (end to end), with the fp32 oracle's own distance as the
    yardstick; train mode also checks every parameter gradient's norm against the fp64 oracle.  f4_roles = ("fwd",) (= MOPA_WINOGRAD_F4_ROLES=fwd):
    only the forward pass uses Winograd F(4x4); backward-data and the weight gradient run the exact-product kernels (direct MFMA / F(2x2))
    and every parameter gradient is bounded at 1 % of its norm instead of 3 % -- the non-F(4x4) weight-gradient path pinned at the
    reference's own call shape."""
    from mopa_amd import dense2d
    if f4_roles is not None:
        monkeypatch.setattr(dense2d, "F4_ROLES", tuple(f4_roles))
    floor = 0.03 if f4_roles is None else 0.01
    B, H, W, N, C = 2, 225, 400, 2000, 11
    rng = np.random.Generator(np.random.PCG64(225400 + int(train)))
    img = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32))
    idx = torch.from_numpy(np.stack([rng.integers(0, H, (B, N // B)), rng.integers(0, W, (B, N // B))], 2).astype(np.int64))
    model = _build_2d(C).train(train)
    out = model({"img": img.cuda(), "img_indices": idx.cuda()})
    assert out["feats"].shape == (N, 64) and out["seg_logit"].shape == (N, C) and out["seg_logit2"].shape == (N, C)
    assert out["seg_logit_all"].shape == (B, H, W, C)
    P = _oracle_params(C, torch.float64)
    ref = net2d.net2dseg_forward(P, img.double(), idx, training=train, dropout_p=0.0)
    P32 = _oracle_params(C, torch.float32)
    ref32 = net2d.net2dseg_forward(P32, img, idx, training=train, dropout_p=0.0)
    for k in ("feats", "seg_logit", "seg_logit2", "seg_logit_all"):
        t = ref[k].detach().numpy()
        scale = float(np.abs(t).max())
        err = float(np.abs(out[k].detach().cpu().numpy() - t).max())
        yard = float(np.abs(ref32[k].detach().double().numpy() - t).max())
        assert err <= max(4.0 * yard, 1e-3 * scale), (k, err, yard, scale)
    if not train:
        e2 = model({"img": img.cuda(), "img_indices": idx.cuda()})
        assert all(torch.equal(out[k], e2[k]) for k in out)    # eval: no state moves
        return
    gin = {k: torch.from_numpy(rng.standard_normal(tuple(out[k].shape), dtype=np.float32)) for k in ("feats", "seg_logit_all", "seg_logit2", "seg_logit")}
    sum((out[k] * gin[k].cuda()).sum() for k in gin).backward()
    sum((ref[k] * gin[k].double()).sum() for k in gin).backward()
    sum((ref32[k] * gin[k]).sum() for k in gin).backward()
    named = dict(model.named_parameters())
    gmax = max(float(p.grad.norm()) for k, p in P.items() if p.requires_grad and p.grad is not None)
    worst = []
    for k, p in P.items():
        if not (p.requires_grad and p.grad is not None):
            continue
        tn = float(p.grad.norm())
        err = float((named[k].grad.double().cpu() - p.grad).norm())
        yard = float((P32[k].grad.double() - p.grad).norm())       # plain fp32 torch-CPU arithmetic against the fp64 truth
        if tn <= 1e-6 * gmax:
            assert err <= 1e-4 * gmax, (k, err)
            continue
        worst.append((err / tn, yard / tn, k))
        # the fp32 oracle's own distance from the fp64 truth is the yardstick (the network's gradient noise on the closed-form
        # weights: ReLU masks within round-off of zero); the floor is 3 % of each tensor's norm -- the stride-1 3x3 layers run
        # F(4x4) Winograd in all three passes ("deliberate deviations": 1.4 / 2.7 / 6.7 % median / p90 / max
        # against fp64 at the 302 x 480 bench shape)
        assert err <= max(4.0 * yard, floor * tn), (k, err, yard, tn)
    print("worst gradient errors (relative to the tensor's norm; HIP, fp32 oracle):", sorted(worst, reverse=True)[:5])
    sd = model.state_dict()
    for k in ("net_2d.bn1.running_mean", "net_2d.layer4.2.bn2.running_var", "net_2d.dec_conv_stage2.1.running_mean"):
        _close(sd[k], P[k].float(), 1e-4, 1e-5)
    assert int(sd["net_2d.bn1.num_batches_tracked"]) == 1


def _build_3d(C, in_channels=1):
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_3d
    cfg = default_cfg(C, True)
    cfg.MODEL_3D.SCN.in_channels = in_channels
    model, _ = build_model_3d(cfg)
    model.load_state_dict({k: det_tensor(k, v.shape) for k, v in model.state_dict().items()})
    return model.cuda()


@pytest.mark.parametrize("train", [True, False])
def test_net3dseg_reference_call_shape_2000_random_voxels_eleven_classes(train):
    """mopa/models/xmuda_arch.py:165-186 as written: 2000 uniformly random (N, 3) long coordinates in 4096^3 (no batch column:
    scn.InputLayer puts every point in sample 0), one random feature, 11 classes, dual head, the shipped 7-level UNet.  The voxels
    are isolated: the integer part must say so (2000 rows at level 0, every rule the centre rule), the logits and every gradient
    are checked against the fp64 oracle (yardstick: the fp32 oracle)."""
    from mopa_amd.sparse3d import Geometry3D
    rng = np.random.Generator(np.random.PCG64(20004096))
    coords = rng.integers(0, 4096, (2000, 3)).astype(np.int64)
    feats = torch.from_numpy(rng.random((2000, 1), dtype=np.float32))
    og = scn3d.Geometry(coords, 7, 4096)
    g = Geometry3D(torch.from_numpy(coords), 7, 4096, "cuda")
    assert g.num_active == og.num_active and g.num_active[0] == 2000
    for l in range(7):
        assert np.array_equal(g.nbr27[l].cpu().numpy(), og.nbr27[l]), l
    nb0 = og.nbr27[0]
    assert (nb0[13] == np.arange(2000)).all() and (np.delete(nb0, 13, 0) == -1).all()    # isolated voxels: the centre rule only
    model = _build_3d(11).train(train)
    if not train:   # a well-conditioned eval case: running statistics := this input's batch statistics (as tests/test_gpu_3d.py)
        P0 = {k: v.detach().cpu().double().clone() for k, v in scn3d.fold_state_dict(model.state_dict()).items()}
        old, scn3d.BN_MOMENTUM = scn3d.BN_MOMENTUM, 1.0
        try:
            scn3d.net3dseg_forward(P0, og, feats.double(), training=True, num_planes=7)
        finally:
            scn3d.BN_MOMENTUM = old
        model.load_state_dict({k: v.float() for k, v in P0.items()})
    sd_before = {k: v.clone() for k, v in scn3d.fold_state_dict(model.state_dict()).items()}
    f_dev = feats.cuda().requires_grad_(True)
    out = model({"x": [torch.from_numpy(coords), f_dev]})     # (N, 3) coordinates, as the reference's test passes them
    assert out["feats"].shape == (2000, 16) and out["seg_logit"].shape == (2000, 11) and out["seg_logit2"].shape == (2000, 11)
    gouts = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape), dtype=np.float32)) for k, v in out.items()}
    sum((out[k] * gouts[k].cuda()).sum() for k in out).backward()

    def run(dtype):
        P = {k: v.detach().cpu().to(dtype).clone() for k, v in sd_before.items()}
        for k in P:
            if "running" not in k:
                P[k].requires_grad_(True)
        f = feats.to(dtype).clone().requires_grad_(True)
        o = scn3d.net3dseg_forward(P, og, f, dual_head=True, training=train, num_planes=7)
        sum((o[k] * gouts[k].to(dtype)).sum() for k in o).backward()
        return P, f, o

    P, f, ref = run(torch.float64)
    P32, f32, ref32 = run(torch.float32)

    def close(got, truth, yard, what):
        scale = max(1e-6, float(np.abs(truth).max()))
        err = float(np.abs(got - truth).max())
        yerr = float(np.abs(yard - truth).max())
        assert err <= max(4.0 * yerr, 2e-4 * scale), (what, err, yerr, scale)
        # (isolated voxels make every row the same function of one scalar feature: whole columns sit near a ReLU's zero together and
        #  the fp32 oracle itself is up to 0.7 % off the fp64 gradients in eval mode -- the absolute cap follows the yardstick)
        assert err <= max(1e-2 * scale, 4.0 * yerr), (what, err, scale)
        return err / scale, yerr / scale

    for k in ("feats", "seg_logit", "seg_logit2"):
        close(out[k].detach().cpu().numpy(), ref[k].detach().numpy(), ref32[k].detach().double().numpy(), k)
    named = dict(model.named_parameters())
    worst = []
    for k, p in P.items():
        if p.requires_grad:
            worst.append(close(named[k].grad.cpu().numpy(), p.grad.numpy(), P32[k].grad.double().numpy(), k) + (k,))
    close(f_dev.grad.cpu().numpy(), f.grad.numpy(), f32.grad.double().numpy(), "dfeats")
    print("worst gradient errors (of the tensor's scale; HIP, fp32 oracle):", sorted(worst, reverse=True)[:5])
    if train:
        sd = model.state_dict()
        for k in P:
            if "running" in k:
                np.testing.assert_allclose(sd[k].cpu().numpy(), P[k].float().numpy(), rtol=1e-4, atol=1e-5)


def test_joint_step_at_225x400_properties():
    """A joint 2D + 3D training step at the shipped nuScenes resolution (mopa/config/xmuda.py:98: 400 x 225), 4 source + 4 target
    synthetic scans through the bench's stream schedule: the two halves' gradients add up, losses finite, and a scan of the batch is
    computed like the same scan alone (eval mode) -- the size-independent properties of tests/test_gpu_configs.py at this resolution."""
    from mopa_amd import synth
    from mopa_amd.common.utils.loss import seg_ce, xm_kl
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d, build_model_3d
    from mopa_amd.models.xmuda_arch import Net2DSeg
    from mopa_amd.step import DualStream
    B, H, W = 4, 225, 400
    batches = []
    for j in range(2):
        scans = [synth.make_scan(j * B + i, H=H, W=W) for i in range(B)]
        b = synth.collate(scans)
        batches.append(dict(locs=b["x"][0].cuda(), feats=b["x"][1].cuda(), label=b["seg_label"].cuda(), img=b["img"].cuda(),
                            pix=Net2DSeg.pack_indices(b["img_indices"], H, W, "cuda"), idx=b["img_indices"], scans=scans))
    src, trg = batches
    assert src["img"].shape == (B, 3, H, W)
    torch.manual_seed(0)
    cfg = default_cfg(num_classes=5, dual_head=True)
    m2, m3 = build_model_2d(cfg)[0].cuda().train(), build_model_3d(cfg)[0].cuda().train()
    m2.net_2d.dropout.p = 0.0
    cw = torch.tensor([2.68678412, 4.36182969, 5.47896839, 3.89026883, 1.0], device="cuda")
    dual = DualStream("cuda", order_2d_first=True)
    ready = torch.cuda.Event()
    ready.record()

    def half(b, lam, supervised):
        o2, o3 = dual.forward(m2, m3, {"img": b["img"], "point_pix_2d": b["pix"], "img_indices": None}, {"x": [b["locs"], b["feats"]]},
                              inputs_ready=ready)
        l2 = lam * xm_kl(o2["seg_logit2"], o3["seg_logit"])
        if supervised:
            l2 = l2 + seg_ce(o2["seg_logit"], b["label"], cw)
        with dual.on_side(o2["seg_logit"]):
            l3 = lam * xm_kl(o3["seg_logit2"], o2["seg_logit"])
            if supervised:
                l3 = l3 + seg_ce(o3["seg_logit"], b["label"], cw)
        l2.backward()
        l3.backward()
        dual.join()
        torch.cuda.synchronize()
        assert o2["seg_logit_all"].shape == (B, H, W, 5)
        return float(l2), float(l3)

    def grads():
        return [p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p) for m in (m2, m3) for p in m.parameters()]

    def zero():
        for m in (m2, m3):
            for p in m.parameters():
                p.grad = None

    state = [{k: v.clone() for k, v in m.state_dict().items()} for m in (m2, m3)]

    def restore():
        for m, s in zip((m2, m3), state):
            m.load_state_dict(s)

    l_src = half(src, 1.0, True)
    g_src = grads()
    zero(); restore()
    l_trg = half(trg, 0.1, False)
    g_trg = grads()
    zero(); restore()
    a = half(src, 1.0, True)
    half(trg, 0.1, False)
    g_both = grads()
    assert all(np.isfinite(x) for x in l_src + l_trg) and a == l_src
    for gs, gt, gb in zip(g_src, g_trg, g_both):
        ref = gs.double() + gt.double()
        scale = max(1e-12, float(ref.abs().max()))
        assert float((gb.double() - ref).abs().max()) <= 2e-6 * scale + 1e-12
    m2.eval(); m3.eval()
    with torch.no_grad():
        o2 = m2({"img": src["img"], "img_indices": src["idx"]})
        o3 = m3({"x": [src["locs"], src["feats"]]})
        i, n = 2, len(src["scans"][2]["coords"])
        n0 = sum(len(s["coords"]) for s in src["scans"][:i])
        s = src["scans"][i]
        c1 = torch.cat([torch.from_numpy(s["coords"]), torch.zeros(n, 1, dtype=torch.int64)], 1)
        o3_1 = m3({"x": [c1, torch.ones(n, 1)]})
        o2_1 = m2({"img": src["img"][i:i + 1], "img_indices": [src["idx"][i]]})
    for full, one, tol in ((o3["seg_logit"][n0:n0 + n], o3_1["seg_logit"], 1e-4), (o2["seg_logit"][n0:n0 + n], o2_1["seg_logit"], 1e-3),
                           (o2["seg_logit_all"][i], o2_1["seg_logit_all"][0], 1e-3)):
        scale = float(one.abs().max())
        assert float((full - one).abs().max()) <= tol * scale, float((full - one).abs().max()) / scale
