"""Parity of the HIP 2D branch (through the C-ABI) against torch-CPU functional ops and the reference-generated
golden fixture G1.  fp32 tolerances: rtol 1e-4 / atol 1e-5 per op (relative to the tensor scale), network-level
gradients 2e-3 of each tensor's max."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import net2d
from oracle.params import det_tensor

pytestmark = pytest.mark.gpu


def _nhwc(t):  # (B,C,H,W) cpu -> Img on cuda
    from mopa_amd.dense2d import Img
    B, C, H, W = t.shape
    return Img(t.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().cuda(), B, H, W)


def _nchw(img):
    return img.dense().reshape(img.B, img.H, img.W, img.C).permute(0, 3, 1, 2).cpu()


def _close(got, ref, rtol=1e-4, atol=2e-5):
    ref = ref.detach().float().numpy() if torch.is_tensor(ref) else ref
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol * max(1.0, float(np.abs(ref).max())))


@pytest.mark.parametrize("cin,cout,k,s,p,H,W", [(64, 64, 3, 1, 1, 19, 30), (64, 128, 3, 2, 1, 22, 36), (64, 128, 1, 2, 0, 22, 36),
                                                (128, 64, 3, 1, 1, 16, 48), (256, 256, 3, 1, 1, 5, 7), (128, 256, 3, 2, 1, 9, 15),
                                                (128, 256, 3, 1, 1, 6, 11), (256, 128, 3, 1, 1, 18, 30),   # Winograd fwd / dgrad / wgrad: F(2x2) below 8 pixels,
                                                (256, 256, 3, 1, 1, 9, 14), (128, 128, 3, 1, 1, 10, 13),    # F(4x4) above (ragged tiles)
                                                (64, 64, 3, 1, 1, 8, 12), (128, 64, 3, 1, 1, 32, 48), (256, 128, 3, 1, 1, 8, 12)])
def test_conv_fwd_dgrad_wgrad(cin, cout, k, s, p, H, W):
    from mopa_amd.dense2d import ConvOp, Img, new_img
    rng = np.random.Generator(np.random.PCG64(cin + cout + k))
    B = 2
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W), dtype=np.float32))
    w = torch.from_numpy(rng.standard_normal((cout, cin, k, k), dtype=np.float32) * 0.05)
    b = torch.from_numpy(rng.standard_normal(cout, dtype=np.float32))
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, s, p)
    gout = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
    (ref * gout.double()).sum().backward()
    op = ConvOp(w.cuda(), b.cuda(), k, s, p)
    # input lives in the right half of a wider (join) buffer
    wide = torch.zeros(B * H * W, 2 * cin, device="cuda")
    wide[:, cin:] = _nhwc(x).t
    xi = Img(wide, B, H, W, cin, cin)
    oh, ow = op.out_hw(H, W)
    out = new_img(B, oh, ow, cout, "cuda")
    op.forward(xi, out)
    _close(_nchw(out), ref)
    dx = new_img(B, H, W, cin, "cuda")
    dx.t.fill_(1.0)
    dw, db = torch.empty_like(op.w), torch.empty_like(op.b)
    op.backward(xi, _nhwc(gout), dx, dw, db, acc_dx=True)
    _close(_nchw(dx) - 1.0, xr.grad, rtol=1e-3, atol=1e-4)
    _close(dw, wr.grad, rtol=1e-3, atol=1e-4)
    _close(db, br.grad, rtol=1e-3, atol=1e-4)
    dx2 = new_img(B, H, W, cin, "cuda")
    op.backward(xi, _nhwc(gout), dx2, dw, db, acc_dx=False)
    _close(_nchw(dx2), xr.grad, rtol=1e-3, atol=1e-4)


def test_conv_transpose_fwd_bwd():
    from mopa_amd.dense2d import ConvTOp, new_img
    rng = np.random.Generator(np.random.PCG64(5))
    B, cin, cout, H, W = 2, 128, 64, 7, 9
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W), dtype=np.float32))
    w = torch.from_numpy(rng.standard_normal((cin, cout, 2, 2), dtype=np.float32) * 0.05)
    b = torch.from_numpy(rng.standard_normal(cout, dtype=np.float32))
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, br, 2)
    gout = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
    (ref * gout.double()).sum().backward()
    op = ConvTOp(w.cuda(), b.cuda())
    out = new_img(B, 2 * H, 2 * W, cout, "cuda")
    xi = _nhwc(x)
    op.forward(xi, out)
    _close(_nchw(out), ref)
    dx = new_img(B, H, W, cin, "cuda")
    dw, db = torch.empty_like(op.w), torch.empty_like(op.b)
    op.backward(xi, _nhwc(gout), dx, dw, db)
    _close(_nchw(dx), xr.grad, rtol=1e-3, atol=1e-4)
    _close(dw, wr.grad, rtol=1e-3, atol=1e-4)
    _close(db, br.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("G,C,res", [(2, 64, True), (2, 24, False), (3, 128, True)])
def test_grouped_batchnorm_is_bit_identical_to_one_call_per_group(G, C, res):
    """mopa_bn_act_{fwd,bwd}_groups (all row groups of a layer in one set of launches, blockIdx.y = group) against one mopa_bn_act_*
    call per row range, the form the reference's G separate passes correspond to: outputs, saved statistics, running statistics
    (updated group after group), input / residual / parameter gradients -- identical bits."""
    from mopa_amd.dense2d import _group, bn_bwd, bn_bwd_groups, bn_fwd, bn_fwd_groups, new_img
    rng = np.random.Generator(np.random.PCG64(3 + G + C))
    B, H, W = 2 * G, 9, 11
    mk = lambda scale=1.0, shift=0.0: _nhwc(torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32) * scale + shift))  # noqa: E731
    x, r, g = mk(2.0, 1.0), (mk() if res else None), mk()
    def params():
        return {"bn.weight": torch.linspace(0.5, 1.5, C).cuda(), "bn.bias": torch.linspace(-1, 1, C).cuda(),
                "bn.running_mean": torch.zeros(C, device="cuda"), "bn.running_var": torch.ones(C, device="cuda")}
    Pa, Pb = params(), params()
    ya, yb = new_img(B, H, W, C, "cuda"), new_img(B, H, W, C, "cuda")
    sa, sb = torch.empty(G, 4, C, device="cuda"), torch.empty(G, 4, C, device="cuda")
    for k in range(G):
        bn_fwd(_group(x, k, G), _group(ya, k, G), Pa, "bn", 1, None if r is None else _group(r, k, G), True, sa[k])
    bn_fwd_groups(x, yb, Pb, "bn", 1, r, True, sb, G)
    assert torch.equal(ya.t, yb.t) and torch.equal(sa, sb)
    assert torch.equal(Pa["bn.running_mean"], Pb["bn.running_mean"]) and torch.equal(Pa["bn.running_var"], Pb["bn.running_var"])
    out = []
    for grouped in (False, True):
        dx, dres = new_img(B, H, W, C, "cuda"), (new_img(B, H, W, C, "cuda") if res else None)
        dg, db = torch.full((C,), 0.25, device="cuda"), torch.full((C,), -0.5, device="cuda")   # accumulate into existing values
        if grouped:
            bn_bwd_groups(g, x, dx, sb, 1, yb if res else None, dres, False, True, dg, db, G, acc_params=True)
        else:
            for k in range(G):
                bn_bwd(_group(g, k, G), _group(x, k, G), _group(dx, k, G), sa[k], 1, _group(ya, k, G) if res else None,
                       None if dres is None else _group(dres, k, G), False, True, dg, db, acc_params=True)
        out.append((dx.t.clone(), None if dres is None else dres.t.clone(), dg, db))
    for a, b in zip(*out):
        assert (a is None and b is None) or torch.equal(a, b)


def test_maxpool_and_residual_bn():
    from mopa_amd._lib import call, ptr, stream
    from mopa_amd.dense2d import bn_bwd, bn_fwd, new_img
    rng = np.random.Generator(np.random.PCG64(6))
    B, C, H, W = 2, 64, 14, 18
    x = torch.from_numpy(np.maximum(rng.standard_normal((B, C, H, W), dtype=np.float32), 0))  # many ties at 0, like post-ReLU
    xr = x.double().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    gout = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
    (ref * gout.double()).sum().backward()
    xi = _nhwc(x)
    out = new_img(B, H // 2, W // 2, C, "cuda")
    amax = torch.empty(out.rows * C, dtype=torch.uint8, device="cuda")
    call("mopa_maxpool3x3s2_fwd", xi.p, xi.ld, B, H, W, C, out.p, out.ld, ptr(amax), stream())
    _close(_nchw(out), ref)
    dx = new_img(B, H, W, C, "cuda")
    go = _nhwc(gout)
    call("mopa_maxpool3x3s2_bwd", go.p, go.ld, ptr(amax), B, H, W, C, dx.p, dx.ld, 0, stream())
    _close(_nchw(dx), xr.grad)
    # BN (no act) -> + residual -> ReLU  == BasicBlock tail
    z = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32) * 2 + 1)
    idt = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32))
    P = {"bn.weight": torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)), "bn.bias": torch.from_numpy(rng.standard_normal(C).astype(np.float32)),
         "bn.running_mean": torch.zeros(C), "bn.running_var": torch.ones(C)}
    zr, ir = z.double().requires_grad_(True), idt.double().requires_grad_(True)
    gr, br = P["bn.weight"].double().requires_grad_(True), P["bn.bias"].double().requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    ref = F.relu(F.batch_norm(zr, rm, rv, gr, br, True, 0.1, 1e-5) + ir)
    g2 = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32))
    (ref * g2.double()).sum().backward()
    Pd = {k: v.cuda() for k, v in P.items()}
    zi, ii, y = _nhwc(z), _nhwc(idt), new_img(B, H, W, C, "cuda")
    stats = torch.empty(4, C, device="cuda")
    bn_fwd(zi, y, Pd, "bn", 1, ii, True, stats)
    _close(_nchw(y), ref)
    _close(Pd["bn.running_var"], rv, rtol=1e-5, atol=1e-6)
    dz, dres = new_img(B, H, W, C, "cuda"), new_img(B, H, W, C, "cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    bn_bwd(_nhwc(g2), zi, dz, stats, 1, y, dres, False, True, dg, db)
    _close(_nchw(dz), zr.grad, rtol=1e-3, atol=1e-4)
    _close(_nchw(dres), ir.grad)
    _close(dg, gr.grad, rtol=1e-3, atol=1e-4)
    _close(db, br.grad, rtol=1e-3, atol=1e-4)


def _build_2d(C=5):
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    model, _ = build_model_2d(default_cfg(C, True))
    model.load_state_dict({k: det_tensor(k, v.shape) for k, v in model.state_dict().items()})
    model.net_2d.dropout.p = 0.0
    return model.cuda()


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def test_net2dseg_golden_eval(golden_dir):
    for name in ("pad_eval", "nopad_eval"):
        g = _load(golden_dir, f"g1_net2dseg_{name}.npz")
        model = _build_2d().eval()
        B = g["img"].shape[0]
        out = model({"img": torch.from_numpy(g["img"]), "img_indices": [g[f"idx{i}"] for i in range(B)]})
        for k in ("feats", "seg_logit", "seg_logit2", "seg_logit_all"):
            _close(out[k], g["out_" + k], rtol=1e-3, atol=2e-4)


def test_net2dseg_golden_train_grads(golden_dir):
    g = _load(golden_dir, "g1_net2dseg_pad_train.npz")
    model = _build_2d().train()
    out = model({"img": torch.from_numpy(g["img"]), "img_indices": [g["idx0"], g["idx1"]]})
    for k in ("feats", "seg_logit", "seg_logit2", "seg_logit_all"):
        _close(out[k], g["out_" + k], rtol=1e-3, atol=2e-4)
    sum((out[k] * torch.from_numpy(g["gin_" + k]).cuda()).sum() for k in out).backward()
    named = dict(model.named_parameters())
    sd = model.state_dict()
    # Deep gradients of this tiny input (BN over 12 samples in layer4) are ill-conditioned: the reference's own fp32
    # result is ~1 % off the fp64 truth for conv1.weight.  So the truth is the fp64 oracle, and the golden
    # (reference, fp32) error against it is the yardstick for the HIP (fp32) error.
    P = {k: (det_tensor(k, v).double() if "num_batches" not in k else det_tensor(k, v)) for k, v in net2d.param_shapes(5, True).items()}
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    ref = net2d.net2dseg_forward(P, torch.from_numpy(g["img"]).double(), [g["idx0"], g["idx1"]], training=True, dropout_p=0.0)
    sum((ref[k] * torch.from_numpy(g["gin_" + k]).double()).sum() for k in ref).backward()
    # A ReLU whose pre-activation is ~1e-5 from zero can take the other branch in a different fp32 summation
    # order; on this tiny input the bottleneck (layer4 / decoder stage 5 live on a 2x3 map, 12-48 samples per
    # channel) turns ONE such flip into a several-percent change of every gradient that passes through it
    # (measured: exactly 1 of 12,288 masks differs from the fp64 oracle).  Those tensors get a loose bound.
    def bottleneck(name):
        return any(t in name for t in ("layer4", "dec_t_conv_stage5", "dec_conv_stage4", "layer3.5"))

    for k, v in g.items():
        if k.startswith("pgrad_"):
            truth = P[k[6:]].grad.numpy()
            scale = float(np.abs(truth).max())
            err = float(np.abs(named[k[6:]].grad.cpu().numpy() - truth).max())
            yard = float(np.abs(v - truth).max())
            bound = 0.1 * scale if bottleneck(k) else max(3.0 * yard, 2e-4 * scale)
            assert err <= bound, (k, err, yard, scale)
        if k.startswith("buf_"):
            _close(sd[k[4:]], v, rtol=1e-4, atol=1e-5)
    norms = json.load(open(os.path.join(golden_dir, "g1_net2dseg_pad_train_gradnorms.json")))
    gmax = max(float(P[k].grad.norm()) for k in norms)
    for k, (s, n) in norms.items():
        truth = float(P[k].grad.norm())
        got = float(named[k].grad.double().norm())
        bound = 0.1 * truth if bottleneck(k) else max(3.0 * abs(n - truth), 5e-3 * truth)
        assert abs(got - truth) <= bound + 1e-6 * gmax, (k, got, truth, n)
    assert int(sd["net_2d.bn1.num_batches_tracked"]) == 1


def test_net2dseg_vs_oracle_odd_size_and_dropout_semantics():
    rng = np.random.Generator(np.random.PCG64(8))
    B, H, W = 2, 45, 70
    img = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32))
    idx = [np.stack([rng.integers(0, H, 300), rng.integers(0, W, 300)], 1) for _ in range(B)]
    model = _build_2d(10).train()
    out = model({"img": img, "img_indices": idx})
    P = {k: det_tensor(k, v) for k, v in net2d.param_shapes(10, True).items()}
    ref = net2d.net2dseg_forward(P, img, idx, training=True, dropout_p=0.0)
    for k in ("feats", "seg_logit", "seg_logit2", "seg_logit_all"):
        _close(out[k], ref[k], rtol=1e-3, atol=2e-4)
    # dropout p=0.4: ~40 % zeros at the two sites changes the output but keeps it finite; eval ignores p
    model.net_2d.dropout.p = 0.4
    o1 = model({"img": img, "img_indices": idx})
    assert torch.isfinite(o1["seg_logit"]).all() and not torch.allclose(o1["seg_logit"], out["seg_logit"])
    model.eval()
    e1 = model({"img": img, "img_indices": idx})["seg_logit"]
    e2 = model({"img": img, "img_indices": idx})["seg_logit"]
    assert torch.equal(e1, e2)
    with pytest.raises(IndexError):
        model({"img": img, "img_indices": [idx[0], np.array([[H, 0]])]})


@pytest.mark.parametrize("cin,cout,H,W,acc", [(64, 64, 37, 51, False), (128, 64, 21, 30, True), (64, 128, 16, 24, False)])
def test_wino4_fused_gemm_output_vs_fp64_conv(cin, cout, H, W, acc, monkeypatch):
    """mopa_wino4_gemm_output (the 36 GEMMs + output transform of F(4x4) in one kernel, LDS-DMA staged) through dense2d.wino_conv,
    forced on for small shapes: against an fp64 conv3x3 (padding 1) and against the batched-GEMM + output-transform path; odd
    sizes (ragged tiles, a last row block beyond T), bias, and accumulation into an existing output."""
    import torch.nn.functional as F
    from mopa_amd import dense2d
    from mopa_amd._lib import call, ptr, stream
    rng = np.random.Generator(np.random.PCG64(400 + cin + W))
    B = 2
    x = torch.from_numpy(rng.standard_normal((B, H, W, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
    prev = torch.from_numpy(rng.standard_normal((B * H * W, cout)).astype(np.float32)).cuda()
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), None if acc else bias.double().cpu(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, cout)
    if acc:
        ref = ref + prev.double().cpu()
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(dense2d, "WINO4_FUSED_MIN_BLOCKS", 0 if fused else 1 << 62)
        assert dense2d.wino4_fused(cin, cout, B, H, W) == fused
        U = torch.empty(36, cout, cin, device="cuda") if fused else torch.empty(36, cin, cout, device="cuda")
        call("mopa_wino4_weight_t" if fused else "mopa_wino4_weight", ptr(w), cout, cin, 0, ptr(U), stream())
        out = prev.clone() if acc else torch.full((B * H * W, cout), float("nan"), device="cuda")
        dense2d.wino_conv(ptr(x), cin, B, H, W, cin, cout, U, None if acc else bias, ptr(out), cout, accumulate=acc, F=4)
        _close(out, ref.float().numpy(), rtol=1e-4, atol=3e-5)
        outs.append(out)
    _close(outs[0], outs[1].cpu(), rtol=1e-4, atol=2e-5)


def test_stem_dgrad_image_kernel_vs_torch():
    """mopa_stem_dgrad_image alone: backward-data of the 7x7 / stride 1 / padding 3 stem over the image window of the /16-padded
    frame, against fp64 autograd of F.conv2d on the zero-padded image (resnet34_unet.py:133-144)."""
    import torch.nn.functional as F
    from mopa_amd._lib import call, ptr, stream
    rng = np.random.Generator(np.random.PCG64(22))
    B, H, W = 2, 37, 50
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    w = torch.from_numpy(rng.standard_normal((64, 3, 7, 7)).astype(np.float32))
    dout = torch.from_numpy(rng.standard_normal((B, Hp, Wp, 64)).astype(np.float32))   # NHWC rows of 64
    img = torch.zeros(B, 3, H, W, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(F.pad(img, [0, Wp - W, 0, Hp - H]), w.double(), padding=3)
    (y * dout.permute(0, 3, 1, 2).double()).sum().backward()
    dimg = torch.empty(B, 3, H, W, device="cuda")
    dd, wd = dout.cuda().contiguous(), w.cuda().contiguous()
    call("mopa_stem_dgrad_image", ptr(dd), 64, B, Hp, Wp, H, W, ptr(wd), ptr(dimg), stream())
    _close(dimg, img.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_gradient_wrt_input_image_vs_oracle():
    """The reference's Net2DSeg is an ordinary autograd module: an image with requires_grad gets a gradient
    (xmuda_arch.py:49-79 through resnet34_unet.py:131-191).  Here that is the backward-data of the 7x7 stem restricted to the
    image window of the /16-padded frame (mopa_stem_dgrad_image, exact test above) behind the whole backward pass.  Truth: the
    fp64 oracle's autograd.  Tolerance: like every gradient that has crossed the whole network on a small input it carries the
    ReLU-mask-flip noise of the ill-conditioned bottleneck (see the parameter-gradient tests: few samples per channel in layer4)
    -- 3 % in the L2 norm, 5 % of the scale pointwise."""
    rng = np.random.Generator(np.random.PCG64(21))
    B, H, W = 2, 72, 104
    img_np = rng.random((B, 3, H, W), dtype=np.float32)
    idx = [np.stack([rng.integers(0, H, 400), rng.integers(0, W, 400)], 1) for _ in range(B)]
    gin = {k: rng.standard_normal(s).astype(np.float32) for k, s in (("seg_logit", (800, 5)), ("seg_logit2", (800, 5)))}
    model = _build_2d().train()
    model.net_2d.dropout.p = 0.0
    img = torch.from_numpy(img_np).cuda().requires_grad_(True)
    out = model({"img": img, "img_indices": idx})
    sum((out[k] * torch.from_numpy(v).cuda()).sum() for k, v in gin.items()).backward()
    assert img.grad is not None and img.grad.shape == img.shape
    P = {k: (det_tensor(k, v).double() if "num_batches" not in k else det_tensor(k, v)) for k, v in net2d.param_shapes(5, True).items()}
    img64 = torch.from_numpy(img_np).double().requires_grad_(True)
    ref = net2d.net2dseg_forward(P, img64, idx, training=True, dropout_p=0.0)
    sum((ref[k] * torch.from_numpy(v).double()).sum() for k, v in gin.items()).backward()
    truth = img64.grad.numpy()
    got = img.grad.double().cpu().numpy()
    scale = float(np.abs(truth).max())
    assert scale > 0
    assert float(np.linalg.norm(got - truth)) <= 3e-2 * float(np.linalg.norm(truth))
    assert float(np.abs(got - truth).max()) <= 5e-2 * scale, (float(np.abs(got - truth).max()), scale)
    # a CPU leaf image: the gradient flows back through the upload
    img_c = torch.from_numpy(img_np).requires_grad_(True)
    out = model({"img": img_c, "img_indices": idx})
    out["seg_logit"].sum().backward()
    assert img_c.grad is not None and img_c.grad.device.type == "cpu" and float(img_c.grad.abs().max()) > 0


def test_winograd_f4_network_level(monkeypatch):
    """F(4x4,3x3) at network level.  Backward passes only (MOPA_WINOGRAD_F4_ROLES=dgrad,wgrad): same logits bit for bit as
    without it, gradients as close to the fp64 oracle as the F(2x2) run.  Forward pass too (the default): logits within the
    stated tolerance of the oracle and 1e-4 of the exact-product forward."""
    from mopa_amd import dense2d
    import os
    if os.environ.get("MOPA_WINOGRAD", "1") == "0":
        pytest.skip("Winograd switched off for this run (MOPA_WINOGRAD=0)")
    rng = np.random.Generator(np.random.PCG64(11))
    B, H, W = 2, 160, 224
    img = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32))
    idx = [np.stack([rng.integers(0, H, 500), rng.integers(0, W, 500)], 1) for _ in range(B)]

    def run(roles, backward=True):
        monkeypatch.setattr(dense2d, "F4_ROLES", roles)
        model = _build_2d().train()
        model.net_2d.dropout.p = 0.0
        out = model({"img": img, "img_indices": idx})
        if backward:
            (out["seg_logit"].square().mean() + out["seg_logit2"].square().mean() + out["seg_logit_all"].square().mean()).backward()
        return {k: v.detach().cpu() for k, v in out.items()}, {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}

    used = []
    inner = dense2d.wino_conv
    monkeypatch.setattr(dense2d, "wino_conv", lambda *a, **kw: (used.append(kw.get("F", 2)), inner(*a, **kw))[1])
    import os
    if "MOPA_WINOGRAD_F4_ROLES" not in os.environ and os.environ.get("MOPA_WINOGRAD_F4", "1") != "0":
        assert dense2d.F4_ROLES == ("fwd", "dgrad", "wgrad")   # the shipped default
    o4, g4 = run(("dgrad", "wgrad"))
    assert 4 in used and 2 in used
    o2, g2 = run(())
    for k in o2:
        assert torch.equal(o4[k], o2[k])
    # truth = the fp64 oracle.  This random-init network is ill-conditioned in fp32 whatever the conv algorithm: the F(2x2) run
    # (and the direct kernels) are off by a median 1.3 % of each gradient's scale; F(4x4) in the backward passes must not add to it.
    P = {k: (det_tensor(k, v).double() if "num_batches" not in k else det_tensor(k, v)) for k, v in net2d.param_shapes(5, True).items()}
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    ref = net2d.net2dseg_forward(P, img.double(), idx, training=True, dropout_p=0.0)
    (ref["seg_logit"].square().mean() + ref["seg_logit2"].square().mean() + ref["seg_logit_all"].square().mean()).backward()
    worst = []
    for k in g2:
        truth = P[k].grad.float()
        scale = float(truth.abs().max())
        if scale < 1e-6:   # biases in front of a BatchNorm: the true gradient is zero
            continue
        e4, e2 = float((g4[k] - truth).abs().max()) / scale, float((g2[k] - truth).abs().max()) / scale
        if e4 > max(1.5 * e2, 2e-4):
            worst.append((k, e4, e2))
    assert not worst, worst[:8]
    # forward pass on F(4x4) too (the default): logits within the tolerance of the logit parity tests
    of, _ = run(("fwd", "dgrad", "wgrad"), backward=False)
    for k in ("feats", "seg_logit", "seg_logit2", "seg_logit_all"):
        _close(of[k], ref[k].detach().float(), rtol=1e-3, atol=2e-4)
        _close(of[k], o2[k], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("B,H,W", [(1, 128, 128), (1, 112, 128), (1, 126, 130), (2, 90, 46)])
def test_f4_forward_at_the_min_pixels_boundary(B, H, W, monkeypatch):
    """Image sizes around dense2d.F4_FWD_MIN_PIXELS (4,096 samples per channel: below it the FORWARD pass keeps the exact-product
    kernels because layers that normalise over few samples amplify ReLU flips): 128x128 puts layer1 exactly ON the threshold
    (64 x 64), 112x128 just below (56 x 64 = 3,584), 126x130 is odd and pads to 128x144, 2x90x46 mixes both sides across the
    stages.  Logits within the stated tolerance of the fp64 oracle, and within 1e-4 of the exact-product forward."""
    from mopa_amd import dense2d
    import os
    if os.environ.get("MOPA_WINOGRAD", "1") == "0" or os.environ.get("MOPA_WINOGRAD_F4", "1") == "0":
        pytest.skip("Winograd F(4x4) switched off for this run")
    rng = np.random.Generator(np.random.PCG64(100 + H + W))
    img = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32))
    idx = [np.stack([rng.integers(0, H, 200), rng.integers(0, W, 200)], 1) for _ in range(B)]
    picked = []
    inner = dense2d.wino_tile

    def spy(cin, cout, k, s, p, B_, H_, W_, role="fwd"):
        f = inner(cin, cout, k, s, p, B_, H_, W_, role)
        picked.append((role, B_ * H_ * W_, f))
        return f

    monkeypatch.setattr(dense2d, "wino_tile", spy)

    def run(roles):
        monkeypatch.setattr(dense2d, "F4_ROLES", roles)
        model = _build_2d().train()
        model.net_2d.dropout.p = 0.0
        return {k: v.detach().cpu() for k, v in model({"img": img, "img_indices": idx}).items()}

    of = run(("fwd", "dgrad", "wgrad"))
    fwd = [(n, f) for role, n, f in picked if role == "fwd"]
    assert all(f != 4 for n, f in fwd if n < dense2d.F4_FWD_MIN_PIXELS)          # below the threshold: never F(4x4) in the forward pass
    assert any(f == 4 for n, f in fwd if n >= dense2d.F4_FWD_MIN_PIXELS)         # at / above it: F(4x4) runs
    oe = run(("dgrad", "wgrad"))
    P = {k: (det_tensor(k, v).double() if "num_batches" not in k else det_tensor(k, v)) for k, v in net2d.param_shapes(5, True).items()}
    ref = net2d.net2dseg_forward(P, img.double(), idx, training=True, dropout_p=0.0)
    for k in ("feats", "seg_logit", "seg_logit2", "seg_logit_all"):
        _close(of[k], ref[k].detach().float(), rtol=1e-3, atol=2e-4)
        _close(of[k], oe[k], rtol=1e-3, atol=1e-4)


def test_cached_weight_layouts_follow_every_kind_of_weight_update():
    """Forward / backward-data weight layouts are cached per weight version: an optimizer step through the flat buffer
    (HIP kernel, invisible to autograd's version counters), a tracked in-place update and an EMA swap must all be seen."""
    from mopa_amd import synth
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    from mopa_amd.optim import FlatAdam
    from mopa_amd.pseudo import FlatEMA
    b = synth.make_batch(1, H=32, W=48)
    torch.manual_seed(0)
    m = build_model_2d(default_cfg())[0].cuda().eval()
    opt = FlatAdam(m.parameters(), lr=1e-3)
    ema = FlatEMA(opt, decay=0.5)

    def logits():
        with torch.no_grad():
            return m(b)["seg_logit"].clone()

    y0 = logits()
    assert torch.equal(logits(), y0)                       # cached layouts reproduce the same bits
    opt.grad.copy_(torch.randn_like(opt.grad))
    opt.step()                                             # flat-buffer Adam step
    y1 = logits()
    assert not torch.equal(y1, y0)
    with torch.no_grad():
        m.net_2d.layer1[0].conv1.weight.mul_(1.5)          # tracked in-place update of a cached layer
    y2 = logits()
    assert not torch.equal(y2, y1)
    with ema.average_parameters():                         # shadow = the initial weights
        assert torch.allclose(logits(), y0, rtol=1e-5, atol=1e-6)
    assert torch.equal(logits(), y2)
    # torch's convention: step(closure) re-evaluates the model and returns the loss (the closure runs with grad enabled)
    m.train()
    calls = []

    def closure():
        opt.zero_grad()
        loss = m(b)["seg_logit"].square().mean()
        loss.backward()
        calls.append(float(loss))
        return loss

    t0 = opt.t
    ret = opt.step(closure)
    assert len(calls) == 1 and float(ret) == calls[0] and opt.t == t0 + 1


@pytest.mark.parametrize("one_kernel", [False, True])
def test_batched_weight_form_refresh_gives_the_bits_of_the_single_kernels(one_kernel, monkeypatch):
    """After a weight update every stale igemm layout / Winograd transform of the network is rebuilt in ONE launch
    (mopa_conv2d_weight_forms_batched, shared device code with the one-form kernels): training steps with the batched refresh
    on and off must give identical logits.  one_kernel: the one-kernel convolutions forced on at this size, so that the fragment forms
    (layouts 2 and 3: mopa_wino4_weight_f / _q) are among the refreshed ones -- a form whose two builders differ in the last bit moves
    ReLU masks and with them every gradient by percent (found with layout 3 built in another translation unit)."""
    from mopa_amd import dense2d, synth
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    from mopa_amd.optim import FlatAdam
    b = synth.make_batch(2, H=64, W=96)
    if one_kernel:
        monkeypatch.setattr(dense2d, "WINO4_DIRECT_MIN_TILES", 0)
        monkeypatch.setattr(dense2d, "WINO4_WGRAD_FUSED_MIN_TILES", 0)

    def run(batched):
        monkeypatch.setattr(dense2d, "BATCHED_REFRESH", batched)
        dense2d._relayout_cache.clear()
        dense2d._refreshed.clear()
        torch.manual_seed(0)
        m = build_model_2d(default_cfg())[0].cuda().train()
        m.net_2d.dropout.p = 0.0
        opt = FlatAdam(m.parameters(), lr=1e-3)
        outs = []
        for _ in range(3):
            opt.zero_grad()
            o = m(b)
            (o["seg_logit"].square().mean() + o["seg_logit2"].square().mean()).backward()
            opt.step()
            outs.append(o["seg_logit"].detach().clone())
        return outs

    a, s = run(True), run(False)
    for x, y in zip(a, s):
        assert torch.equal(x, y)
    assert not torch.equal(a[0], a[1])


def test_net2dseg_well_conditioned_fixture_bounds_every_gradient_at_one_percent(golden_dir):
    """Fixture G1b (reference Net2DSeg, train mode, 2 x 64 x 96: layer4 sees 48 samples per channel).  Outputs against the
    reference's fp32 values; EVERY parameter gradient against the fp64 oracle within 1 % of its norm -- including the
    bottleneck tensors that the 30 x 46 fixture can only bound at 10 % (one BN-ReLU mask flip moves them by percent there)."""
    g = _load(golden_dir, "g1b_net2dseg_64x96_train.npz")
    rng = np.random.Generator(np.random.PCG64(64096))             # the generator's inputs, regenerated (oracle/gen_golden.py::gen_g1b)
    img = torch.from_numpy(rng.random((2, 3, 64, 96), dtype=np.float32))
    idx = [np.stack([rng.integers(0, 64, 200), rng.integers(0, 96, 200)], 1).astype(np.int64) for _ in range(2)]
    model = _build_2d().train()
    out = model({"img": img, "img_indices": idx})
    gin = {k: torch.from_numpy(rng.standard_normal(tuple(out[k].shape), dtype=np.float32)) for k in ("feats", "seg_logit_all", "seg_logit2", "seg_logit")}
    _close(out["feats"][::4], g["out_feats_s4"], rtol=1e-3, atol=2e-4)
    _close(out["seg_logit"], g["out_seg_logit"], rtol=1e-3, atol=2e-4)
    _close(out["seg_logit2"], g["out_seg_logit2"], rtol=1e-3, atol=2e-4)
    _close(out["seg_logit_all"][:, ::4, ::4], g["out_seg_logit_all_s4"], rtol=1e-3, atol=2e-4)
    sum((out[k] * gin[k].cuda()).sum() for k in gin).backward()
    named = dict(model.named_parameters())
    P = {k: (det_tensor(k, v).double() if "num_batches" not in k else det_tensor(k, v)) for k, v in net2d.param_shapes(5, True).items()}
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    ref = net2d.net2dseg_forward(P, img.double(), idx, training=True, dropout_p=0.0)
    sum((ref[k] * gin[k].double()).sum() for k in gin).backward()
    norms = json.load(open(os.path.join(golden_dir, "g1b_net2dseg_64x96_train_gradnorms.json")))
    gmax = max(float(P[k].grad.norm()) for k in norms)
    for k, (s, n) in norms.items():
        truth = P[k].grad
        tn = float(truth.norm())
        err = float((named[k].grad.double().cpu() - truth).norm())
        if tn <= 1e-6 * gmax:   # a conv bias in front of a BatchNorm: its true gradient is exactly zero (the batch mean absorbs it)
            assert err <= 1e-4 * gmax, (k, err)
            continue
        assert abs(n - tn) <= 1e-2 * tn, ("the reference's own fp32 gradient", k, n, tn)   # the fixture is well conditioned (0.18 % worst)
        assert err <= 1e-2 * tn, (k, err, tn)
    for k, v in g.items():                                  # element-wise on the stored slices (reference fp32 values)
        if k.startswith("pgrad_"):
            got = named[k[6:]].grad.cpu().numpy()[: v.shape[0]]
            scale = float(np.abs(P[k[6:]].grad.numpy()).max())
            assert float(np.abs(got - v).max()) <= 1e-2 * scale, k


def _replay_mode(monkeypatch, mode):
    """mode: False / None = the Python walk, True / "graph" = hipGraph replay (MOPA_GRAPH_2D=1), "native" = the recorded command
    list replayed by csrc/exec2d.hip (the default since round 5)."""
    from mopa_amd import dense2d
    monkeypatch.setattr(dense2d, "GRAPH_2D", mode in (True, "graph"))
    monkeypatch.setattr(dense2d, "NATIVE_2D", mode == "native")


REPLAY_MODES = ["graph", "native"]


def _graph_training_run(monkeypatch, graph, steps=4, pattern="alternate", seed_t=7):
    """`steps` iterations of (source batch fwd+bwd, target batch fwd+bwd, FlatAdam step) on 2 x 64 x 96 images with dropout 0.4
    and a different number of points per half.  -> logits of every pass, the final state_dict, dense2d.GRAPH_STATS."""
    from mopa_amd import dense2d, synth
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    from mopa_amd.optim import FlatAdam
    _replay_mode(monkeypatch, graph)
    for k in dense2d.GRAPH_STATS:
        dense2d.GRAPH_STATS[k] = 0
    src, trg = synth.make_batch(2, H=64, W=96), synth.make_batch(2, first=5, H=64, W=96)
    trg["img_indices"] = [a[: len(a) // 2] for a in trg["img_indices"]]   # the heads see a different N per half
    torch.manual_seed(seed_t)
    m = build_model_2d(default_cfg())[0].cuda().train()
    opt = FlatAdam(m.parameters(), lr=1e-3)
    outs = []

    def loss(o):
        return o["seg_logit"].square().mean() + o["seg_logit2"].square().mean() * 0.5 + o["feats"].mean()

    for _ in range(steps):
        opt.zero_grad()
        if pattern == "alternate":
            for b in (src, trg):
                o = m(b)
                loss(o).backward()
                outs.append(o["seg_logit"].detach().clone())
        else:   # both forwards first, one backward through both: the second forward finds the graph's activations in use
            o1, o2 = m(src), m(trg)
            (loss(o1) + loss(o2)).backward()
            outs += [o1["seg_logit"].detach().clone(), o2["seg_logit"].detach().clone()]
        opt.step()
    torch.cuda.synchronize()
    return outs, {k: v.detach().clone() for k, v in m.state_dict().items()}, dict(dense2d.GRAPH_STATS)


@pytest.mark.parametrize("mode", REPLAY_MODES)
def test_graph_replay_of_the_backbone_is_bit_identical_to_the_eager_pass(monkeypatch, mode):
    """dense2d.Graph2D: from the second pass of a shape on, the backbone's forward and backward launches are replayed from HIP
    graphs.  Same kernels, same order, same addresses: logits of every pass, every parameter, BatchNorm running statistics and
    num_batches_tracked after four iterations (dropout on, a new seed per pass, weight forms refreshed after each update) must
    equal the eager run bit for bit."""
    eo, es, est = _graph_training_run(monkeypatch, False)
    go, gs, gst = _graph_training_run(monkeypatch, mode)
    assert gst.get("native_lists", 0) == (2 if mode == "native" else 0)   # forward + backward, each recorded once
    assert est["forward_replays"] == 0 and gst["recorded"] == 1
    assert gst["forward_replays"] == 7 and gst["backward_replays"] == 7 and gst["eager_backward"] == 0   # 8 passes, the first eager
    for i, (a, b) in enumerate(zip(eo, go)):
        assert torch.equal(a, b), f"logits of pass {i} differ: {float((a - b).abs().max())}"
    assert not torch.equal(go[0], go[2])   # the weights did move
    for k in es:
        assert torch.equal(es[k], gs[k]), k


@pytest.mark.parametrize("mode", REPLAY_MODES)
def test_graph_replay_of_a_grouped_pass(monkeypatch, mode):
    """bn_groups = 2 under graph replay: one dropout seed per group in device memory, BatchNorm per group inside the graphs."""
    from mopa_amd import dense2d, synth
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    from mopa_amd.optim import FlatAdam
    src, trg = synth.make_batch(2, H=64, W=96), synth.make_batch(2, first=5, H=64, W=96)
    batch = {"img": torch.cat([src["img"], trg["img"]]), "img_indices": list(src["img_indices"]) + list(trg["img_indices"]), "bn_groups": 2}

    def run(graph):
        _replay_mode(monkeypatch, graph)
        for k in dense2d.GRAPH_STATS:
            dense2d.GRAPH_STATS[k] = 0
        torch.manual_seed(9)
        m = build_model_2d(default_cfg())[0].cuda().train()
        opt = FlatAdam(m.parameters(), lr=1e-3)
        outs = []
        for _ in range(4):
            opt.zero_grad()
            o = m(batch)
            (o["seg_logit"].square().mean() + o["seg_logit2"].square().mean()).backward()
            opt.step()
            outs.append(o["seg_logit"].detach().clone())
        torch.cuda.synchronize()
        return outs, {k: v.detach().clone() for k, v in m.state_dict().items()}, dict(dense2d.GRAPH_STATS)

    eo, es, _ = run(False)
    go, gs, st = run(mode)
    assert st["forward_replays"] == 3 and st["backward_replays"] == 3, st
    for a, b in zip(eo, go):
        assert torch.equal(a, b)
    for k in es:
        assert torch.equal(es[k], gs[k]), k


@pytest.mark.parametrize("mode", REPLAY_MODES)
def test_graph_replay_steps_aside_when_its_activations_are_still_in_use(monkeypatch, mode):
    """Two forwards, then one backward through both: the second forward must not replay over the activations the first one's
    backward still needs -- it runs eagerly; results equal the all-eager run."""
    eo, es, _ = _graph_training_run(monkeypatch, False, steps=3, pattern="both")
    go, gs, gst = _graph_training_run(monkeypatch, mode, steps=3, pattern="both")
    assert gst["eager_busy"] >= 2 and gst["forward_replays"] >= 2
    for a, b in zip(eo, go):
        assert torch.equal(a, b)
    for k in es:
        assert torch.equal(es[k], gs[k]), k


@pytest.mark.parametrize("mode", REPLAY_MODES)
def test_graph_replay_without_attached_gradients_and_after_moved_parameters(monkeypatch, mode):
    """(1) torch.optim.SGD with set_to_none: no attached .grad buffers -> the forward replays, the backward walks the recorded tape
    eagerly and autograd receives the gradient tensors.  (2) Re-pointing .data of a parameter drops the graphs (they hold raw
    addresses): the next passes run eagerly and record again."""
    from mopa_amd import dense2d, synth
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    b = synth.make_batch(2, H=64, W=96)

    def run(graph):
        _replay_mode(monkeypatch, graph)
        for k in dense2d.GRAPH_STATS:
            dense2d.GRAPH_STATS[k] = 0
        torch.manual_seed(3)
        m = build_model_2d(default_cfg())[0].cuda().train()
        opt = torch.optim.SGD(m.parameters(), lr=1e-3)
        outs = []
        for it in range(5):
            opt.zero_grad(set_to_none=True)
            if it == 3:
                w = m.net_2d.layer2[0].conv1.weight
                w.data = w.data.clone()
            o = m(b)
            (o["seg_logit"].square().mean() + o["seg_logit2"].square().mean()).backward()
            opt.step()
            outs.append(o["seg_logit"].detach().clone())
        torch.cuda.synchronize()
        return outs, dict(dense2d.GRAPH_STATS)

    eo, _ = run(False)
    go, st = run(mode)
    assert st["forward_replays"] >= 2 and st["eager_backward"] >= 2 and st["backward_replays"] == 0 and st["dropped"] == 1, st
    for a, c in zip(eo, go):
        assert torch.equal(a, c)


@pytest.mark.parametrize("winograd", [False, True])
def test_source_and_target_batch_in_one_pass_equal_two_calls(winograd, monkeypatch):
    """data_batch["bn_groups"] = 2: source and target images go through the backbone as one batch, with BatchNorm statistics,
    running-statistics updates and dropout masks per half in call order.  Against the two separate calls (the reference's loop,
    train_xmuda_mopa.py:342,426), num_batches_tracked identical, logits and running statistics equal to fp32 round-off, and
    * with the direct (exact-product) convolution kernels the per-image arithmetic is the same in both forms: parameter gradients
      agree to 1e-4 of each tensor's scale (one reduction over 4 images instead of 2 + 2 accumulated);
    * with the default algorithm choice the 3x3 layers may pick another Winograd tile for the larger batch (the thresholds
      count pixels per launch): gradients then agree like two algorithms do on this tiny, badly conditioned input (dropout 0.4,
      a few hundred samples per BatchNorm channel in layer4; profiles/f4_gradient_noise.py has the same figures against fp64):
      5 % of each tensor's L2 norm."""
    if not winograd:
        from mopa_amd import dense2d
        monkeypatch.setattr(dense2d, "WINOGRAD", False)
    from mopa_amd import synth
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    src, trg = synth.make_batch(2, H=64, W=96), synth.make_batch(2, first=5, H=64, W=96)

    def loss(o, w):
        return (o["seg_logit"] * w[0]).sum() + (o["seg_logit2"] * w[1]).sum() + (o["seg_logit_all"] * w[2]).sum()

    def weights(o, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        return [torch.randn(o[k].shape, device="cuda", generator=g) for k in ("seg_logit", "seg_logit2", "seg_logit_all")]

    def model():
        torch.manual_seed(11)
        m = build_model_2d(default_cfg())[0].cuda().train()
        m.output_all = True   # the full-image head takes part (the MoPA SAM-mask loss reads seg_logit_all)
        return m

    a = model()
    outs_a = []
    for it in range(2):
        for k, b in enumerate((src, trg)):
            o = a(b)
            loss(o, weights(o, 10 * it + k)).backward()
            outs_a.append(o["seg_logit"].detach().clone())
    b2 = model()
    outs_b = []
    ns = sum(len(i) for i in src["img_indices"])
    for it in range(2):
        o = b2({"img": torch.cat([src["img"], trg["img"]]), "img_indices": list(src["img_indices"]) + list(trg["img_indices"]),
                "bn_groups": 2})
        os_ = {k: v[:ns] if k != "seg_logit_all" else v[:2] for k, v in o.items()}
        ot_ = {k: v[ns:] if k != "seg_logit_all" else v[2:] for k, v in o.items()}
        (loss(os_, weights(os_, 10 * it)) + loss(ot_, weights(ot_, 10 * it + 1))).backward()
        outs_b += [os_["seg_logit"].detach().clone(), ot_["seg_logit"].detach().clone()]
    torch.cuda.synchronize()
    for x, y in zip(outs_a, outs_b):
        _close(y, x.cpu(), rtol=1e-4, atol=1e-5)
    sa, sb = a.state_dict(), b2.state_dict()
    for k in sa:
        if k.endswith("num_batches_tracked"):
            assert int(sa[k]) == int(sb[k]) == 4, k
        else:
            _close(sb[k], sa[k].cpu(), rtol=1e-4, atol=1e-5)
    worst = 0.0
    # (a bias in front of a BatchNorm has a zero gradient in exact arithmetic -- what it holds is round-off: norms below 1e-3 of the
    #  largest tensor's are measured against that floor)
    floor = 1e-3 * max(float(p.grad.norm()) for p in a.parameters() if p.grad is not None)
    for (n, p), (_, q) in zip(a.named_parameters(), b2.named_parameters()):
        if p.grad is None:
            assert q.grad is None, n
            continue
        if winograd:
            err = float((p.grad - q.grad).norm() / (p.grad.norm() + floor))
            assert err <= 5e-2, (n, err)
        else:
            err = float((p.grad - q.grad).abs().max()) / (float(p.grad.abs().max()) + 1e-20)
            assert err <= 1e-4, (n, err)
        worst = max(worst, err)
    print("worst parameter-gradient difference (relative):", worst)
    with pytest.raises(ValueError):
        b2({"img": src["img"][:1].repeat(3, 1, 1, 1), "img_indices": list(src["img_indices"][:1]) * 3, "bn_groups": 2})


@pytest.mark.parametrize("G,B,C,H,W", [(1, 2, 64, 13, 18), (2, 4, 128, 9, 11), (1, 1, 64, 16, 24)])
def test_input_transform_with_batchnorm_applied_on_the_way_in(G, B, C, H, W):
    """mopa_wino4_input_bn on a BatchNorm's input x (+ the layer's stats, mopa_bn_act_fwd_groups with y = null) against
    mopa_wino4_input on the materialised relu(batchnorm(x)): identical bits, including the zero padding around the image and
    the ragged last tile row / column; the statistics-only call leaves stats and running statistics as the full call does."""
    from mopa_amd._lib import call, ptr, stream
    from mopa_amd.dense2d import bn_fwd_groups, new_img
    rng = np.random.Generator(np.random.PCG64(17 + C + H))
    x = _nhwc(torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32) * 1.5 + 0.5))
    def params():
        return {"bn.weight": torch.linspace(0.5, 1.5, C).cuda(), "bn.bias": torch.linspace(-1, 1, C).cuda(),
                "bn.running_mean": torch.zeros(C, device="cuda"), "bn.running_var": torch.ones(C, device="cuda")}
    Pa, Pb = params(), params()
    y = new_img(B, H, W, C, "cuda")
    sa, sb = torch.empty(G, 4, C, device="cuda"), torch.empty(G, 4, C, device="cuda")
    bn_fwd_groups(x, y, Pa, "bn", 1, None, True, sa, G)
    bn_fwd_groups(x, None, Pb, "bn", 1, None, True, sb, G)
    assert torch.equal(sa, sb)
    assert torch.equal(Pa["bn.running_mean"], Pb["bn.running_mean"]) and torch.equal(Pa["bn.running_var"], Pb["bn.running_var"])
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    Va, Vb = torch.empty(36 * T * C, device="cuda"), torch.full((36 * T * C,), 7.0, device="cuda")
    call("mopa_wino4_input", y.p, y.ld, B, H, W, C, ptr(Va), stream())
    call("mopa_wino4_input_bn", x.p, x.ld, B, H, W, C, ptr(sb), G, 0, ptr(Vb), stream())
    assert torch.equal(Va, Vb)
    assert float((y.t < 0).sum()) == 0 and float((y.t == 0).float().mean()) > 0.1   # (the ReLU did cut something)


needs_mfma = pytest.mark.skipif(os.environ.get("MOPA_CONV2D_MFMA", "1") == "0",
                                reason="the deferred-BatchNorm / fused stem paths ride on the MFMA weight-gradient kernels (MOPA_CONV2D_MFMA=0 switches them off)")


@needs_mfma
@pytest.mark.parametrize("groups", [1, 2])
def test_deferred_batchnorm_gives_the_bits_of_the_materialised_one(groups, monkeypatch):
    """DEFER_BN: bn1 of every ResNet block whose conv2 runs F(4x4) in the forward pass and in the weight gradient is applied inside
    conv2's input transform, its output never exists.  Whole network, training mode, two iterations (running statistics carry
    over): logits, every parameter gradient and every buffer bit-identical to the pass that writes the tensor out."""
    from mopa_amd import dense2d, synth
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    batch = synth.make_batch(2 * groups, H=96, W=128)
    if groups > 1:
        batch["bn_groups"] = groups

    def run(defer):
        monkeypatch.setattr(dense2d, "DEFER_BN", defer)
        torch.manual_seed(5)
        m = build_model_2d(default_cfg())[0].cuda().train()
        m.output_all = True
        outs = []
        for it in range(2):
            o = m(batch)
            g = torch.Generator(device="cuda").manual_seed(it)
            sum((o[k] * torch.randn(o[k].shape, device="cuda", generator=g)).sum() for k in ("seg_logit", "seg_logit2", "seg_logit_all")).backward()
            outs.append(o["seg_logit"].detach().clone())
        torch.cuda.synchronize()
        return outs, [p.grad.clone() for p in m.parameters() if p.grad is not None], [b.clone() for b in m.buffers()]

    calls = []
    inner = dense2d.call
    monkeypatch.setattr(dense2d, "call", lambda name, *a: (calls.append(name), inner(name, *a))[1])
    oa, ga, ba = run(False)
    n_off = calls.count("mopa_wino4_input_bn")
    ob, gb, bb = run(True)
    assert n_off == 0 and calls.count("mopa_wino4_input_bn") > 0, "no layer of this input size took the deferred path"
    for a, b in zip(oa + ga + ba, ob + gb + bb):
        assert torch.equal(a, b)


@pytest.mark.parametrize("cin,cout,B,H,W,wide_out", [(16, 64, 2, 23, 18, False), (48, 192, 1, 32, 40, True), (32, 64, 3, 9, 70, False), (80, 256, 2, 12, 12, True)])
def test_wino4_nine_point_convolution_channel_shapes_the_dispatcher_does_not_use(cin, cout, B, H, W, wide_out):
    """mopa_wino4_conv9 straight through the C ABI on the channel counts its contract allows (Cin % 16 == 0, Cout % 64 == 0) but
    dense2d.wino4_direct never sends (it wants 64-aligned inputs and at most 128 outputs): one to five 16-channel steps, several
    64-channel output blocks per tile group, the output as a column slice of a wider buffer -- against an fp64 conv3x3."""
    import torch.nn.functional as F
    from mopa_amd._lib import call, ptr, stream
    rng = np.random.Generator(np.random.PCG64(77 + cin + cout))
    x = torch.from_numpy(rng.standard_normal((B * H * W, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
    U = torch.empty(36, cin, cout, device="cuda")
    call("mopa_wino4_weight_q", ptr(w), cout, cin, 0, ptr(U), stream())
    ld_out = cout + 64 if wide_out else cout
    out = torch.full((B * H * W, ld_out), float("nan"), device="cuda")
    call("mopa_wino4_conv9", ptr(x), cin, ptr(U), ptr(bias), ptr(out, 64 if wide_out else 0), ld_out, B, H, W, cin, cout, 0, None, 1, 0, stream())
    ref = F.conv2d(x.reshape(B, H, W, cin).permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), bias.double().cpu(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, cout)
    got = out[:, 64:] if wide_out else out
    _close(got, ref.float().numpy(), rtol=1e-4, atol=3e-5)
    if wide_out:
        assert bool(torch.isnan(out[:, :64]).all())   # nothing outside the slice is written


@pytest.mark.parametrize("cin,cout,H,W,acc,dgrad,G", [(64, 64, 37, 51, False, False, 0), (128, 64, 21, 30, True, False, 0),
                                                      (64, 128, 16, 24, False, True, 0), (64, 64, 19, 22, False, False, 2),
                                                      (128, 128, 9, 13, True, True, 1)])
@pytest.mark.parametrize("tiles32", [0, 1])
def test_wino4_one_kernel_convolution_vs_fp64_conv(cin, cout, H, W, acc, dgrad, G, tiles32, monkeypatch, request):
    """mopa_wino4_conv (input transform, 36 GEMMs, output transform of F(4x4) in one kernel; V and M never written) -- both of its
    kernels: k_wino4_conv (16 tiles per work item) and k_wino4_conv32 (32 tiles, the two point halves of the output transform summed
    through LDS), forced with mopa_wino4_conv_tiles32 (by default the shape decides: 32 tiles from 8 items per CU on) -- through
    dense2d.wino_conv, forced on for small shapes: against an fp64 conv3x3 (padding 1; dgrad: the transposed convolution's weight form)
    and against the two-kernel path; ragged tiles, a tile group beyond T, the input as a column slice of a wider buffer, bias,
    accumulation, and (G > 0) a deferred BatchNorm + ReLU applied on the way in for G image groups."""
    import torch.nn.functional as F
    from mopa_amd import dense2d
    from mopa_amd._lib import call, ptr, stream
    from mopa_amd.dense2d import bn_fwd_groups
    call("mopa_wino4_conv_tiles32", tiles32)
    request.addfinalizer(lambda: call("mopa_wino4_conv_tiles32", -1))
    rng = np.random.Generator(np.random.PCG64(900 + cin + W))
    B = 2 if G != 1 else 3
    wide = torch.from_numpy(rng.standard_normal((B * H * W, cin + 64)).astype(np.float32)).cuda()
    xin = dense2d.Img(wide, B, H, W, 64, cin)                      # columns [64, 64 + cin) of a wider buffer
    x = wide[:, 64:].contiguous().reshape(B, H, W, cin)
    stats = None
    if G:
        P = {"bn.weight": torch.linspace(0.5, 1.5, cin).cuda(), "bn.bias": torch.linspace(-1, 1, cin).cuda(),
             "bn.running_mean": torch.zeros(cin, device="cuda"), "bn.running_var": torch.ones(cin, device="cuda")}
        stats = torch.empty(G, 4, cin, device="cuda")
        y = dense2d.new_img(B, H, W, cin, "cuda")
        bn_fwd_groups(xin, y, P, "bn", 1, None, True, stats, G)
        x = y.t.reshape(B, H, W, cin)                               # what the convolution sees
    w = torch.from_numpy((rng.standard_normal((cin, cout, 3, 3) if dgrad else (cout, cin, 3, 3)) * 0.05).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
    prev = torch.from_numpy(rng.standard_normal((B * H * W, cout)).astype(np.float32)).cuda()
    wref = w.flip(2, 3).transpose(0, 1) if dgrad else w            # backward-data of a conv = conv with the rotated, transposed filter
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wref.double().cpu(), None if acc else bias.double().cpu(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, cout)
    if acc:
        ref = ref + prev.double().cpu()
    O, I = (cin, cout) if dgrad else (cout, cin)
    outs = []
    for direct in (True, False):
        monkeypatch.setattr(dense2d, "WINO4_DIRECT", True)
        monkeypatch.setattr(dense2d, "WINO4_DIRECT_ROLES", ("fwd", "fwd_eval", "dgrad"))
        monkeypatch.setattr(dense2d, "WINO4_DIRECT_MIN_TILES", 0 if direct else 1 << 62)
        monkeypatch.setattr(dense2d, "WINO4_FUSED_MIN_BLOCKS", 0)
        assert dense2d.wino4_direct(cin, cout, B, H, W) == direct
        U = torch.empty(36, cin, cout, device="cuda") if direct else torch.empty(36, cout, cin, device="cuda")
        call("mopa_wino4_weight_f" if direct else "mopa_wino4_weight_t", ptr(w), O, I, int(dgrad), ptr(U), stream())
        out = prev.clone() if acc else torch.full((B * H * W, cout), float("nan"), device="cuda")
        V = dense2d.wino_conv(xin.p, xin.ld, B, H, W, cin, cout, U, None if acc else bias, ptr(out), cout, accumulate=acc, F=4,
                              bn_in=(stats, G, 0) if G else None, want_v=not acc)
        if direct and not acc:   # the by-product: the bits mopa_wino4_input writes for the tensor the convolution sees
            Vr = torch.empty_like(V)
            xs = x.reshape(B * H * W, cin).contiguous()
            call("mopa_wino4_input", ptr(xs), cin, B, H, W, cin, ptr(Vr), stream())
            assert torch.equal(V, Vr)
        assert (V is None) == (direct and acc)
        _close(out, ref.float().numpy(), rtol=1e-4, atol=3e-5)
        outs.append(out)
    _close(outs[0], outs[1].cpu(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("cin,cout,B,H,W,acc,G,c0", [(64, 64, 2, 37, 51, False, 0, 0), (128, 64, 3, 40, 64, True, 0, 0),
                                                    (64, 128, 2, 33, 32, False, 0, 0), (64, 64, 2, 36, 44, False, 2, 0),
                                                    (128, 64, 4, 32, 40, True, 2, 64), (32, 64, 1, 64, 64, False, 0, 0),
                                                    (96, 128, 2, 21, 27, False, 1, 32)])
def test_wino4_one_kernel_weight_gradient_vs_fp64(cin, cout, B, H, W, acc, G, c0):
    """mopa_wino4_wgrad_fused (csrc/wino4wg.hip: the F(4x4) weight gradient from x and dY in one kernel, neither V nor dM in HBM)
    against torch autograd's fp64 conv2d weight gradient (the reference's arithmetic behind resnet34_unet.py:97-110) and against the
    two-operand form (mopa_wino4_input[_bn] + mopa_wino4_dout + mopa_wino4_bwd_weight): ragged tiles (H, W not multiples of 4), a tile
    range that ends inside a chunk, both operands as column slices of wider buffers, accumulation into an existing OIHW gradient, and
    (G > 0) a deferred BatchNorm + ReLU applied on the way in for G image groups, with (c0 > 0) the lower channels passing through."""
    import torch.nn.functional as F
    from mopa_amd import dense2d
    from mopa_amd._lib import call, ptr, query, stream
    from mopa_amd.dense2d import bn_fwd_groups
    rng = np.random.Generator(np.random.PCG64(77 + cin + W))
    wide = torch.from_numpy(rng.standard_normal((B * H * W, cin + 32)).astype(np.float32)).cuda()
    if c0:
        wide[:, 32:32 + c0].abs_()                                   # the pass-through half of a join buffer is non-negative already
    xin = dense2d.Img(wide, B, H, W, 32, cin)
    x = wide[:, 32:].contiguous()
    stats = None
    if G:
        cn = cin - c0
        P = {"bn.weight": torch.linspace(0.5, 1.5, cn).cuda(), "bn.bias": torch.linspace(-1, 1, cn).cuda(),
             "bn.running_mean": torch.zeros(cn, device="cuda"), "bn.running_var": torch.ones(cn, device="cuda")}
        stats = torch.empty(G, 4, cn, device="cuda")
        y = dense2d.new_img(B, H, W, cn, "cuda")
        bn_fwd_groups(dense2d.Img(wide, B, H, W, 32 + c0, cn), y, P, "bn", 1, None, True, stats, G)
        x = torch.cat([x[:, :c0], y.t], 1).contiguous()              # what the convolution saw
    dwide = torch.from_numpy(rng.standard_normal((B * H * W, cout + 64)).astype(np.float32)).cuda()
    dy = dwide[:, 64:].contiguous()
    prev = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)).cuda()
    xr = x.reshape(B, H, W, cin).permute(0, 3, 1, 2).double().cpu().requires_grad_(False)
    wr = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, None, padding=1).backward(dy.reshape(B, H, W, cout).permute(0, 3, 1, 2).double().cpu())
    ref = wr.grad + (prev.double().cpu() if acc else 0)
    assert query("mopa_wino4_wgrad_fused_ok", B, H, W, cin, cout) == 1
    ws = torch.empty(query("mopa_wino4_wgrad_fused_workspace_bytes", B, H, W, cin, cout), dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):   # twice: deterministic
        dw = prev.clone() if acc else torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
        call("mopa_wino4_wgrad_fused", xin.p, xin.ld, ptr(stats), max(G, 1), c0, ptr(dwide, 64), dwide.shape[1], B, H, W, cin, cout,
             ptr(dw), int(acc) | 2, ptr(ws), ws.numel(), stream())
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    scale = float(ref.abs().max())
    err = float((outs[0].double().cpu() - ref).abs().max()) / scale
    assert err < 2e-5, err
    if cin % 64:
        return
    # the two-operand form on the same inputs
    dw2 = prev.clone() if acc else torch.empty(cout, cin, 3, 3, device="cuda")
    dense2d.wino_wgrad(dense2d.LazyImg(xin, stats, G, c0) if G else xin, dense2d.Img(dwide, B, H, W, 64, cout), cin, cout, dw2,
                       accumulate=acc, F=4, fused=False)
    assert float((outs[0] - dw2).abs().max()) / scale < 2e-5


@pytest.mark.parametrize("F,cin,cout,B,H,W,acc", [(4, 128, 128, 2, 32, 64, False), (4, 256, 128, 1, 64, 64, True), (4, 128, 384, 4, 16, 16, False),
                                                  (2, 128, 256, 2, 16, 32, True), (4, 256, 256, 16, 38, 60, False)])
def test_transform_domain_weight_gradient_gemm_vs_fp64(F, cin, cout, B, H, W, acc, monkeypatch):
    """The ring-buffered LDS-DMA GEMM behind mopa_wino4_bwd_weight / mopa_wino_bwd_weight for 128-aligned channels (csrc/wgemm.hip:
    dU[p] = V[p]^T dM[p], 128 x 128 blocks, split-K slabs) -- through dense2d.wino_wgrad (input transform, dout transform, GEMM, G^T dU G)
    against torch autograd's fp64 conv2d weight gradient, and against k_conv2d_wgrad_mfma (MOPA_WGEMM=0 in a second process is the A/B
    switch; here: the same call on a tile count that is NOT a multiple of 16, which the GEMM refuses).  One range and several ranges of
    tiles, both transform sizes, accumulation into an existing OIHW gradient, the layer3 shape of the bench (2400 tiles)."""
    import torch.nn.functional as Fn
    from mopa_amd import dense2d
    T = B * ((H + F - 1) // F) * ((W + F - 1) // F)
    assert T % 16 == 0
    rng = np.random.Generator(np.random.PCG64(5 + cin + H))
    x = torch.from_numpy(rng.standard_normal((B * H * W, cin)).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((B * H * W, cout)).astype(np.float32)).cuda()
    prev = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)).cuda()
    wr = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    Fn.conv2d(x.reshape(B, H, W, cin).permute(0, 3, 1, 2).double().cpu(), wr, None, padding=1).backward(
        dy.reshape(B, H, W, cout).permute(0, 3, 1, 2).double().cpu())
    ref = wr.grad + (prev.double().cpu() if acc else 0)
    outs = []
    for _ in range(2):
        dw = prev.clone() if acc else torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
        dense2d.wino_wgrad(dense2d.Img(x, B, H, W), dense2d.Img(dy, B, H, W), cin, cout, dw, accumulate=acc, F=F, fused=False)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])   # deterministic
    scale = float(ref.abs().max())
    err = float((outs[0].double().cpu() - ref).abs().max()) / scale
    assert err < (3e-5 if F == 4 else 1e-5), err


@pytest.mark.parametrize("cin,cout,B,H,W,acc,dgrad,G,c0", [(64, 64, 2, 37, 51, False, False, 0, 0), (128, 64, 3, 21, 30, True, False, 0, 0),
                                                          (64, 128, 2, 16, 24, False, True, 0, 0), (64, 64, 2, 19, 22, False, False, 2, 0),
                                                          (128, 128, 3, 9, 13, True, True, 1, 0), (128, 128, 1, 64, 64, False, False, 0, 0),
                                                          (128, 64, 2, 40, 36, False, False, 2, 64), (64, 64, 16, 76, 120, False, False, 0, 0)])
def test_wino4_nine_point_one_kernel_convolution_vs_fp64_conv(cin, cout, B, H, W, acc, dgrad, G, c0, monkeypatch):
    """mopa_wino4_conv9 (csrc/wino4c9.hip: the one-kernel F(4x4) convolution with nine transform points per wave on v_mfma_f32_32x32x2_f32,
    raw patches staged by LDS-DMA, the four partial output transforms exchanged through LDS) through dense2d.wino_conv against an fp64
    conv3x3 (padding 1; dgrad: the transposed convolution's weight form) and against the first form (mopa_wino4_conv): ragged tiles, tile
    groups beyond T, the input as a column slice of a wider buffer, bias, accumulation, a deferred BatchNorm + ReLU on the way in for G
    image groups (c0 > 0: the lower channels pass through), and a shape with many workgroups in flight
    (the LDS-DMA ordering bug of its first version only showed under load)."""
    import torch.nn.functional as F
    from mopa_amd import dense2d
    from mopa_amd._lib import call, ptr, stream
    from mopa_amd.dense2d import bn_fwd_groups
    rng = np.random.Generator(np.random.PCG64(9000 + cin + W))
    wide = torch.from_numpy(rng.standard_normal((B * H * W, cin + 64)).astype(np.float32)).cuda()
    if c0:
        wide[:, 64:64 + c0].abs_()
    xin = dense2d.Img(wide, B, H, W, 64, cin)
    x = wide[:, 64:].contiguous()
    stats = None
    if G:
        cn = cin - c0
        P = {"bn.weight": torch.linspace(0.5, 1.5, cn).cuda(), "bn.bias": torch.linspace(-1, 1, cn).cuda(),
             "bn.running_mean": torch.zeros(cn, device="cuda"), "bn.running_var": torch.ones(cn, device="cuda")}
        stats = torch.empty(G, 4, cn, device="cuda")
        y = dense2d.new_img(B, H, W, cn, "cuda")
        bn_fwd_groups(dense2d.Img(wide, B, H, W, 64 + c0, cn), y, P, "bn", 1, None, True, stats, G)
        x = torch.cat([x[:, :c0], y.t], 1).contiguous()
    w = torch.from_numpy((rng.standard_normal((cin, cout, 3, 3) if dgrad else (cout, cin, 3, 3)) * 0.05).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
    prev = torch.from_numpy(rng.standard_normal((B * H * W, cout)).astype(np.float32)).cuda()
    wref = w.flip(2, 3).transpose(0, 1) if dgrad else w
    ref = F.conv2d(x.reshape(B, H, W, cin).permute(0, 3, 1, 2).double().cpu(), wref.double().cpu(), None if acc else bias.double().cpu(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, cout)
    if acc:
        ref = ref + prev.double().cpu()
    O, I = (cin, cout) if dgrad else (cout, cin)
    monkeypatch.setattr(dense2d, "WINO4_DIRECT", True)
    monkeypatch.setattr(dense2d, "WINO4_DIRECT_ROLES", ("fwd", "fwd_eval", "dgrad"))
    monkeypatch.setattr(dense2d, "WINO4_DIRECT_MIN_TILES", 0)
    monkeypatch.setattr(dense2d, "WINO4_DIRECT_MAX_CIN", 512)
    outs = []
    for nine in (True, False):
        if not nine and cin % 64:
            break
        monkeypatch.setattr(dense2d, "WINO4_CONV9", nine)
        assert dense2d.wino4_layout(cin, cout, B, H, W, "fwd_eval") == (3 if nine else 2)
        U = torch.empty(36, cin, cout, device="cuda")
        call("mopa_wino4_weight_q" if nine else "mopa_wino4_weight_f", ptr(w), O, I, int(dgrad), ptr(U), stream())
        U._mopa_wino_layout = (4, 3 if nine else 2)
        out = prev.clone() if acc else torch.full((B * H * W, cout), float("nan"), device="cuda")
        V = dense2d.wino_conv(xin.p, xin.ld, B, H, W, cin, cout, U, None if acc else bias, ptr(out), cout, accumulate=acc, F=4,
                              bn_in=(stats, G, c0) if G else None, role="fwd_eval", want_v=False)
        assert V is None
        _close(out, ref.float().numpy(), rtol=1e-4, atol=3e-5)
        outs.append(out)
    if len(outs) == 2:
        _close(outs[0], outs[1].cpu(), rtol=1e-4, atol=2e-5)
    o2 = prev.clone() if acc else torch.empty_like(outs[0])   # deterministic
    U = torch.empty(36, cin, cout, device="cuda")
    call("mopa_wino4_weight_q", ptr(w), O, I, int(dgrad), ptr(U), stream())
    call("mopa_wino4_conv9", xin.p, xin.ld, ptr(U), None if acc else ptr(bias), ptr(o2), cout, B, H, W, cin, cout, int(acc),
         ptr(stats), max(G, 1), c0, stream())
    assert torch.equal(o2, outs[0])


def test_one_kernel_convolution_picks_32_tiles_per_item_on_the_long_layers(request):
    """mopa_wino4_conv by shape: the decoder's full-resolution layer (64 -> 128 backward-data at 4 x 304 x 480: 1140 tile groups x 2 = 2280
    work items of 32 tiles >= 8 per CU) runs k_wino4_conv32 -- the bits of the forced 32-tile kernel -- and agrees with the 16-tile kernel
    to fp32 round-off (the two point halves are summed in a different order); a short layer (64 -> 64 at 1 x 64 x 64) keeps the 16-tile
    kernel's bits."""
    from mopa_amd._lib import call, ptr, stream
    request.addfinalizer(lambda: call("mopa_wino4_conv_tiles32", -1))
    g = torch.Generator(device="cuda").manual_seed(11)
    for (B, H, W, cin, cout, want32) in ((4, 304, 480, 64, 128, True), (1, 64, 64, 64, 64, False)):
        x = torch.randn(B * H * W, cin, device="cuda", generator=g)
        w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05
        U = torch.empty(36, cin, cout, device="cuda")
        call("mopa_wino4_weight_f", ptr(w), cout, cin, 0, ptr(U), stream())
        outs = {}
        for mode in (-1, 0, 1):
            call("mopa_wino4_conv_tiles32", mode)
            o = torch.empty(B * H * W, cout, device="cuda")
            call("mopa_wino4_conv", ptr(x), cin, ptr(U), None, ptr(o), cout, B, H, W, cin, cout, 0, None, 1, 0, None, stream())
            outs[mode] = o
        torch.cuda.synchronize()
        assert torch.equal(outs[-1], outs[1 if want32 else 0])
        assert not torch.equal(outs[0], outs[1])
        scale = float(outs[0].abs().max())
        assert float((outs[0] - outs[1]).abs().max()) <= 2e-5 * scale


def test_one_kernel_convolution_network_level(monkeypatch):
    """The whole network with mopa_wino4_conv forced on for every eligible layer and role (forward with the V by-product for the weight
    gradient, deferred BatchNorm on the way in, backward-data) against the default algorithm choice at this size: logits to 1e-3,
    parameter gradients like two algorithms agree on this small, badly conditioned input (5 % of each tensor's L2 norm; see
    test_source_and_target_batch_in_one_pass_equal_two_calls), running statistics to fp32 round-off."""
    from mopa_amd import dense2d, synth
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    batch = synth.make_batch(4, H=96, W=128)
    batch["bn_groups"] = 2

    def run(direct):
        monkeypatch.setattr(dense2d, "WINO4_DIRECT", direct)
        monkeypatch.setattr(dense2d, "WINO4_DIRECT_ROLES", ("fwd", "fwd_eval", "dgrad"))
        monkeypatch.setattr(dense2d, "WINO4_DIRECT_MIN_TILES", 0)
        torch.manual_seed(5)
        m = build_model_2d(default_cfg())[0].cuda().train()
        m.output_all = True
        o = m(batch)
        g = torch.Generator(device="cuda").manual_seed(3)
        sum((o[k] * torch.randn(o[k].shape, device="cuda", generator=g)).sum() for k in ("seg_logit", "seg_logit2", "seg_logit_all")).backward()
        torch.cuda.synchronize()
        return o["seg_logit"].detach().clone(), [(n, p.grad.clone()) for n, p in m.named_parameters() if p.grad is not None], \
            {k: v.clone() for k, v in m.state_dict().items() if "running" in k}

    calls = []
    inner = dense2d.call
    monkeypatch.setattr(dense2d, "call", lambda name, *a: (calls.append(name), inner(name, *a))[1])
    la, ga, ba = run(False)
    assert calls.count("mopa_wino4_conv") + calls.count("mopa_wino4_conv9") == 0
    lb, gb, bb = run(True)
    # (both forms: mopa_wino4_conv where V is kept for a two-operand weight gradient, mopa_wino4_conv9 elsewhere)
    assert calls.count("mopa_wino4_conv") + calls.count("mopa_wino4_conv9") >= 20, (calls.count("mopa_wino4_conv"), calls.count("mopa_wino4_conv9"))
    if dense2d.WINO4_CONV9:   # (MOPA_WINO4_CONV9=0 = the first form everywhere)
        assert calls.count("mopa_wino4_conv9") >= 8
    _close(lb, la.cpu(), rtol=1e-3, atol=2e-4)
    for k in ba:
        _close(bb[k], ba[k].cpu(), rtol=1e-4, atol=1e-5)
    floor = 1e-3 * max(float(g.norm()) for _, g in ga)
    for (n, a), (_, b) in zip(ga, gb):
        err = float((a - b).norm() / (a.norm() + floor))
        assert err <= 5e-2, (n, err)


@needs_mfma
@pytest.mark.parametrize("groups,training", [(1, True), (2, True), (1, False)])
def test_stem_batchnorm_backward_inside_the_stem_weight_gradient(groups, training, monkeypatch):
    """STEM_BN_FUSED_BWD: the stem BatchNorm's backward is sums + parameter gradients only, the stem's weight gradient forms dx from
    (dy, x, stats, coef) while it stages its tiles (mopa_bn_bwd_sums_groups + mopa_stem_bwd_weight_bn) -- against the apply pass + plain
    weight gradient: every parameter gradient of the network (conv1.weight and bn1 are the ones that could move) to 1e-6 of its scale,
    logits identical; also in eval mode (running statistics, dx = scale * dz) and with two BatchNorm groups."""
    from mopa_amd import dense2d, synth
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d
    batch = synth.make_batch(2 * groups, H=80, W=112)
    if groups > 1:
        batch["bn_groups"] = groups

    def run(fused):
        monkeypatch.setattr(dense2d, "STEM_BN_FUSED_BWD", fused)
        torch.manual_seed(9)
        m = build_model_2d(default_cfg())[0].cuda()
        m = m.train() if training else m.eval()
        o = m(batch)
        g = torch.Generator(device="cuda").manual_seed(1)
        sum((o[k] * torch.randn(o[k].shape, device="cuda", generator=g)).sum() for k in ("seg_logit", "seg_logit2")).backward()
        torch.cuda.synchronize()
        return o["seg_logit"].detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    calls = []
    inner = dense2d.call
    monkeypatch.setattr(dense2d, "call", lambda name, *a: (calls.append(name), inner(name, *a))[1])
    la, ga = run(False)
    assert "mopa_stem_bwd_weight_bn" not in calls
    lb, gb = run(True)
    assert calls.count("mopa_stem_bwd_weight_bn") == 1 and calls.count("mopa_bn_bwd_sums_groups") == 1
    assert torch.equal(la, lb)
    for n in ga:
        scale = float(ga[n].abs().max()) + 1e-20
        err = float((ga[n] - gb[n]).abs().max()) / scale
        assert err <= 1e-6, (n, err)
    assert float(gb["net_2d.conv1.weight"].abs().max()) > 0


@pytest.mark.parametrize("G,one_kernel", [(1, False), (2, False), (2, True), (2, 32)])
def test_join_buffer_with_the_up_convolution_batchnorm_applied_on_the_way_in(G, one_kernel, monkeypatch, request):
    """bn_c0: a decoder join buffer [skip (already relu(batchnorm(.)): non-negative, with exact zeros) | raw up-convolution]; the
    BatchNorm of the second half is statistics only and the consumer normalises channels [c0, C) while it reads -- mopa_wino4_input_bn
    and, in one kernel with the convolution, mopa_wino4_conv -- against the buffer whose second half was written out by the apply pass:
    V identical bits; the convolution's output identical to the same kernel on the materialised buffer."""
    from mopa_amd import dense2d
    from mopa_amd._lib import call, ptr, stream
    from mopa_amd.dense2d import Img, bn_fwd_groups
    if one_kernel:   # (True: k_wino4_conv, 32: k_wino4_conv32)
        call("mopa_wino4_conv_tiles32", int(one_kernel == 32))
        request.addfinalizer(lambda: call("mopa_wino4_conv_tiles32", -1))
    rng = np.random.Generator(np.random.PCG64(77 + G))
    B, H, W, cj = 2 * G, 14, 19, 64
    raw = torch.from_numpy(rng.standard_normal((B * H * W, 2 * cj)).astype(np.float32)).cuda()
    raw[:, :cj] = torch.relu(raw[:, :cj])                     # the skip half
    mat = raw.clone()
    P = {"bn.weight": torch.linspace(0.5, 1.5, cj).cuda(), "bn.bias": torch.linspace(-1, 1, cj).cuda(),
         "bn.running_mean": torch.zeros(cj, device="cuda"), "bn.running_var": torch.ones(cj, device="cuda")}
    stats = torch.empty(G, 4, cj, device="cuda")
    bn_fwd_groups(Img(raw, B, H, W, cj, cj), Img(mat, B, H, W, cj, cj), P, "bn", 1, None, True, stats, G)   # mat = [skip | applied]
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    if not one_kernel:
        Va, Vb = torch.empty(36 * T * 2 * cj, device="cuda"), torch.empty(36 * T * 2 * cj, device="cuda")
        call("mopa_wino4_input", ptr(mat), 2 * cj, B, H, W, 2 * cj, ptr(Va), stream())
        call("mopa_wino4_input_bn", ptr(raw), 2 * cj, B, H, W, 2 * cj, ptr(stats), G, cj, ptr(Vb), stream())
        assert torch.equal(Va, Vb)
        return
    w = torch.from_numpy((rng.standard_normal((64, 2 * cj, 3, 3)) * 0.05).astype(np.float32)).cuda()
    U = torch.empty(36, 2 * cj, 64, device="cuda")
    call("mopa_wino4_weight_f", ptr(w), 64, 2 * cj, 0, ptr(U), stream())
    oa, ob = torch.empty(B * H * W, 64, device="cuda"), torch.empty(B * H * W, 64, device="cuda")
    Va, Vb = torch.empty(36 * T * 2 * cj, device="cuda"), torch.empty(36 * T * 2 * cj, device="cuda")
    call("mopa_wino4_conv", ptr(mat), 2 * cj, ptr(U), None, ptr(oa), 64, B, H, W, 2 * cj, 64, 0, None, 1, 0, ptr(Va), stream())
    call("mopa_wino4_conv", ptr(raw), 2 * cj, ptr(U), None, ptr(ob), 64, B, H, W, 2 * cj, 64, 0, ptr(stats), G, cj, ptr(Vb), stream())
    assert torch.equal(oa, ob) and torch.equal(Va, Vb)
