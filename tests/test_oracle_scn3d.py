"""Anchor the 3D restatement ("parity unpinned" vs sparseconvnet) on dense torch equivalents.

SURVEY.md 8c (i)-(v): SubMConv == conv3d(pad 1) sampled at active sites, Conv k2s2 == conv3d(stride 2),
Deconv k2s2 == conv_transpose3d(stride 2) sampled at the fine active set, BN rows == F.batch_norm,
InputLayer mode 4 == index_add/count, OutputLayer == rows[inverse].
"""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import scn3d


def _cloud(seed, n=400, size=16, batch=2):
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.integers(0, size, (n, 3))
    b = rng.integers(0, batch, (n, 1))
    return np.concatenate([c, b], 1).astype(np.int64)


def test_first_seen_order_and_input_output_layer():
    c = _cloud(0)
    g = scn3d.Geometry(c, num_levels=3, full_scale=16)
    keys = scn3d.pack_keys(c)
    seen = {}
    for i, k in enumerate(keys.tolist()):
        seen.setdefault(k, len(seen))
        assert g.point_row[i] == seen[k]
    feats = torch.randn(c.shape[0], 3, dtype=torch.float64)
    x = scn3d.input_layer(g.point_row, feats, g.num_active[0])
    for r in (0, 5, g.num_active[0] - 1):
        np.testing.assert_allclose(x[r].numpy(), feats[torch.from_numpy(g.point_row == r)].mean(0).numpy())
    y = scn3d.output_layer(g.point_row, x)
    assert torch.equal(y, x[torch.from_numpy(g.point_row.astype(np.int64))])


def test_subm_equals_dense_conv3d():
    c = _cloud(1)
    g = scn3d.Geometry(c, num_levels=2, full_scale=16)
    A = g.num_active[0]
    x = torch.randn(A, 4, dtype=torch.float64)
    W = torch.randn(27, 4, 6, dtype=torch.float64)
    out = scn3d.sparse_conv(x, g.nbr27[0], W)
    dense = scn3d.to_dense(x, g.row_keys[0], 16)
    wd = W.view(3, 3, 3, 4, 6).permute(4, 3, 0, 1, 2)
    ref = scn3d.from_dense(F.conv3d(dense, wd, padding=1), g.row_keys[0])
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-10, atol=1e-10)
    # rule symmetry used by the backward-data kernel: nbr[o][i]=j <=> nbr[26-o][j]=i
    for o in range(27):
        i = np.nonzero(g.nbr27[0][o] >= 0)[0]
        j = g.nbr27[0][o][i]
        assert (g.nbr27[0][26 - o][j] == i).all()


def test_strided_conv_and_deconv_equal_dense():
    c = _cloud(2)
    g = scn3d.Geometry(c, num_levels=2, full_scale=16)
    A0, A1 = g.num_active[:2]
    x = torch.randn(A0, 3, dtype=torch.float64)
    W = torch.randn(8, 3, 5, dtype=torch.float64)
    out = scn3d.sparse_conv(x, g.ch[0], W)
    assert out.shape[0] == A1
    dense = scn3d.to_dense(x, g.row_keys[0], 16)
    wd = W.view(2, 2, 2, 3, 5).permute(4, 3, 0, 1, 2)
    ref = scn3d.from_dense(F.conv3d(dense, wd, stride=2), g.row_keys[1])
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-10, atol=1e-10)
    # every coarse site has >= 1 child and every fine row exactly one rule
    assert ((g.ch[0] >= 0).sum(0) >= 1).all() and ((g.up[0] >= 0).sum(0) == 1).all()
    # deconvolution back onto the cached fine set
    y = torch.randn(A1, 5, dtype=torch.float64)
    Wt = torch.randn(8, 5, 3, dtype=torch.float64)
    up = scn3d.sparse_conv(y, g.up[0], Wt)
    densey = scn3d.to_dense(y, g.row_keys[1], 8)
    wtd = Wt.view(2, 2, 2, 5, 3).permute(3, 4, 0, 1, 2)
    refu = scn3d.from_dense(F.conv_transpose3d(densey, wtd, stride=2), g.row_keys[0])
    np.testing.assert_allclose(up.numpy(), refu.numpy(), rtol=1e-10, atol=1e-10)


def test_bn_relu_matches_torch_batch_norm():
    x = torch.randn(300, 7, dtype=torch.float64) * 2 + 0.5
    w, b = torch.rand(7, dtype=torch.float64) + 0.5, torch.randn(7, dtype=torch.float64)
    rm, rv = torch.zeros(7, dtype=torch.float64), torch.ones(7, dtype=torch.float64)
    rm2, rv2 = rm.clone(), rv.clone()
    y = scn3d.bn_relu(x, w, b, rm, rv, True)
    ref = F.relu(F.batch_norm(x, rm2, rv2, w, b, True, 0.1, 1e-4))
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(rm.numpy(), rm2.numpy(), rtol=1e-12)
    np.testing.assert_allclose(rv.numpy(), rv2.numpy(), rtol=1e-12)
    y = scn3d.bn_relu(x, w, b, rm, rv, False)
    ref = F.relu(F.batch_norm(x, rm2, rv2, w, b, False, 0.1, 1e-4))
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-10, atol=1e-12)


def test_param_count_matches_survey():
    shapes = scn3d.unet_param_shapes()
    conv = sum(int(np.prod(s)) for k, s in shapes.items() if k.endswith(".weight") and len(s) == 3)
    bn_affine = sum(int(np.prod(s)) for k, s in shapes.items() if len(s) == 1 and "running" not in k)
    assert conv == 2_684_848 and bn_affine == 3_808
    assert len([k for k in shapes if k.endswith("running_mean")]) == 26


def test_unet_runs_and_is_deterministic_tiny():
    c = _cloud(3, n=300, size=32)
    g = scn3d.Geometry(c, num_levels=3, full_scale=32)
    from oracle.params import det_state
    P = det_state(scn3d.unet_param_shapes(1, 16, 3))
    f = torch.ones(c.shape[0], 1)
    a = scn3d.unet_forward(P, g, f, num_planes=3, training=False)
    b = scn3d.unet_forward(P, g, f, num_planes=3, training=False)
    assert a.shape == (300, 16) and torch.equal(a, b) and torch.isfinite(a).all()


def test_g7_synthetic_pins(golden_dir):
    from mopa_amd import synth
    pins = json.load(open(os.path.join(golden_dir, "g7_synth_pins.json")))["0"]
    s = synth.make_scan(0)
    c = np.concatenate([s["coords"], np.zeros((len(s["coords"]), 1), np.int64)], 1)
    g = scn3d.Geometry(c)
    assert g.num_active == pins["active"] and g.num_rules == pins["rules"]
    assert c.shape[0] == pins["n_points"] == 34880


from oracle.dense3d import dense_case as _dense_case  # noqa: E402


def _net3d_params(dtype=torch.float64, num_classes=5):
    from oracle.params import det_state
    shapes = scn3d.unet_param_shapes(1, 16, 7, prefix="net_3d.sparseModel.")
    shapes.update({"linear.weight": (num_classes, 16), "linear.bias": (num_classes,),
                   "linear2.weight": (num_classes, 16), "linear2.bias": (num_classes,)})
    P = det_state(shapes, dtype=dtype)
    for k, v in P.items():
        if "running" not in k:
            v.requires_grad_(True)
    return P


def test_unet_equals_dense_network():
    """The whole Net3DSeg (7 levels, the shipped hyper-parameters) of oracle/scn3d.py == the same network executed as DENSE torch
    ops (oracle/dense3d.py: conv3d / conv_transpose3d / masked batch-norm driven by the reference's own recorded layer graph,
    fixture G6) -- logits, BatchNorm batch statistics and every parameter gradient, in float64.  Two unrelated formulations
    (rule tables + canonical rows vs dense grids); this does not pin SparseConvNet, it removes 'one author, one restatement'."""
    from oracle import dense3d
    c, feats = _dense_case()
    g = scn3d.Geometry(c, num_levels=7, full_scale=64)
    assert g.num_active[6] >= 1 and g.num_active[0] < c.shape[0]       # all 7 levels populated, duplicate points present
    up1 = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).standard_normal((c.shape[0], 5)))
    up2 = torch.from_numpy(np.random.Generator(np.random.PCG64(4)).standard_normal((c.shape[0], 5)))

    Pa, Pb = _net3d_params(), _net3d_params()
    oa = scn3d.net3dseg_forward(Pa, g, feats, training=True)
    stats = {}
    ob = dense3d.net3dseg_dense(Pb, c, feats, 64, stats=stats)
    for k in ("feats", "seg_logit", "seg_logit2"):
        np.testing.assert_allclose(oa[k].detach().numpy(), ob[k].detach().numpy(), rtol=1e-9, atol=1e-10)
    ((oa["seg_logit"] * up1).sum() + (oa["seg_logit2"] * up2).sum()).backward()
    ((ob["seg_logit"] * up1).sum() + (ob["seg_logit2"] * up2).sum()).backward()
    worst = 0.0
    for k in Pa:
        if Pa[k].requires_grad:
            ga, gb = Pa[k].grad, Pb[k].grad
            assert ga is not None and gb is not None, k
            scale = float(gb.abs().max()) + 1e-30
            worst = max(worst, float((ga - gb).abs().max()) / scale)
    assert worst < 1e-8, worst
    # running statistics of the sparse oracle == 0.9 * init + 0.1 * (batch mean, UNBIASED batch variance) of the dense network
    P0 = _net3d_params()
    for name, (mean, var, n) in stats.items():
        np.testing.assert_allclose(Pa[name + ".running_mean"].numpy(), (0.9 * P0[name + ".running_mean"] + 0.1 * mean).numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(Pa[name + ".running_var"].numpy(),
                                   (0.9 * P0[name + ".running_var"] + 0.1 * var * n / max(n - 1, 1)).numpy(), rtol=1e-9, atol=1e-12)
    assert len(stats) == 26
