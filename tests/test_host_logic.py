"""Host-side mirror of the reference interface: factory, parameter names, batch layout, error behaviour (CPU only)."""
import numpy as np
import pytest
import torch

from oracle import net2d, scn3d


def test_factory_returns_model_and_metric_with_reference_names():
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d, build_model_3d
    from mopa_amd.models.metric import SegIoU
    cfg = default_cfg(num_classes=5, dual_head=True)
    m2, i2 = build_model_2d(cfg)
    m3, i3 = build_model_3d(cfg)
    assert isinstance(i2, SegIoU) and i2.name == "iou_2d" and i3.name == "iou_3d"
    assert sum(p.numel() for p in m2.parameters()) == 23_614_794
    assert sum(p.numel() for p in m3.parameters()) == 2_688_826
    sd2, sd3 = m2.state_dict(), m3.state_dict()
    for k, shape in net2d.param_shapes(5, True).items():
        assert tuple(sd2[k].shape) == tuple(shape), k
    for k, shape in scn3d.unet_param_shapes().items():   # checkpoints carry SparseConvNet's (K, 1, Cin, Cout)
        got = tuple(sd3["net_3d." + k].shape)
        assert (got[:1] + got[2:] if len(got) == 4 else got) == tuple(shape), k
    assert set(sd2) == set(net2d.param_shapes(5, True))
    # plain dict / attribute configs both work, unsupported backbones raise like the reference (xmuda_arch.py:37,98)
    cfg.MODEL_3D.TYPE = "SPVCNN"
    cfg.MODEL_3D["SPVCNN"] = {}
    with pytest.raises(NotImplementedError):
        build_model_3d(cfg)


def test_no_cpu_fallback():
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d, build_model_3d
    m3, _ = build_model_3d(default_cfg())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m3({"x": [torch.zeros(4, 4, dtype=torch.int64), torch.ones(4, 1)]})
    m2, _ = build_model_2d(default_cfg())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m2({"img": torch.zeros(1, 3, 32, 48), "img_indices": [np.zeros((2, 2), np.int64)]})


def test_pretrained_needs_local_weights(monkeypatch):
    from mopa_amd.models.resnet34_unet import UNetResNet34
    monkeypatch.delenv("MOPA_RESNET34_WEIGHTS", raising=False)
    with pytest.raises(RuntimeError, match="MOPA_RESNET34_WEIGHTS"):
        UNetResNet34(pretrained=True)


def test_pack_indices_layout_and_bounds():
    from mopa_amd.models.xmuda_arch import Net2DSeg
    idx = [np.array([[0, 0], [29, 45], [3, 7]]), torch.tensor([[1, 2]])]
    pix = Net2DSeg.pack_indices(idx, 30, 46, "cpu").numpy()
    Hp, Wp = 32, 48
    assert pix.tolist() == [0, 29 * Wp + 45, 3 * Wp + 7, (Hp + 1) * Wp + 2]
    with pytest.raises(IndexError):
        Net2DSeg.pack_indices([np.array([[30, 0]])], 30, 46, "cpu")


def test_synthetic_batch_layout_matches_collate():
    from mopa_amd import synth
    b = synth.make_batch(2, H=30, W=46)
    locs, feats = b["x"]
    assert locs.dtype == torch.int64 and locs.shape[1] == 4 and feats.shape == (locs.shape[0], 1)
    assert locs[:, 3].unique().tolist() == [0, 1] and int(locs[:, :3].max()) < 4096 and int(locs.min()) == 0
    assert b["img"].shape == (2, 3, 30, 46) and len(b["img_indices"]) == 2 and b["img_indices"][0].shape[1] == 2
    assert b["seg_label"].shape[0] == locs.shape[0] and int(b["seg_label"].min()) == -100
    assert b["sam_mask_ls"][0].dtype == torch.int32 and b["sam_mask_ls"][0].shape == (30, 46)
    # seeds: scan i of rank r uses 1000*r + i
    assert not torch.equal(synth.make_batch(1, rank=1, H=30, W=46)["x"][0], synth.make_batch(1, rank=0, H=30, W=46)["x"][0])


def test_segiou_matches_golden(golden_dir):
    import os
    from mopa_amd.models.metric import SegIoU
    g = dict(np.load(os.path.join(golden_dir, "g5_misc.npz")))
    m = SegIoU(5, name="iou")
    logit, gt = torch.from_numpy(g["logit"]), torch.from_numpy(g["gt"])
    m.update_dict({"seg_logit": logit}, {"seg_label": gt})
    m.update_dict({"seg_logit": logit.flip(0)}, {"seg_label": gt})
    assert (m.mat.numpy() == g["iou_mat"]).all()
    np.testing.assert_allclose(m.iou.numpy(), g["iou"], rtol=1e-6)


def test_voxelizer_matches_golden(golden_dir):
    import os
    from mopa_amd import synth
    g = dict(np.load(os.path.join(golden_dir, "g4_voxelize.npz")))
    # case 0 of G4 is augment_and_scale_3d without augmentation == synth.voxelize
    assert np.array_equal(synth.voxelize(g["points0"], 20), g["coords0"])


def test_scn_checkpoint_weight_layouts_load():
    """state_dict round trip, and SparseConvNet's 4-D (filter_volume, 1, nIn, nOut) weight layout."""
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_3d
    m, _ = build_model_3d(default_cfg())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd4 = {k: (v.unsqueeze(1) if (k.endswith(".weight") and v.dim() == 3) else v) for k, v in sd.items()}
    m2, _ = build_model_3d(default_cfg())
    m2.load_state_dict(sd4)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_oracle_pseudo_labels_pinned_by_golden(golden_dir):
    """oracle/pseudo.py against the reference's own outputs (fixture G5: refine_pseudo_labels, prob_2_entropy)."""
    import os
    from oracle import pseudo
    g = dict(np.load(os.path.join(golden_dir, "g5_misc.npz")))
    prob = torch.softmax(torch.from_numpy(g["logit"]), 1)
    maxp, lab = prob.max(1)
    assert np.array_equal(pseudo.refine_pseudo_labels(maxp, lab).numpy(), g["refined"])
    np.testing.assert_allclose(pseudo.prob_2_entropy(prob).numpy(), g["entropy"], rtol=1e-6, atol=1e-7)
    w = pseudo.fuse_probs(torch.from_numpy(g["logit"]), torch.from_numpy(g["logit"]))
    np.testing.assert_allclose(w.numpy(), prob.numpy(), rtol=1e-6)       # fusing a modality with itself is the identity


def test_conv_algorithm_choice(monkeypatch):
    """dense2d.wino_tile: which algorithm runs a convolution in which pass (the policy behind the measurements in DESIGN.md)."""
    from mopa_amd import dense2d
    monkeypatch.setattr(dense2d, "F4_ROLES", ("dgrad", "wgrad"))
    monkeypatch.setattr(dense2d, "WINOGRAD", True)
    monkeypatch.setattr(dense2d, "WINOGRAD_WGRAD", True)
    B = 8
    t = dense2d.wino_tile
    # forward: F(2x2) for >= 128 channels up to 76x120, direct otherwise; backward passes: F(4x4) up to full resolution
    assert t(256, 256, 3, 1, 1, B, 38, 60, "fwd") == 2 and t(256, 256, 3, 1, 1, B, 38, 60, "dgrad") == 4
    assert t(64, 64, 3, 1, 1, B, 152, 240, "fwd") == 0 and t(64, 64, 3, 1, 1, B, 152, 240, "wgrad") == 4
    assert t(128, 64, 3, 1, 1, B, 304, 480, "fwd") == 0 and t(64, 128, 3, 1, 1, B, 304, 480, "dgrad") == 4
    # not a stride-1 3x3 / ragged channel counts / maps below 8 pixels
    assert t(64, 128, 3, 2, 1, B, 152, 240, "dgrad") == 0 and t(64, 128, 1, 1, 0, B, 76, 120, "fwd") == 0
    assert t(256, 96, 3, 1, 1, B, 38, 60, "dgrad") == 0
    assert t(512, 512, 3, 1, 1, 2, 2, 3, "dgrad") == 2 and t(512, 512, 3, 1, 1, 2, 2, 3, "fwd") == 2
    # forward pass on F(4x4) (the default since round 2): only with enough samples per channel
    monkeypatch.setattr(dense2d, "F4_ROLES", ("fwd", "dgrad", "wgrad"))
    assert t(256, 256, 3, 1, 1, B, 38, 60, "fwd") == 4 and t(256, 256, 3, 1, 1, 2, 8, 12, "fwd") == 2
    # the weight gradient in the transform domain needs 64-aligned channels on both sides
    monkeypatch.setattr(dense2d, "F4_ROLES", ("dgrad", "wgrad"))
    assert dense2d.wino_wgrad_eligible(64, 64, 3, 1, 1, B, 152, 240) and not dense2d.wino_wgrad_eligible(16, 64, 3, 1, 1, B, 152, 240)
    # the fused GEMM + output-transform kernel: 64-aligned input channels and >= 1024 of its 64-tile x 32-channel blocks
    monkeypatch.setattr(dense2d, "WINO4_FUSED_MIN_BLOCKS", 1024)
    assert dense2d.wino4_fused(128, 64, B, 304, 480) and dense2d.wino4_fused(64, 128, B, 304, 480) and dense2d.wino4_fused(64, 128, B, 152, 240)
    assert dense2d.wino4_fused(64, 64, B, 152, 240) and not dense2d.wino4_fused(128, 64, B, 152, 240)   # one K chunk per point: one round suffices
    assert not dense2d.wino4_fused(64, 64, B, 76, 120) and not dense2d.wino4_fused(512, 512, B, 19, 30) and not dense2d.wino4_fused(48, 128, B, 304, 480)
    # the one-kernel F(4x4) convolution (round 4): 64-aligned channels, Cin <= 128, >= 16,384 tiles; by default for backward-data
    # and for a forward pass that keeps nothing, not for the forward pass of a training step (it wants V again)
    monkeypatch.setattr(dense2d, "F4_ROLES", ("fwd", "dgrad", "wgrad"))
    monkeypatch.setattr(dense2d, "WINO4_DIRECT", True)
    monkeypatch.setattr(dense2d, "WINO4_DIRECT_ROLES", ("fwd_eval", "dgrad"))
    monkeypatch.setattr(dense2d, "WINO4_DIRECT_MIN_TILES", 16384)
    d = dense2d.wino4_direct
    assert d(64, 64, 16, 152, 240, "dgrad") and d(64, 128, 16, 304, 480, "dgrad") and d(128, 64, 16, 304, 480, "fwd_eval")
    assert not d(128, 64, 16, 304, 480, "fwd") and not d(128, 128, 16, 76, 120, "dgrad") and not d(256, 256, 16, 152, 240, "dgrad")
    assert not d(64, 64, 4, 152, 240, "dgrad")   # (9,120 tiles)
    # ... in its second form (nine points per wave: fragment layout 3) wherever no V is kept and the output channels come in 64s
    assert dense2d.wino4_layout(64, 64, 16, 152, 240, "dgrad") == 3 and dense2d.wino4_layout(64, 64, 16, 152, 240, "fwd") == 1
    monkeypatch.setattr(dense2d, "WINO4_CONV9", False)
    assert dense2d.wino4_layout(64, 64, 16, 152, 240, "dgrad") == 2
    monkeypatch.setattr(dense2d, "WINO4_CONV9", True)
    # a deferred BatchNorm needs an F(4x4) consumer (and, in training, its transform-domain weight gradient)
    import torch
    op = dense2d.ConvOp(torch.empty(64, 128, 3, 3), None, 3, 1, 1)
    assert op.takes_lazy(16, 304, 480, True) and op.takes_lazy(2, 96, 128, False)
    assert not dense2d.ConvOp(torch.empty(64, 48, 3, 3), None, 3, 1, 1).takes_lazy(16, 304, 480, True)      # ragged input channels
    assert not dense2d.ConvOp(torch.empty(128, 64, 3, 3), None, 3, 2, 1).takes_lazy(16, 152, 240, True)     # strided
    monkeypatch.setattr(dense2d, "WINOGRAD", False)   # (MOPA_WINOGRAD=0 at import)
    assert t(256, 256, 3, 1, 1, B, 38, 60, "dgrad") == 0
    assert not op.takes_lazy(16, 304, 480, True)


def test_parameter_list_cache_follows_module_surgery():
    """The per-forward parameter list is cached (mopa_amd/models/xmuda_arch.py::_FlatCache); replacing a head (another label
    set), moving the module or asking for a refresh must drop it -- the reference's modules are plain nn.Modules where such
    surgery just works."""
    import torch.nn as nn
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d, build_model_3d
    for build in (build_model_2d, build_model_3d):
        m, _ = build(default_cfg(5, True))
        order, flat = m._cache.get(m)
        i = order.index("linear.weight")
        assert flat[i] is m.linear.weight
        m.linear = nn.Linear(m.linear.in_features, 11)
        order2, flat2 = m._cache.get(m)
        assert flat2[order2.index("linear.weight")] is m.linear.weight and flat2[order2.index("linear.weight")].shape[0] == 11
        m.double()
        assert m._cache.flat is None
        m._cache.get(m)
        m.refresh_parameters()
        assert m._cache.flat is None


def test_parameter_cache_notices_replaced_tensor_objects():
    """`load_state_dict(assign=True)` / `sub_module.to(...)` swap tensor OBJECTS inside sub-modules; the cached flat
    list must follow (the kernels would otherwise update orphaned running statistics)."""
    import torch
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_3d
    m, _ = build_model_3d(default_cfg())
    order, flat = m._cache.get(m)
    assert m._cache.get(m)[1] is flat                                    # unchanged objects: the same list
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)                                   # every parameter / buffer object is replaced
    order2, flat2 = m._cache.get(m)
    named = dict(m.named_parameters())
    named.update(dict(m.named_buffers()))
    assert order2 == order and all(t is named[k] for k, t in zip(order2, flat2)) and flat2 is not flat
    bn = next(mod for mod in m.modules() if "running_mean" in mod._buffers)
    bn.double()                                                          # Module._apply on a SUB-module replaces its buffers
    _, flat3 = m._cache.get(m)
    named = dict(m.named_parameters())
    named.update(dict(m.named_buffers()))
    assert all(t is named[k] for k, t in zip(order2, flat3))


def test_paired_batch_point_rows_equal_the_concatenated_calls():
    """bench.py::pair_batch_of builds the point-row ids of the source + target pass as cat([pix_src, pix_trg + B_src * Hp * Wp]):
    the same ids as packing the concatenated img_indices list (what Net2DSeg does with a bn_groups=2 batch given as a dict)."""
    from mopa_amd.models.xmuda_arch import Net2DSeg
    rng = np.random.Generator(np.random.PCG64(3))
    H, W, Bs, Bt = 30, 45, 2, 3
    idx_s = [np.stack([rng.integers(0, H, n), rng.integers(0, W, n)], 1) for n in (7, 0)]
    idx_t = [np.stack([rng.integers(0, H, n), rng.integers(0, W, n)], 1) for n in (5, 9, 1)]
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    ps, pt = Net2DSeg.pack_indices(idx_s, H, W, "cpu"), Net2DSeg.pack_indices(idx_t, H, W, "cpu")
    both = Net2DSeg.pack_indices(idx_s + idx_t, H, W, "cpu")
    assert torch.equal(both, torch.cat([ps, pt + Bs * Hp * Wp]))
    assert both.dtype == torch.int32 and int(both.max()) < (Bs + Bt) * Hp * Wp


def test_merge_domains_helpers():
    """mopa_amd.step.merge_domains_2d / _3d build the one-pass batches of bench.py's paired step: point rows of the target shifted
    behind the source's images, scan indices of later batches behind the earlier ones', group boundaries in points."""
    from mopa_amd.models.xmuda_arch import Net2DSeg
    from mopa_amd.step import merge_domains_2d, merge_domains_3d
    rng = np.random.Generator(np.random.PCG64(4))
    H, W, B = 30, 45, 2
    idx_s = [np.stack([rng.integers(0, H, n), rng.integers(0, W, n)], 1) for n in (7, 3)]
    idx_t = [np.stack([rng.integers(0, H, n), rng.integers(0, W, n)], 1) for n in (5, 9)]
    img_s, img_t = torch.zeros(B, 3, H, W), torch.ones(B, 3, H, W)
    m2 = merge_domains_2d(img_s, img_t, Net2DSeg.pack_indices(idx_s, H, W, "cpu"), Net2DSeg.pack_indices(idx_t, H, W, "cpu"))
    assert m2["bn_groups"] == 2 and m2["img"].shape[0] == 2 * B and float(m2["img"][B:].min()) == 1.0
    assert torch.equal(m2["point_pix_2d"], Net2DSeg.pack_indices(idx_s + idx_t, H, W, "cpu"))
    with pytest.raises(ValueError):
        merge_domains_2d(img_s, img_t[:1], m2["point_pix_2d"], m2["point_pix_2d"])

    def cloud(n, nb):
        c = torch.from_numpy(rng.integers(0, 50, (n, 4)))
        c[:, 3] = torch.sort(torch.from_numpy(rng.integers(0, nb, n)))[0]
        return c, torch.rand(n, 1)

    a, b, c = cloud(11, 2), cloud(7, 2), cloud(5, 3)
    m = merge_domains_3d([a, b], [2, 2])
    assert m["bn_group_points"] == 11 and m["x"][0].shape == (18, 4) and m["x"][1].shape == (18, 1)
    assert torch.equal(m["x"][0][:11], a[0]) and torch.equal(m["x"][0][11:, :3], b[0][:, :3]) and torch.equal(m["x"][0][11:, 3], b[0][:, 3] + 2)
    assert torch.equal(b[0][:, 3], torch.sort(b[0][:, 3])[0]) and int(b[0][:, 3].max()) <= 1   # the inputs are not modified
    m3 = merge_domains_3d([a, b, c], [2, 2, 3])
    assert m3["bn_group_points"] == [11, 18] and int(m3["x"][0][18:, 3].min()) >= 4
    with pytest.raises(ValueError):
        merge_domains_3d([a], [2])


def test_dispatcher_follows_the_measured_algorithm_table():
    """profiles/r6_algo_table.json (profiles/algo_table.py on an MI355X: every stride-1 3x3 layer shape at 4 / 8 / 16 images and at
    225 x 400 / 302 x 480, each algorithm timed) records what the dispatcher picked per role when it was taken.  The thresholds in
    mopa_amd/dense2d.py (WINO4_DIRECT_MIN_TILES, the nine-point form's fill rule, WINO4_WGRAD_FUSED_MIN_TILES, WINO4_FUSED_MIN_BLOCKS,
    WINOGRAD_F4_PIXELS) are decisions read off that table: changing them without re-measuring fails here.  Also: where nothing is kept
    (fwd_eval / backward-data) the pick is within 10 % of the fastest measured alternative on every row -- except the rows the table
    itself flags (listed below with the reason)."""
    import json
    import os
    from mopa_amd import dense2d
    rows = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r6_algo_table.json")))
    assert len(rows) >= 48

    def choice(cin, cout, B, H, W, role):
        F = dense2d.wino_tile(cin, cout, 3, 1, 1, B, H, W, "fwd" if role == "fwd_eval" else role)
        if F != 4:
            return {0: "direct", 2: "F2"}[F]
        return {3: "F4 one9", 2: "F4 one", 1: "F4 fused", 0: "F4"}[dense2d.wino4_layout(cin, cout, B, H, W, role)]

    for r in rows:
        for role, picked in r["chosen"].items():
            if picked in ("-", "(next row)"):
                continue
            # the training forward pass is dispatched as "fwd_eval" where the weight gradient needs no V (dense2d.forward_role)
            drole = dense2d.forward_role(r["cin"], r["cout"], 3, 1, 1, r["B"], r["H"], r["W"], True)[1] if role == "fwd" else role
            assert choice(r["cin"], r["cout"], r["B"], r["H"], r["W"], drole) == picked, (r["res"], r["B"], r["layer"], role)
        for role in ("fwd_eval", "dgrad"):
            ratio = r["chosen_over_best"].get(role)
            if ratio is None:
                continue
            # F(2x2) on layer4 below 4,096 samples per channel is an ACCURACY rule (dense2d.F4_FWD_MIN_PIXELS: gradient noise of F(4x4)
            # where BatchNorm normalises over few samples), not a speed choice
            slack = 1.30 if r["chosen"][role] == "F2" else 1.10
            assert ratio <= slack, (r["res"], r["B"], r["layer"], role, ratio)


def test_weight_gradient_dispatcher_follows_the_measured_table():
    """mopa_spconv_wgrad_run_wanted (csrc/sprun.hip): the rule read off profiles/r5_wgrad_run.md -- every (table, rows, Cin, Cout) of the
    8- and 16-scan tables where the run-list kernel measured faster than the dense-table one by 5 % or more is sent to it, the ones
    where it measured slower are not.  (A size query: no GPU needed.)"""
    from mopa_amd._lib import query
    want = lambda *a: query("mopa_spconv_wgrad_run_wanted", *a)
    # 27-offset tables: (rows, cin, cout) -> picked
    for rows, cin, cout, pick in ((257465, 16, 16, 0), (257465, 32, 16, 0), (193135, 32, 32, 0), (193135, 64, 32, 1), (103554, 48, 48, 1),
                                  (103554, 96, 48, 1), (49022, 64, 64, 1), (49022, 128, 64, 1), (19312, 80, 80, 0), (39360, 80, 80, 1),
                                  (19312, 160, 80, 1), (6789, 96, 96, 1), (6789, 192, 96, 1), (2330, 112, 112, 1), (2330, 224, 112, 0),
                                  (4782, 224, 112, 0), (99391, 128, 64, 1), (207867, 96, 48, 1)):
        assert want(27, rows, cin, cout, 0) == pick, (rows, cin, cout)
    # 8-offset tables (rows = fine rows of the deconvolution table; the stride-2 convolution's gradient runs on the same lists)
    for rows, cin, cout, pick in ((257465, 32, 16, 0), (257465, 16, 32, 0), (193135, 48, 32, 1), (193135, 32, 48, 1), (103554, 64, 48, 1),
                                  (49022, 80, 64, 1), (49022, 64, 80, 1), (19312, 96, 80, 0), (39360, 96, 80, 0), (6789, 112, 96, 0),
                                  (515277, 32, 16, 0), (388842, 48, 32, 1)):
        assert want(8, rows, cin, cout, 1) == pick, (rows, cin, cout)
    assert want(27, 49022, 20, 64, 0) == 0 and want(27, 49022, 64, 120, 0) == 0 and want(27, 49022, 272, 64, 0) == 0   # unsupported shapes
    assert query("mopa_spconv_wgrad_run_workspace_bytes", 27, 49022, 128, 64, 0) >= 27 * 128 * 64 * 4
