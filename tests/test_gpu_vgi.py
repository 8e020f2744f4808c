"""VGI on the device (mopa_amd/vgi.py, csrc/vgi.hip) against fixture G8 = the reference's own outputs on two full-size
synthetic scans (oracle/gen_golden.py::gen_g8), and against the oracle.  Integer outputs bit-exact; the placed object's
coordinates within 1e-5 m (the road height is a float32 mean whose summation order differs)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g8(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "g8_vgi.npz")))


def _case(k):
    from oracle.gen_golden import vgi_case
    return vgi_case(k)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_overlap_test_is_bit_exact_with_the_reference(g8, k):
    from mopa_amd import vgi
    c = _case(k)
    m = vgi.OverlapMap(c["ori_pc"], 0.5, (25.0, 25.0), -2.0, c["front"])
    fc = m.free_cells(c["objs"][0][:, :3])
    free = fc["free"].cpu().numpy().astype(bool)
    assert tuple(free.shape) == tuple(g8[f"free_shape{k}"])
    assert np.array_equal(np.packbits(free.reshape(-1)), g8[f"free_bits{k}"])
    vc = vgi.check_overlap(c["ori_pc"], c["objs"][0][:, :3], 0.5, (25.0, 25.0), -2.0, None, c["front"])
    assert len(vc) == int(g8[f"n_centers{k}"]) and np.array_equal(vc[:64], g8[f"centers_head{k}"])
    np.testing.assert_allclose(vc.sum(0), g8[f"centers_sum{k}"], rtol=1e-12)
    assert int(m.status.item()) == 0


@pytest.mark.parametrize("k", [0, 1, 2])
def test_ground_insertion_matches_the_reference(g8, k):
    from mopa_amd import vgi
    from oracle import vgi as ovgi
    c = _case(k)
    # the candidate ground cells are integers: exactly the oracle's set, in its (lexicographic) order
    m = vgi.OverlapMap(c["ori_pc"], 0.5, (25.0, 25.0), -2.0, c["front"], g_mask=c["g_mask"])
    anchor = c["objs"][0]
    cells, n_free, n_kept = m.ground_cells(m.free_cells(anchor[:, :3]), anchor, c["proj"], c["image_size"])
    vc = ovgi.filter_centers(ovgi.check_overlap(c["ori_pc"], anchor[:, :3], 0.5, (25.0, 25.0), -2.0, None, c["front"]), anchor,
                             c["proj"], c["image_size"])
    want, _, _, _ = ovgi.ground_centers(c["ori_pc"], vc, c["g_mask"], 0.5)
    assert n_free == int(g8[f"n_centers{k}"]) and n_kept == len(vc)
    assert np.array_equal(cells, want.astype(np.int64))
    # the whole insertion with the reference's seed: same cells picked, objects land where the reference put them
    np.random.seed(100 + k)
    cat_pc, cat_label, mask, _ = vgi.point_mixmatch(torch.from_numpy(c["ori_pc"]).cuda(), c["label"], [o.copy() for o in c["objs"]],
                                                    c["obj_labels"], insert_mode="ground", search_voxel_size=0.5, search_range=[25.0, 25.0],
                                                    search_z_min=-2.0, proj_matrix=c["proj"], image_size=c["image_size"],
                                                    g_indices=c["g_mask"], front_axis=c["front"])
    n0 = len(c["ori_pc"])
    assert cat_pc.dtype == torch.float64 and int(mask.sum()) == len(g8[f"obj_xyz{k}"]) and not bool(mask[:n0].any())
    np.testing.assert_allclose(cat_pc[n0:].cpu().numpy(), g8[f"obj_xyz{k}"], rtol=0, atol=1e-5)
    assert np.array_equal(cat_label[n0:].cpu().numpy(), g8[f"cat_label_tail{k}"])
    assert torch.equal(cat_pc[:n0].cpu(), torch.from_numpy(c["ori_pc"][:, :3]).double())


def test_road_height_is_bit_reproducible():
    """The road height under a chosen cell (mixmatch_ss.py:431-444) is an ordered sum (one block, fixed wave order:
    csrc/vgi.hip::k_vgi_road; it was two global double atomics per wave): the whole insertion gives the same bits run to run."""
    from mopa_amd import vgi
    c = _case(0)
    m = vgi.OverlapMap(c["ori_pc"], 0.5, (25.0, 25.0), -2.0, c["front"], g_mask=c["g_mask"])
    anchor = c["objs"][0]
    cells, _, _ = m.ground_cells(m.free_cells(anchor[:, :3]), anchor, c["proj"], c["image_size"])
    for cell in cells[:: max(1, len(cells) // 8)][:8]:
        h = [m.road_height(cell) for _ in range(4)]
        assert all(np.array_equal(np.asarray(x), np.asarray(h[0])) for x in h[1:])
    outs = []
    for _ in range(2):
        np.random.seed(100)
        cat_pc, _, _, _ = vgi.point_mixmatch(torch.from_numpy(c["ori_pc"]).cuda(), c["label"], [o.copy() for o in c["objs"]],
                                             c["obj_labels"], insert_mode="ground", search_voxel_size=0.5, search_range=[25.0, 25.0],
                                             search_z_min=-2.0, proj_matrix=c["proj"], image_size=c["image_size"],
                                             g_indices=c["g_mask"], front_axis=c["front"])
        outs.append(cat_pc)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("k", [0, 1, 2])
def test_range_image_culling_and_post_process_bit_exact(g8, k):
    from mopa_amd import vgi
    c = _case(k)
    n0 = len(c["ori_pc"])
    cat = np.concatenate([c["ori_pc"][:, :3].astype(np.float64), g8[f"obj_xyz{k}"]], 0)   # the reference's own cloud
    cat_dev = torch.from_numpy(cat).cuda()
    keep = vgi.range_keep(cat_dev, n0, 0.05235, -0.43633, 1024, 64)
    assert len(keep) == int(g8[f"n_cat{k}"]) and np.array_equal(np.packbits(keep.cpu().numpy()), g8[f"pres_bits{k}"])
    mask = torch.zeros(len(cat), dtype=torch.bool, device="cuda")
    mask[n0:] = True
    label = torch.from_numpy(np.concatenate([c["label"], g8[f"cat_label_tail{k}"]])).cuda()
    np.random.seed(200 + k)
    aug = {"noisy_rot": 0.1, "flip_y": 0.5, "rot_z": 6.2831, "transl": True}
    cat_input, ps, om, _ = vgi.post_process([cat_dev], [label], [mask], 20, 4096, aug, use_proj=True, backbone="SCN")
    locs = cat_input["x"][0].cpu().numpy()
    assert len(locs) == int(g8[f"locs_n{k}"]) and np.array_equal(locs[:128], g8[f"locs_head{k}"])
    key = (locs[:, 0] << 24) | (locs[:, 1] << 12) | locs[:, 2]
    assert int(key.sum()) == int(g8[f"locs_keysum{k}"]) and int(np.bitwise_xor.reduce(key)) == int(g8[f"locs_keyxor{k}"])
    assert int(ps.sum()) == int(g8[f"ps_sum{k}"]) and int(om.sum()) == int(g8[f"om_sum{k}"])
    assert cat_input["x"][1].shape == (len(locs), 1) and locs[:, 3].max() == 0
    # the re-voxelised cloud feeds the 3D branch directly (third pass of a MoPA iteration)
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_3d
    m3 = build_model_3d(default_cfg())[0].cuda().eval()
    with torch.no_grad():
        out = m3(cat_input)
    assert out["seg_logit"].shape == (len(locs), 5) and torch.isfinite(out["seg_logit"]).all()


def test_no_valid_placement_returns_the_scan_unchanged():
    from mopa_amd import vgi
    c = _case(0)
    huge = [np.concatenate([(np.random.default_rng(0).random((300, 3)) - 0.5) * 120.0, np.zeros((300, 1))], 1).astype(np.float32)]
    cat_pc, lab, mask, _ = vgi.point_mixmatch(c["ori_pc"], c["label"], huge, [np.full(300, 2)], insert_mode="ground", search_voxel_size=0.5,
                                              search_range=[25.0, 25.0], search_z_min=-2.0, proj_matrix=c["proj"],
                                              image_size=c["image_size"], g_indices=c["g_mask"], front_axis=c["front"])
    assert cat_pc.shape[0] == len(c["ori_pc"]) and not bool(mask.any())


@pytest.mark.parametrize("k", [0, 1])
def test_front_view_insertion_matches_the_reference(golden_dir, k):
    """insert_mode="fv" (mixmatch_ss.py:83-105) against fixture G8b: object coordinates bit-exact (float32 values in the float64
    device cloud), labels, masks, and the reference's in-place edit of the caller's object arrays."""
    import os
    from mopa_amd import vgi
    from oracle.gen_golden import vgi_fv_objects
    g = dict(np.load(os.path.join(golden_dir, "g8b_vgi_fv.npz")))
    c = _case(k)
    objs = vgi_fv_objects(k)
    cat_pc, cat_label, mask, ps_mask = vgi.point_mixmatch(torch.from_numpy(c["ori_pc"]).cuda(), c["label"], objs, c["obj_labels"],
                                                          z_disc=-0.324, insert_mode="fv")
    n0 = len(c["ori_pc"])
    assert cat_pc.dtype == torch.float64 and cat_pc.is_cuda and int(mask.sum()) == len(g[f"obj_xyz{k}"]) and not bool(mask[:n0].any())
    assert np.array_equal(cat_pc[n0:].cpu().numpy(), g[f"obj_xyz{k}"].astype(np.float64))
    assert np.array_equal(cat_label[n0:].cpu().numpy(), g[f"cat_label_tail{k}"]) and torch.equal(mask, ps_mask)
    assert torch.equal(cat_pc[:n0].cpu(), torch.from_numpy(c["ori_pc"][:, :3]).double())
    assert np.array_equal(objs[0], g[f"objs_after{k}_0"]) and np.array_equal(objs[1], g[f"objs_after{k}_1"])
    with pytest.raises(ValueError):
        vgi.point_mixmatch(c["ori_pc"], c["label"], objs, c["obj_labels"], insert_mode="sideways")


def test_batched_insertion_equals_the_per_scan_loop():
    """`point_mixmatch_batch` (two host round trips per batch) == the reference's per-scan loop (`train_xmuda_mopa.py:516-555`) under
    the same numpy seed: same RNG draws in the same order, so the same cells, the same placements, bit for bit -- including a scan
    whose first anchor fits nowhere (case 2: it takes the sequential path at its turn, between the fast-path scans)."""
    from mopa_amd import vgi
    cases = [_case(k) for k in (0, 2, 4)]                      # all with front axis y (one projection per batch, like a data set)
    assert {c["front"] for c in cases} == {"y"}
    kw = dict(search_voxel_size=0.5, search_range=[25.0, 25.0], search_z_min=-2.0, proj_matrix=cases[0]["proj"],
              image_size=cases[0]["image_size"], front_axis="y")

    def items():
        return [dict(ori_pc=torch.from_numpy(c["ori_pc"]).cuda(), ori_label=c["label"], obj_pc_ls=[o.copy() for o in c["objs"]],
                     obj_label_ls=c["obj_labels"], g_indices=c["g_mask"]) for c in cases]

    np.random.seed(77)
    loop = [vgi.point_mixmatch(it["ori_pc"], it["ori_label"], it["obj_pc_ls"], it["obj_label_ls"], insert_mode="ground",
                               g_indices=it["g_indices"], **kw) for it in items()]
    after_loop = np.random.rand()
    np.random.seed(77)
    batch = vgi.point_mixmatch_batch(items(), **kw)
    assert np.random.rand() == after_loop                       # the global RNG ends in the same state
    assert len(batch) == len(loop) == 3
    for a, b in zip(loop, batch):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        assert a[2].n_obj == b[2].n_obj and int(b[2].sum()) > 0
