"""Structure pin of the 3D branch (fixture G6, CPU only).

tests/golden/g6_scn_structure.json was produced by importing the reference's ``UNetSCN``, ``UNetSCN_ED`` and ``Net3DSeg``
(mopa/models/scn_unet.py:9-34,38-219, mopa/models/xmuda_arch.py:82-126) under the recording ``sparseconvnet`` stand-in
(oracle/scn_recorder.py, generator oracle/gen_golden.py::gen_g6).  What is pinned here: layer list, constructor
arguments, channel counts, execution order and JoinTable operand order of the reference's own wiring, and the
state_dict names that follow from it.  NOT pinned (source absent): SparseConvNet's arithmetic -- tests/test_oracle_scn3d.py.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import scn3d


@pytest.fixture(scope="module")
def g6(golden_dir):
    with open(os.path.join(golden_dir, "g6_scn_structure.json")) as f:
        return json.load(f)


def _fold(shape):
    """SCN stores conv weights as (volume, nIn, nOut) or (volume, 1, nIn, nOut): compare on the folded 3-D form."""
    s = list(shape)
    return s[:1] + s[2:] if len(s) == 4 and s[1] == 1 else s


def test_reference_constructor_arguments(g6):
    """What scn_unet.py:25-30 passes to sparseconvnet, as logged by the recording stand-in."""
    calls = g6["UNetSCN"]["ctor_calls"]
    assert calls[0] == {"type": "InputLayer", "dimension": 3, "spatial_size": 4096, "mode": 4}
    assert calls[1]["type"] == "SubmanifoldConvolution" and (calls[1]["nIn"], calls[1]["nOut"], calls[1]["filter_size"],
                                                             calls[1]["bias"]) == (1, 16, 3, False)
    unet = calls[2]
    assert unet["type"] == "UNet" and unet["reps"] == 1 and unet["nPlanes"] == [16, 32, 48, 64, 80, 96, 112]
    assert unet["residual_blocks"] is False and unet["downsample"] == [2, 2] and unet["leakiness"] == 0
    assert calls[-2]["type"] == "BatchNormReLU" and calls[-2]["nPlanes"] == 16 and calls[-2]["eps"] == 1e-4
    assert calls[-1]["type"] == "OutputLayer"
    # every convolution of the network is bias-free, every BN has SCN's eps / momentum, leakiness 0
    for c in calls:
        if "Convolution" in c["type"] or c["type"] == "Deconvolution":
            assert c["bias"] is False
            assert c["filter_volume"] == (27 if c["type"] == "SubmanifoldConvolution" else 8)
            if c["type"] != "SubmanifoldConvolution":
                assert (c["filter_size"], c["filter_stride"]) == (2, 2)
        if c["type"].startswith("BatchNorm"):
            assert (c["eps"], c["momentum"], c["leakiness"]) == (1e-4, 0.9, 0)


def test_unrolled_reference_network_equals_scn_unet(g6):
    """UNetSCN_ED (the reference's own unrolled restatement, scn_unet.py:38-135) runs the same layer sequence."""
    assert g6["UNetSCN_equals_UNetSCN_ED"] is True
    seq = g6["layer_sequence"]
    assert len(seq) == 60
    joins = [t for t in seq if t[0] == "JoinTable"]
    assert len(joins) == 6
    for t in joins:   # [skip | up]: the encoder feature first, the deconvolution output second (scn_unet.py:108-124)
        assert t[1][0] == t[1][1] and t[2] == ["SubmanifoldConvolution", "Deconvolution"]
    assert [t[3] for t in joins] == [5, 4, 3, 2, 1, 0]


VARIANTS = [("UNetSCN", {}), ("UNetSCN_reps2", {"block_reps": 2}), ("UNetSCN_m32_planes5", {"m": 32, "num_planes": 5}),
            ("UNetSCN_residual", {"residual_blocks": True}), ("UNetSCN_residual_reps2", {"residual_blocks": True, "block_reps": 2})]


@pytest.mark.parametrize("variant,kw", VARIANTS)
def test_oracle_parameter_names_and_shapes(g6, variant, kw):
    want = [[k, _fold(v)] for k, v in g6[variant]["state_dict"]]
    got = [[k, list(v)] for k, v in scn3d.unet_param_shapes(**kw).items()]
    assert got == want   # names, shapes AND module order


@pytest.mark.parametrize("variant,kw", VARIANTS)
def test_product_module_tree_and_layer_program(g6, variant, kw):
    """The product's UNetSCN (state_dict names / shapes / order) and the layer program its forward EXECUTES
    (mopa_amd/sparse3d.py::Program) against the reference's constructors run under the recording stand-in."""
    from mopa_amd.models.scn_unet import UNetSCN
    from mopa_amd.sparse3d import Program
    net = UNetSCN(1, **kw)
    assert [[k, _fold(v.shape)] for k, v in net.state_dict().items()] == [[k, _fold(v)] for k, v in g6[variant]["state_dict"]]
    prog = Program(1, kw.get("m", 16), kw.get("num_planes", 7), kw.get("block_reps", 1), kw.get("residual_blocks", False), "sparseModel.")
    want = []
    for t in g6[variant]["trace"]:
        op = {"BatchNormLeakyReLU": "BatchNormReLU"}.get(t["op"], t["op"])
        if op == "JoinTable":
            continue      # free in the product: both producers write halves of one buffer (checked below)
        want.append([op, t["cout"], t["level_out"]] if op == "AddTable" else [op, t["cin"], t["cout"], t["level_in"], t["level_out"]])
    got = prog.layer_sequence()
    if kw.get("residual_blocks"):
        # ResNet blocks: the shortcut's NetworkInNetwork runs after the residual branch here (so that its backward is the first
        # writer of the block input's gradient); the two branches are independent, every other position is identical
        assert sorted(map(str, got)) == sorted(map(str, want))
        assert [t for t in got if t[0] != "NetworkInNetwork"] == [t for t in want if t[0] != "NetworkInNetwork"]
    else:
        assert got == want
    # every parameter-carrying op names a tensor of the reference's state_dict, with matching channel counts
    sd = {k: _fold(v) for k, v in g6[variant]["state_dict"]}
    for op in prog.ops:
        if op[0] == "bn":
            assert sd[op[1] + ".weight"] == [op[2].C] and op[2].C == op[3].C
        elif op[0] == "conv":
            w = sd[op[1] + ".weight"]
            assert w[-2:] == [op[4].C, op[5].C] and (len(w) == 2) == (op[2] == "nin")
    # JoinTable([skip, up]) (scn_unet.py:108-124): the skip producer writes columns [0, P), the deconvolution [P, 2P) of one buffer
    ups = [op for op in prog.ops if op[0] == "conv" and op[2] == "up"]
    for up in ups:
        dst = up[5]
        skip = [op for op in prog.ops if op[-1].buf == dst.buf and op[-1].col == 0 and op[-1] is not dst]
        assert dst.col == dst.C and len(skip) == 1 and skip[0][-1].C == dst.C


@pytest.mark.parametrize("variant,kw", VARIANTS[:2] + VARIANTS[3:])
def test_oracle_executes_the_reference_layer_sequence(g6, variant, kw):
    reps = kw.get("block_reps", 1)
    rng = np.random.Generator(np.random.PCG64(3))
    c = np.concatenate([rng.integers(0, 150, (300, 3)), rng.integers(0, 2, (300, 1))], 1).astype(np.int64)
    geom = scn3d.Geometry(c)
    res = kw.get("residual_blocks", False)
    shapes = scn3d.unet_param_shapes(block_reps=reps, residual_blocks=res)
    P = {"sparseModel." + k[len("sparseModel."):]: (torch.ones(s) if "running_var" in k else torch.zeros(s)) for k, s in shapes.items()}
    trace = []
    scn3d.unet_forward(P, geom, torch.ones(300, 1), block_reps=reps, training=False, trace=trace, residual_blocks=res)

    def arith(tr):
        out = []
        for t in tr:
            op = {"BatchNormLeakyReLU": "BatchNormReLU"}.get(t["op"], t["op"])
            if op == "JoinTable":
                out.append([op, t["parts"], t["part_ops"], t["level_out"]])
            elif op == "AddTable":
                out.append([op, t["cout"], t["level_out"]])
            else:
                out.append([op, t["cin"], t["cout"], t["level_in"], t["level_out"]])
        return out

    assert trace == arith(g6[variant]["trace"])


@pytest.mark.parametrize("dual,key", [(True, "Net3DSeg_dual"), (False, "Net3DSeg_single"), (True, "Net3DSeg_MCD")])
def test_product_state_dict_matches_reference_names(g6, dual, key):
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_3d
    cfg = default_cfg(num_classes=5, dual_head=dual)
    model, _ = build_model_3d(cfg)
    if key.endswith("MCD"):   # da_method='MCD': a third head the reference creates and never calls (xmuda_arch.py:110-126)
        from mopa_amd.models.xmuda_arch import Net3DSeg
        model = Net3DSeg(5, True, "SCN", dict(cfg.MODEL_3D.SCN), da_method="MCD")
    got = [[k, _fold(v.shape)] for k, v in model.state_dict().items()]   # (K,1,Cin,Cout) in checkpoints == (K,Cin,Cout)
    want = [[k, _fold(v)] for k, v in g6[key]["state_dict"]]
    # names, shapes and traversal order: torch optimizers / torch_ema index their state by parameter order
    assert got == want
    assert [k for k, _ in model.named_parameters()] == [k for k, _ in want if "running" not in k]


def test_native_executor_tables_follow_the_program():
    """The host tables the native executor walks (csrc/scn_exec.hip: one C-ABI call per pass) are a faithful encoding of the layer
    program that G6 pins: same ops in the same order, every parameter named once, the backward plan covers every op once in
    reverse, BatchNorm statistics slots do not overlap."""
    from mopa_amd.sparse3d import Program
    for reps, residual in ((1, False), (2, False), (1, True)):
        prog = Program(1, 16, 7, reps, residual, "net_3d.sparseModel.")
        nt = prog.native_tables()
        assert nt is prog.native_tables()                      # built once
        t, names = nt["prog"], nt["names"]
        assert len(t) == len(prog.ops) == len(names)
        kinds = {"bn": 0, "conv": 1, "add": 2}
        ck = {"subm": 0, "down": 1, "up": 2, "nin": 3}
        slots = []
        for row, op, name in zip(t, prog.ops, names):
            assert row[0] == kinds[op[0]]
            if op[0] == "bn":
                assert name == op[1] and (row[4], row[5], row[6]) == (op[2].buf, op[2].col, op[2].C) and row[9] == op[3].C
                slots.append((int(row[10]), int(row[10]) + 4 * op[2].C))
            elif op[0] == "conv":
                assert name == op[1] and row[1] == ck[op[2]] and (row[2], row[3]) == (op[4].level, op[5].level)
                assert (row[6], row[9]) == (op[4].C, op[5].C)
            else:
                assert name is None and (row[10], row[11]) == (op[2].buf, op[2].col)
        slots.sort()
        assert all(a[1] <= b[0] for a, b in zip(slots, slots[1:])) and slots[-1][1] == nt["stats_floats"]
        named = [n for n in names if n is not None]
        assert len(set(named)) == len(named)
        plan = nt["plan"]
        assert sorted(int(r[1]) for r in plan) == [i for i, n in enumerate(names) if n is not None]   # every bn / conv exactly once
        order = [int(r[1]) for r in plan]
        assert order == sorted(order, reverse=True)            # ... in reverse program order
        assert plan[nt["stem_step"], 0] == 1 and names[plan[nt["stem_step"], 1]].endswith("sparseModel.1")


def test_unetscn_width_limits_are_stated_not_discovered_at_a_kernel_launch():
    """The reference accepts any m (mopa/models/scn_unet.py:11,23); the product constructs every m % 4 == 0 (names and shapes as
    pinned above) and says at the first forward -- not in a kernel's error code -- when the HIP kernels cannot run the width."""
    from mopa_amd.models.scn_unet import UNetSCN
    with pytest.raises(NotImplementedError):
        UNetSCN(1, m=6)
    for ok in (dict(m=4), dict(m=8), dict(m=12), dict(m=16), dict(m=32, num_planes=3), dict(m=16, in_channels=4)):
        assert UNetSCN(ok.pop("in_channels", 1), **ok).not_runnable is None
    for bad in (dict(m=20), dict(m=32), dict(m=32, num_planes=5), dict(m=64, num_planes=2)):
        net = UNetSCN(1, **bad)
        assert "sparse-conv kernels need" in net.not_runnable
        with pytest.raises(NotImplementedError):
            net.geometry(None)
