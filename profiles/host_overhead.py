#!/usr/bin/env python3
"""Host (Python + ctypes + autograd) time to enqueue one Net2DSeg / Net3DSeg forward+backward vs its GPU time."""
import sys, time
import torch
sys.path.insert(0, ".")
from mopa_amd import synth
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d, build_model_3d
from mopa_amd.optim import FlatAdam
from mopa_amd.sparse3d import Geometry3D

cfg = default_cfg(num_classes=5, dual_head=True)
b = synth.make_batch(8)
dev = "cuda"
for name, build in (("2D", build_model_2d), ("3D", build_model_3d)):
    m = build(cfg)[0].to(dev).train()
    opt = FlatAdam(m.parameters())
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    if name == "3D":
        geom3d = batch["geometry_3d"] = Geometry3D(b["x"][0], 7, 4096, dev)   # geometry (2 host syncs) kept out of this probe
    for it in range(6):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m(batch)
        (out["seg_logit"].square().mean() + out["seg_logit2"].square().mean()).backward()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {(t1 - t0) / 4 * 1e3:.1f} ms per fwd+bwd, GPU-complete {(t2 - t0) / 4 * 1e3:.1f} ms")

# ---- pure host cost: the same passes with every library call replaced by a no-op (nothing reaches the GPU queue)
import mopa_amd._lib as L, mopa_amd.sparse3d as S3, mopa_amd.dense2d as D2, mopa_amd.common.utils.loss as LS
noop = lambda name, *a: 0
for mod in (L, S3, D2, LS):
    if hasattr(mod, "call"):
        mod.call = noop
for name, build in (("2D", build_model_2d), ("3D", build_model_3d)):
    m = build(cfg)[0].to(dev).train()
    opt = FlatAdam(m.parameters())
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    if name == "3D":
        batch["geometry_3d"] = geom3d
    torch.cuda.synchronize()
    for it in range(6):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m(batch)
        (out["seg_logit"].square().mean() + out["seg_logit2"].square().mean()).backward()
    t1 = time.perf_counter()
    print(f"{name}: pure host time {(t1 - t0) / 4 * 1e3:.1f} ms per fwd+bwd (library calls stubbed)")
