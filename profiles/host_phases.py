#!/usr/bin/env python3
"""Host time per phase of the 3D-only training step (geometry / forward / losses / backward / optimizer), device idle between
phases (a sync after each, so the numbers are pure enqueue cost + kernel time of that phase).  Run from a tree's root:
    python profiles/host_phases.py            (works on the round-1 tree too: A/B of host-side changes)"""
import sys, time
import torch
sys.path.insert(0, ".")
from mopa_amd import synth
from mopa_amd.common.utils.loss import seg_ce
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_3d
from mopa_amd.optim import FlatAdam
from mopa_amd.sparse3d import Geometry3D

b = synth.make_batch(8, H=16, W=16)
m = build_model_3d(default_cfg(num_classes=5, dual_head=True))[0].cuda().train()
opt = FlatAdam(m.parameters())
locs, feats, lab = b["x"][0].cuda(), b["x"][1].cuda(), b["seg_label"].cuda()
T = {k: 0.0 for k in ("zero", "geom", "fwd", "loss", "bwd", "opt")}
ENQ = dict(T)


def phase(name, fn):
    t0 = time.perf_counter()
    r = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ENQ[name] += t1 - t0
    T[name] += t2 - t0
    return r


def step():
    phase("zero", opt.zero_grad)
    g = phase("geom", lambda: Geometry3D(locs, 7, 4096, "cuda"))
    out = phase("fwd", lambda: m({"x": [locs, feats], "geometry_3d": g}))
    loss = phase("loss", lambda: seg_ce(out["seg_logit"], lab) + seg_ce(out["seg_logit2"], lab))
    phase("bwd", loss.backward)
    phase("opt", opt.step)


for _ in range(5):
    step()
for k in T:
    T[k] = ENQ[k] = 0.0
N = 30
for _ in range(N):
    step()
print("phase      enqueue ms   enqueue+device ms")
for k in T:
    print(f"{k:8s} {1e3 * ENQ[k] / N:10.3f} {1e3 * T[k] / N:14.3f}")
print(f"total    {1e3 * sum(ENQ.values()) / N:10.3f} {1e3 * sum(T.values()) / N:14.3f}")
