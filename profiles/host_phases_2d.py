#!/usr/bin/env python3
"""Host enqueue time of the 2D branch (Net2DSeg, B x 302 x 480): forward, losses, backward -- wall time of the Python calls with
the device idle at the start of each phase (a sync between phases), beside the device time of the phase."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 302, 480
model, _ = build_model_2d(default_cfg(5, True))
model = model.cuda().train()
from mopa_amd.optim import FlatAdam
opt = FlatAdam(model.parameters(), lr=1e-3)   # gradients attached to the flat buffer: what the replayed backward needs (as in training)
rng = np.random.Generator(np.random.PCG64(0))
img = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32)).cuda()
idx = [np.stack([rng.integers(0, H, 34880), rng.integers(0, W, 34880)], 1) for _ in range(B)]
pix = model.pack_indices(idx, H, W, "cuda")
ENQ = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0}
TOT = dict(ENQ)


def phase(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ENQ[name] += t1 - t0
    TOT[name] += t2 - t0
    return r


def step():
    opt.zero_grad()
    out = phase("fwd", lambda: model({"img": img, "point_pix_2d": pix, "img_indices": None}))
    loss = phase("loss", lambda: out["seg_logit"].square().mean() + out["seg_logit2"].square().mean())
    phase("bwd", loss.backward)


for _ in range(3):
    step()
for k in ENQ:
    ENQ[k] = TOT[k] = 0.0
N = 10
for _ in range(N):
    step()
from mopa_amd import dense2d
print(f"B={B}: 2D executor: {'hipGraph replay' if dense2d.GRAPH_2D else 'native command list' if dense2d.NATIVE_2D else 'python walk'}; {dense2d.GRAPH_STATS}")
print(f"B={B}: phase  host enqueue ms   enqueue+device ms")
for k in ENQ:
    print(f"{k:6s} {1e3 * ENQ[k] / N:10.2f} {1e3 * TOT[k] / N:14.2f}")
print(f"total  {1e3 * sum(ENQ.values()) / N:10.2f} {1e3 * sum(TOT.values()) / N:14.2f}")
