#!/usr/bin/env python3
"""cProfile of the host side of one Net3DSeg forward+backward (where do the ~11 us per launch go?)."""
import cProfile, pstats, sys
import torch
sys.path.insert(0, ".")
from mopa_amd import synth
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_3d
from mopa_amd.optim import FlatAdam
from mopa_amd.sparse3d import Geometry3D

cfg = default_cfg(num_classes=5, dual_head=True)
b = synth.make_batch(8)
m = build_model_3d(cfg)[0].cuda().train()
opt = FlatAdam(m.parameters())
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}


locs_dev = b["x"][0].cuda()


def step(n):
    for _ in range(n):
        opt.zero_grad()
        batch["geometry_3d"] = Geometry3D(locs_dev, 7, 4096, "cuda")
        out = m(batch)
        (out["seg_logit"].square().mean() + out["seg_logit2"].square().mean()).backward()
        opt.step()
    torch.cuda.synchronize()


step(3)
pr = cProfile.Profile()
pr.enable()
step(10)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
pstats.Stats(pr).sort_stats("cumtime").print_stats(25)
