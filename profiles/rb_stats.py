#!/usr/bin/env python3
"""Distribution of 16-rule groups per 64-row tile in the grouped rulebooks of the bench geometry."""
import sys
import torch
sys.path.insert(0, ".")
from mopa_amd import sparse3d as s3, synth  # noqa: E402
b = synth.make_batch(8, H=16, W=16)
g = s3.Geometry3D(b["x"][0], 7, 4096, "cuda")
for l in range(4):
    gs = g.rulebook(g.nbr27[l])[0]
    n = (gs[1:] - gs[:-1]).float().cpu()
    qs = torch.quantile(n, torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0]))
    print(f"level {l}: tiles {n.numel()} groups/tile mean {n.mean():.1f} quantiles(0,10,50,90,99,100) {[int(x) for x in qs]}")
