#!/usr/bin/env python3
"""cProfile of the host side of Net2DSeg forward+backward (same launch count as the bench, one image so that the GPU never
holds the host back; backward on the calling thread so that the profiler sees it)."""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, ".")
from mopa_amd import synth
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d
from mopa_amd.optim import FlatAdam

cfg = default_cfg(num_classes=5, dual_head=True)
b = synth.make_batch(1)
m = build_model_2d(cfg)[0].cuda().train()
opt = FlatAdam(m.parameters())
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
batch["point_pix_2d"] = m.pack_indices(b["img_indices"], b["img"].shape[2], b["img"].shape[3], "cuda")
torch.autograd.set_multithreading_enabled(False)


def step(n):
    for _ in range(n):
        out = m(batch)
        (out["seg_logit"].square().mean() + out["seg_logit2"].square().mean()).backward()


step(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
step(10)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host {(t1 - t0) * 100:.1f} ms per fwd+bwd, GPU-complete {(time.perf_counter() - t0) * 100:.1f} ms")
pr = cProfile.Profile()
pr.enable()
step(10)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
