#!/usr/bin/env python3
"""Device pseudo-label update (fusion + per-class median refinement) vs the reference's host round trip, one target batch."""
import sys, time
import torch
sys.path.insert(0, ".")
from mopa_amd import pseudo
from oracle import pseudo as opseudo

n, c = 279040, 5
g = torch.Generator().manual_seed(0)
l2, l3 = (torch.randn(n, c, generator=g) * 2).cuda(), (torch.randn(n, c, generator=g) * 2).cuda()
for _ in range(3):
    pseudo.pseudo_labels(l2, l3, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    a, b = pseudo.pseudo_labels(l2, l3, True)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t0) / 20 * 1e3
t0 = time.perf_counter()
for _ in range(3):
    r2, r3 = opseudo.pseudo_labels(l2.cpu(), l3.cpu(), True)   # the reference: D2H, torch-CPU / numpy, H2D
    r2, r3 = r2.cuda(), r3.cuda()
torch.cuda.synchronize()
host_ms = (time.perf_counter() - t0) / 3 * 1e3
print(f"pseudo-label update, {n} points x {c} classes: device {dev_ms:.3f} ms, host path {host_ms:.1f} ms "
      f"({torch.get_num_threads()} threads); labels differ on {float((a.cpu() != r2.cpu()).float().mean()):.2e} of the points")
