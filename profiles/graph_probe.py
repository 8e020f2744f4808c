#!/usr/bin/env python3
"""Feasibility probe: can the 2D branch (forward + backward through the C-ABI kernels, weight-gradient side stream included) be
captured into a HIP graph by torch.cuda.graph and replayed with identical results?  Usage: python profiles/graph_probe.py [B=2]"""
import sys, time
import torch
sys.path.insert(0, ".")
from mopa_amd import synth
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d
from mopa_amd.optim import FlatAdam

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = default_cfg(num_classes=5, dual_head=True)
b = synth.make_batch(B)
m = build_model_2d(cfg)[0].cuda().train()
m.net_2d.dropout.p = 0.0
opt = FlatAdam(m.parameters())
batch = {"img": b["img"].cuda(), "point_pix_2d": m.pack_indices(b["img_indices"], b["img"].shape[2], b["img"].shape[3], "cuda"), "img_indices": None}
print("img", tuple(b["img"].shape))
g1 = torch.randn(batch["point_pix_2d"].numel(), 5, device="cuda")


def step():
    out = m(batch)
    ((out["seg_logit"] * g1).sum() + (out["seg_logit2"] * g1).sum() * 0.5).backward()
    return out


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad()
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
opt.zero_grad()
o = step()
torch.cuda.synchronize()
ref_logit, ref_grad = o["seg_logit"].clone(), opt.grad.clone()
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_eager = time.perf_counter() - t0
print(f"eager: host {t_host * 100:.2f} ms, complete {t_eager * 100:.2f} ms per fwd+bwd")
g = torch.cuda.CUDAGraph()
opt.zero_grad()
try:
    with torch.cuda.graph(g):
        so = step()
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:500])
    sys.exit(1)
torch.cuda.synchronize()
opt.zero_grad()
g.replay()
torch.cuda.synchronize()
print("replay logits equal:", torch.equal(so["seg_logit"], ref_logit), "max diff", float((so["seg_logit"] - ref_logit).abs().max()))
print("replay grads equal:", torch.equal(opt.grad, ref_grad), "max diff", float((opt.grad - ref_grad).abs().max()), "scale", float(ref_grad.abs().max()))
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_graph = time.perf_counter() - t0
print(f"graph: host {t_host * 100:.2f} ms, complete {t_graph * 100:.2f} ms per fwd+bwd")
