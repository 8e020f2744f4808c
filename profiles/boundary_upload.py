#!/usr/bin/env python3
"""PCIe-inclusive rate at the reference boundary: the models called with HOST tensors (what `collate_scn_base` hands over)
vs the same tensors resident in HBM, one domain of 8 scans (2D + 3D forward + backward, squared-logit losses)."""
import sys, time
import torch
sys.path.insert(0, ".")
from mopa_amd import synth
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d, build_model_3d
from mopa_amd.optim import FlatAdam

cfg = default_cfg(num_classes=5, dual_head=True)
b = synth.make_batch(8)
m2, m3 = build_model_2d(cfg)[0].cuda().train(), build_model_3d(cfg)[0].cuda().train()
o2, o3 = FlatAdam(m2.parameters()), FlatAdam(m3.parameters())


def run(batch, n=8):
    def one():
        p2, p3 = m2(batch), m3(batch)
        (p2["seg_logit"].square().mean() + p3["seg_logit"].square().mean()).backward()
    for _ in range(3):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


host = run(b)
res = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
res["x"] = [t.cuda() for t in b["x"]]
res["point_pix_2d"] = m2.pack_indices(b["img_indices"], b["img"].shape[2], b["img"].shape[3], "cuda")
dev = run(res)
print(f"8 scans, 2D + 3D fwd+bwd (sequential, one stream): host tensors {host:.1f} ms = {8e3 / host:.0f} scans/s, resident {dev:.1f} ms = {8e3 / dev:.0f} scans/s")
