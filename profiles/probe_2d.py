#!/usr/bin/env python3
"""GPU time of the 2D branch alone (fwd + bwd, bench batch) under different losses -- cross-check of the joint step."""
import sys, time
import torch
sys.path.insert(0, ".")
from mopa_amd import synth
from mopa_amd.common.utils.loss import seg_ce, xm_kl
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d
from mopa_amd.optim import FlatAdam

cfg = default_cfg(num_classes=5, dual_head=True)
b = synth.make_batch(8)
m = build_model_2d(cfg)[0].cuda().train()
opt = FlatAdam(m.parameters())
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
print("img", tuple(batch["img"].shape), "points", sum(len(i) for i in b["img_indices"]))
lab = b["seg_label"].cuda()
other = torch.randn(lab.shape[0], 5, device="cuda")
NIT = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name in ("square", "bench"):
    for it in range(NIT + 2):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad()
        out = m(batch)
        if name == "square":
            loss = out["seg_logit"].square().mean() + out["seg_logit2"].square().mean()
        else:
            loss = xm_kl(out["seg_logit2"], other) + seg_ce(out["seg_logit"], lab)
        loss.backward()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / NIT * 1e3:.1f} ms per fwd+bwd")
