#!/usr/bin/env python3
"""Host floor of the step: the same networks on a tiny input (the launch count is the same, the GPU work is negligible), so the
time per forward+backward is what Python / torch / the launch calls cost."""
import sys, time
import torch
sys.path.insert(0, ".")
from mopa_amd import synth
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d, build_model_3d
from mopa_amd.optim import FlatAdam

cfg = default_cfg(num_classes=5, dual_head=True)
for name, build, kw in (("2D", build_model_2d, dict(H=64, W=96)), ("3D", build_model_3d, dict(H=16, W=16))):
    b = synth.make_batch(2, **kw)
    m = build(cfg)[0].cuda().train()
    opt = FlatAdam(m.parameters())
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    for it in range(13):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m(batch)
        (out["seg_logit"].square().mean() + out["seg_logit2"].square().mean()).backward()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host {(t1 - t0) / 10 * 1e3:.1f} ms per fwd+bwd on a tiny input (GPU-complete {(t2 - t0) / 10 * 1e3:.1f} ms)")
