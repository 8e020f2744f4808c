#!/usr/bin/env python3
"""Which objects of a training step are only freed by Python's cyclic GC?  (They pin activations: the caching allocator
grows every step until a generation-2 collection.)  Runs a few steps with the GC off, then collects with DEBUG_SAVEALL and
prints the garbage by type plus, for tensors, who refers to them.  Usage: find_cycles.py [3d|joint]"""
import collections, gc, sys
import torch
sys.path.insert(0, ".")
from mopa_amd import synth
from mopa_amd.common.utils.loss import seg_ce, xm_kl
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d, build_model_3d
from mopa_amd.sparse3d import Geometry3D

mode = sys.argv[1] if len(sys.argv) > 1 else "3d"
cfg = default_cfg(num_classes=5, dual_head=True)
b = synth.make_batch(2, H=64, W=96)
m3 = build_model_3d(cfg)[0].cuda().train()
m2 = build_model_2d(cfg)[0].cuda().train() if mode == "joint" else None
lab = b["seg_label"].cuda()


def step():
    if mode == "3d":
        geom = Geometry3D(b["x"][0], 7, 4096, "cuda")
        out = m3({"x": b["x"], "geometry_3d": geom})
        loss = seg_ce(out["seg_logit"], lab) + seg_ce(out["seg_logit2"], lab)
        loss.backward()
    else:
        o2, o3 = m2(b), m3(b)
        l2 = seg_ce(o2["seg_logit"], lab) + xm_kl(o2["seg_logit2"], o3["seg_logit"])
        l3 = seg_ce(o3["seg_logit"], lab) + xm_kl(o3["seg_logit2"], o2["seg_logit"])
        l2.backward()
        l3.backward()


step()
gc.collect()
gc.disable()
a0 = torch.cuda.memory_allocated()
for _ in range(3):
    step()
torch.cuda.synchronize()
a1 = torch.cuda.memory_allocated()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print(f"{mode}: allocated {a0 / 1e6:.1f} -> {a1 / 1e6:.1f} MB over 3 steps; gc.collect() found {n} unreachable objects")
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
print(cnt.most_common(15))
seen = 0
for o in gc.garbage:
    if isinstance(o, (torch.autograd.function.FunctionCtx,)) or type(o).__name__.endswith("Backward"):
        print("node:", type(o).__name__)
    if torch.is_tensor(o) and seen < 3:
        seen += 1
        print("tensor", tuple(o.shape), "grad_fn", type(o.grad_fn).__name__ if o.grad_fn is not None else None)
        for r in gc.get_referrers(o)[:6]:
            print("   referred by", type(r).__name__, (list(r.keys())[:8] if isinstance(r, dict) else ""))
fn = [o for o in gc.garbage if type(o).__name__ == "function"]
for f in fn[:12]:
    print("function", f.__qualname__, "closure of", [type(c.cell_contents).__name__ for c in (f.__closure__ or ()) if True][:8])
