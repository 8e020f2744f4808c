import sys, os, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_configs as T
from mopa_amd.common.utils.loss import seg_ce, xm_kl
B = 8
src, trg = T._bench_batches(B)
m2, m3 = T._models()
cw = torch.tensor(T.CLASS_WEIGHTS, device="cuda")
def half2d(b):
    o2 = m2({"img": b["img"], "point_pix_2d": b["pix"], "img_indices": None})
    l2 = seg_ce(o2["seg_logit"], b["label"], cw)
    l2.backward()
    torch.cuda.synchronize()
    return float(l2)
def zero():
    for p in m2.parameters(): p.grad = None
state = {k: v.clone() for k, v in m2.state_dict().items()}
names = [n for n, _ in m2.named_parameters()]
runs = []
for r in range(4):
    zero(); m2.load_state_dict(state)
    l = half2d(src)
    runs.append((l, [(p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for p in m2.parameters()]))
print("losses", [r[0] for r in runs])
for r in range(1, 4):
    bad = [(n, float((a - b).abs().max() / (a.abs().max() + 1e-20))) for n, a, b in zip(names, runs[0][1], runs[r][1]) if not torch.equal(a, b)]
    print("run", r, "vs 0: differing tensors", len(bad), sorted(bad, key=lambda t: -t[1])[:6])
from mopa_amd import dense2d
print(dense2d.GRAPH_STATS)
# reference: the first one-kernel form
dense2d.WINO4_CONV9 = False
ref = []
for r in range(3):
    zero(); m2.load_state_dict(state)
    l = half2d(src)
    ref.append([(p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for p in m2.parameters()])
def worst(A, Bb):
    bad = [(n, float((a - b).abs().max() / (b.abs().max() + 1e-20))) for n, a, b in zip(names, A, Bb)]
    return sorted(bad, key=lambda t: -t[1])[:3]
print("conv9 off: run1 vs run0", worst(ref[1], ref[0])[:2], " run2 vs run0", worst(ref[2], ref[0])[:2])
for r in range(4):
    print("conv9 run", r, "vs conv9-off run 0:", worst(runs[r][1], ref[0]))
print("---- per-parameter relative error, conv9 run 0 vs conv9-off run 0 (network order)")
for n, a, b in zip(names, runs[0][1], ref[0]):
    e = float((a - b).abs().max() / (b.abs().max() + 1e-20))
    if e > 1e-3 or "dec_" in n or n.startswith("linear"):
        print(f"{n:50s} {e:.3e}")
