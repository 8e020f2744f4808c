#!/usr/bin/env python3
"""Per-call timing of the 2D branch alone (Net2DSeg, B x 302 x 480, forward + backward on ONE stream): every C-ABI call made by
mopa_amd/dense2d.py is bracketed with HIP events and aggregated by (entry point, shape key).  Answers "which layer shapes carry
the time and how fast is each, without cross-stream contention".  Usage: python profiles/by_layer_2d.py [B]"""
import collections, ctypes, os, sys
os.environ.setdefault("MOPA_WGRAD_STREAM", "0")
os.environ.setdefault("MOPA_GRAPH_2D", "0")   # every launch goes through the wrapped Python calls
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 302, 480
rec = []
on = [False]
inner_call = dense2d.call


def wrapped(name, *args):
    if not on[0]:
        return inner_call(name, *args)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = inner_call(name, *args)
    e.record()
    key = tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and 0 <= a < (1 << 31))[:8]
    rec.append((name, key, s, e))
    return r


dense2d.call = wrapped
inner_igemm, inner_b = dense2d.igemm, dense2d.igemm_batched
grec = []


def igemm(x_ptr, w, bias, out_ptr, geom, accumulate=False):
    if on[0]:
        g = list(geom)
        grec.append(("igemm", (g[21], g[22], g[15] * g[16], g[0] * g[3] * g[4], 1), len(rec)))
    return inner_igemm(x_ptr, w, bias, out_ptr, geom, accumulate)


def igemm_b(x_ptr, w_ptr, out_ptr, geom, nbatch, a, b, c):
    if on[0]:
        g = list(geom)
        grec.append(("igemm_batched", (g[21], g[22], 1, g[0] * g[3] * g[4], nbatch), len(rec)))
    return inner_b(x_ptr, w_ptr, out_ptr, geom, nbatch, a, b, c)


dense2d.igemm, dense2d.igemm_batched = igemm, igemm_b

model, _ = build_model_2d(default_cfg(5, True))
model = model.cuda().train()
rng = np.random.Generator(np.random.PCG64(0))
img = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32)).cuda()
idx = [np.stack([rng.integers(0, H, 34880), rng.integers(0, W, 34880)], 1) for _ in range(B)]
pix = model.pack_indices(idx, H, W, "cuda")


def step():
    out = model({"img": img, "point_pix_2d": pix, "img_indices": None})
    (out["seg_logit"].square().mean() + out["seg_logit2"].square().mean()).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
on[0] = True
N = 3
for _ in range(N):
    step()
torch.cuda.synchronize()
on[0] = False
# the igemm geometry of a call = the grec entry whose rec index matches
gmap = {i: (k, key) for k, key, i in grec}
agg = collections.defaultdict(lambda: [0, 0.0])
for i, (name, key, s, e) in enumerate(rec):
    if i in gmap:
        k = ("conv " + gmap[i][0],) + gmap[i][1]
    else:
        k = (name,) + key[:6]
    a = agg[k]
    a[0] += 1
    a[1] += s.elapsed_time(e) * 1e3
tot = sum(a[1] for a in agg.values())
print(f"2D branch alone, B={B}: {tot / N / 1e3:.2f} ms of bracketed calls per fwd+bwd (one stream)")
byname = collections.defaultdict(float)
for k, a in agg.items():
    byname[k[0]] += a[1]
print("--- by entry point (ms per step)")
for n, t in sorted(byname.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{n:34s} {t / N / 1e3:7.2f}")
print("--- conv GEMM launches by shape: (cin, cout, taps, M rows, batches) calls/step  avg us  TF/s  ideal GB/s")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if not k[0].startswith("conv"):
        continue
    cin, cout, taps, M, nb = k[1:]
    us = a[1] / a[0]
    fl = 2.0 * M * cin * cout * taps * nb
    by = 4.0 * nb * (M * cin + M * cout) + 4.0 * taps * cin * cout * nb
    print(f"{k[0]:20s} {str(k[1:]):38s} {a[0] / N:5.1f} {us:8.1f} {fl / us / 1e6:6.1f} {by / us / 1e3:7.0f}   {a[1] / N / 1e3:6.2f} ms/step")
print("--- other calls by (name, first int args): calls/step avg us ms/step")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    if k[0].startswith("conv"):
        continue
    print(f"{k[0]:30s} {str(k[1:]):44s} {a[0] / N:5.1f} {a[1] / a[0]:8.1f} {a[1] / N / 1e3:6.2f}")
