#!/usr/bin/env python3
"""The stem convolution (SubmanifoldConvolution 1 -> 16, mopa/models/scn_unet.py:27) and its weight gradient at the bench geometry:
us per launch of mopa_spconv_fwd / mopa_spconv_bwd_weight.  MOPA_SPCONV_STEM=0: the MFMA kernels of rounds 1-4 (16-wide K padding).
Usage: python profiles/bench_stem.py [scans=8]"""
import sys
import torch
sys.path.insert(0, ".")
from mopa_amd import sparse3d as s3, synth
from mopa_amd._lib import call, ptr, stream
from bench_spconv import timed
scans = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = s3.Geometry3D(synth.make_batch(scans, H=16, W=16)["x"][0], 1, 4096, "cuda")
tab = g.nbr27[0]
K, A = tab.shape
x = torch.rand(A, 1, device="cuda") + 0.5
w = torch.randn(K, 1, 16, device="cuda")
out = s3.new_view(A, 16, "cuda")
dy = s3.View(torch.randn(A, 16, device="cuda"))
dw = torch.empty_like(w)
xv = s3.View(x)
t1 = timed(lambda: call("mopa_spconv_fwd", ptr(tab), K, A, xv.p, xv.ld, 1, ptr(w), 16, 0, out.p, out.ld, stream()), 30)
t2 = timed(lambda: s3.spconv_bwd_weight(tab, xv, dy, dw), 30)
rules = int((tab >= 0).sum())
alg = rules * 4 + A * 64 + rules * 8 + K * 64
print(f"scans {scans} rows {A} rules {rules}: forward {t1:.1f} us ({alg / t1 / 8e6:.3f} of 8 TB/s at SURVEY 8d's bytes), weight gradient {t2:.1f} us")
