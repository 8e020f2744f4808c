#!/usr/bin/env python3
"""F(4x4) conv: input transform + fused GEMM/output kernel (or batched GEMM + output transform) vs the ONE-kernel convolution
(mopa_wino4_conv), per layer shape of the joint step (B = 16 by default); checks that the two agree.
Usage: python profiles/bench_wino_direct.py [B]     (WANT_V=1: both paths keep V -- the training forward role)"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d
from mopa_amd._lib import call, ptr, stream

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [("L1 64->64 152x240", 64, 64, 152, 240), ("D2 128->64 152x240", 128, 64, 152, 240), ("D2' 64->128 152x240", 64, 128, 152, 240),
          ("D1 128->64 304x480", 128, 64, 304, 480), ("D1' 64->128 304x480", 64, 128, 304, 480), ("L2 128->128 76x120", 128, 128, 76, 120),
          ("L3 256->256 38x60", 256, 256, 38, 60), ("odd 64->64 37x51", 64, 64, 37, 51)]


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


dense2d.WINO4_DIRECT_MAX_CIN = 1 << 30
dense2d.WINO4_DIRECT = True
dense2d.WINO4_DIRECT_ROLES = ("fwd", "fwd_eval", "dgrad")
for name, cin, cout, H, W in shapes:
    x = torch.randn(B * H * W, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    bias = torch.randn(cout, device="cuda")
    res, outs = [], []
    for direct in (False, True):
        dense2d.WINO4_DIRECT_MIN_TILES = 0 if direct else 1 << 62
        lay = dense2d.wino4_layout(cin, cout, B, H, W)
        U = torch.empty(36, cout, cin, device="cuda") if lay == 1 else torch.empty(36, cin, cout, device="cuda")
        call(("mopa_wino4_weight", "mopa_wino4_weight_t", "mopa_wino4_weight_f")[lay], ptr(w), cout, cin, 0, ptr(U), stream())
        o = torch.zeros(B * H * W, cout, device="cuda")
        # want_v=False: the backward-data role (the default use of the one-kernel path) keeps no V
        res.append(timed(lambda: dense2d.wino_conv(ptr(x), cin, B, H, W, cin, cout, U, bias, ptr(o), cout, F=4, want_v=(not direct) or os.environ.get("WANT_V", "0") != "0")))
        outs.append(o)
    err = float((outs[0] - outs[1]).abs().max()) / float(outs[0].abs().max())
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    gf = 36 * T * cin * cout * 2 / 1e9
    print(f"{name:22s} T={T:6d}  transform + GEMM kernels {res[0]:8.1f} us   one kernel {res[1]:8.1f} us  ({gf / res[1] * 1e3:5.1f} TF/s)   "
          f"rel diff {err:.1e}", flush=True)
