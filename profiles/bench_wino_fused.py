#!/usr/bin/env python3
"""F(4x4) conv: batched GEMM + output transform (round 1) vs the fused GEMM + output-transform kernel (mopa_wino4_gemm_output),
per layer shape, B = 8; checks that the two agree."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d
from mopa_amd._lib import call, ptr, stream

B = 8
shapes = [("L1 64->64 152x240", 64, 64, 152, 240), ("D2 128->64 152x240", 128, 64, 152, 240), ("D2' 64->128 152x240", 64, 128, 152, 240),
          ("D1 128->64 304x480", 128, 64, 304, 480), ("D1' 64->128 304x480", 64, 128, 304, 480), ("L2 128->128 76x120", 128, 128, 76, 120),
          ("L3 256->256 38x60", 256, 256, 38, 60), ("L4 512->512 19x30", 512, 512, 19, 30), ("odd 64->64 37x51", 64, 64, 37, 51)]


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


for name, cin, cout, H, W in shapes:
    x = torch.randn(B * H * W, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    bias = torch.randn(cout, device="cuda")
    U = torch.empty(36, cin, cout, device="cuda")
    Ut = torch.empty(36, cout, cin, device="cuda")
    call("mopa_wino4_weight", ptr(w), cout, cin, 0, ptr(U), stream())
    call("mopa_wino4_weight_t", ptr(w), cout, cin, 0, ptr(Ut), stream())
    assert torch.equal(U.transpose(1, 2).contiguous(), Ut)
    o1 = torch.zeros(B * H * W, cout, device="cuda")
    o2 = torch.zeros(B * H * W, cout, device="cuda")
    res = []
    for fused, U_, o in ((False, U, o1), (True, Ut, o2)):
        dense2d.WINO4_FUSED_MIN_BLOCKS = 0 if fused else 1 << 62
        us = timed(lambda: dense2d.wino_conv(ptr(x), cin, B, H, W, cin, cout, U_, bias, ptr(o), cout, F=4))
        res.append(us)
    err = float((o1 - o2).abs().max()) / float(o1.abs().max())
    # accumulate mode
    dense2d.WINO4_FUSED_MIN_BLOCKS = 0
    o3 = o1.clone()
    dense2d.wino_conv(ptr(x), cin, B, H, W, cin, cout, Ut, None, ptr(o3), cout, accumulate=True, F=4)
    dense2d.WINO4_FUSED_MIN_BLOCKS = 1 << 62
    o4 = o1.clone()
    dense2d.wino_conv(ptr(x), cin, B, H, W, cin, cout, U, None, ptr(o4), cout, accumulate=True, F=4)
    err2 = float((o3 - o4).abs().max()) / float(o4.abs().max())
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    gf = 36 * T * cin * cout * 2 / 1e9
    print(f"{name:22s} T={T:6d}  batched+out {res[0]:8.1f} us   fused {res[1]:8.1f} us  ({gf / res[1] * 1e3:5.1f} TF/s incl. input transform)   rel diff {err:.1e} / acc {err2:.1e}")
