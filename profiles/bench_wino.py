#!/usr/bin/env python3
"""Winograd F(2x2,3x3) vs the direct implicit GEMM for the stride-1 3x3 layer shapes of the image branch (B = 8)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d
from mopa_amd._lib import call, ptr, stream

B = 8
shapes = [("L1 64->64 152x240", 64, 64, 152, 240), ("L2 128->128 76x120", 128, 128, 76, 120), ("L3 256->256 38x60", 256, 256, 38, 60),
          ("L4 512->512 19x30", 512, 512, 19, 30), ("D4 512->256 38x60", 512, 256, 38, 60), ("D3 256->128 76x120", 256, 128, 76, 120),
          ("D2 128->64 152x240", 128, 64, 152, 240), ("D1 128->64 304x480", 128, 64, 304, 480)]


def timed(fn, reps=5):
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
    w_oihw = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wl = torch.empty(3, 3, cin, cout, device="cuda")
    dense2d.relayout(w_oihw, wl, cout, cin, 3, 3, 0)
    out_d = torch.empty(B * H * W, cout, device="cuda")
    out_w = torch.empty(B * H * W, cout, device="cuda")
    g = dense2d._geom(B=B, IH=H, IW=W, OHl=H, OWl=W, OHa=H, OWa=W, IY0=-1, IX0=-1, TH=3, TW=3, KWF=3, Cin=cin, Cout=cout, ld_in=cin, ld_out=cout)

    def direct():
        call("mopa_conv2d_igemm", ptr(x), ptr(wl), None, ptr(out_d), ctypes.addressof(g), 0, stream())

    td = timed(direct)
    res = []
    for F in (2, 4):
        sfx = "" if F == 2 else "4"
        NP = (F + 2) ** 2
        th, tw = (H + F - 1) // F, (W + F - 1) // F
        T = B * th * tw
        U = torch.empty(NP, cin, cout, device="cuda")
        call(f"mopa_wino{sfx}_weight", ptr(w_oihw), cout, cin, 0, ptr(U), stream())
        V = torch.empty(NP, T, cin, device="cuda")
        M = torch.empty(NP, T, cout, device="cuda")
        g1 = dense2d._geom(B=1, IH=1, IW=T, OHl=1, OWl=T, OHa=1, OWa=T, TH=1, TW=1, KWF=1, Cin=cin, Cout=cout, ld_in=cin, ld_out=cout)

        def t_in():
            call(f"mopa_wino{sfx}_input", ptr(x), cin, B, H, W, cin, ptr(V), stream())

        def t_gemm():
            call("mopa_conv2d_igemm_batched", ptr(V), ptr(U), ptr(M), ctypes.addressof(g1), NP, T * cin, cin * cout, T * cout, 0, stream())

        def t_out():
            call(f"mopa_wino{sfx}_output", ptr(M), B, H, W, cout, None, ptr(out_w), cout, 0, stream())

        def wino():
            t_in(); t_gemm(); t_out()

        tw_ = timed(wino)
        ti, tg, to = timed(t_in), timed(t_gemm), timed(t_out)
        err = float((out_d - out_w).abs().max()) / float(out_d.abs().max())
        res.append(f"F({F}x{F}) {tw_:6.1f} us = in {ti:5.1f} + gemm {tg:5.1f} + out {to:5.1f}, {td / tw_:4.2f}x, err {err:.1e}")
    td = timed(direct)
    flops = 2.0 * B * H * W * cout * 9 * cin
    print(f"{name:22s} direct {td:6.1f} us ({flops / td / 1e6:5.1f} TF/s) | " + " | ".join(res))
