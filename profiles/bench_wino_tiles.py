#!/usr/bin/env python3
"""Tile sweep for the 16 batched GEMMs of the Winograd path (mopa_conv2d_igemm_batched, flags bits 8-15 = tile + 1)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d
from mopa_amd._lib import call, ptr, stream

B = 8
shapes = [("L2 128->128 76x120", 128, 128, 76, 120), ("L3 256->256 38x60", 256, 256, 38, 60), ("L4 512->512 19x30", 512, 512, 19, 30),
          ("D4 512->256 38x60", 512, 256, 38, 60), ("D4' 256->512 38x60", 256, 512, 38, 60), ("D3 256->128 76x120", 256, 128, 76, 120),
          ("D3' 128->256 76x120", 128, 256, 76, 120), ("L3a 128->256 38x60", 128, 256, 38, 60), ("L4a 256->512 19x30", 256, 512, 19, 30)]
names = ["256x64", "128x128", "128x64", "64x64"]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


F = int(sys.argv[1]) if len(sys.argv) > 1 else 2      # 2: F(2x2,3x3), 16 points; 4: F(4x4,3x3), 36 points
NP = (F + 2) ** 2
if F == 4:
    shapes = shapes + [("L1 64->64 152x240", 64, 64, 152, 240), ("D2 128->64 152x240", 128, 64, 152, 240),
                       ("D2' 64->128 152x240", 64, 128, 152, 240), ("D1 128->64 304x480", 128, 64, 304, 480),
                       ("D1' 64->128 304x480", 64, 128, 304, 480)]
for name, cin, cout, H, W in shapes:
    T = B * ((H + F - 1) // F) * ((W + F - 1) // F)
    U = torch.randn(NP, cin, cout, device="cuda")
    V = torch.randn(NP, T, cin, device="cuda")
    M = torch.empty(NP, T, cout, device="cuda")
    g1 = dense2d._geom(B=1, IH=1, IW=T, OHl=1, OWl=T, OHa=1, OWa=T, TH=1, TW=1, KWF=1, Cin=cin, Cout=cout, ld_in=cin, ld_out=cout)
    gf = NP * T * cin * cout * 2 / 1e9
    row = []
    for tile in range(4):
        if tile == 1 and cout % 128:
            row.append("   -   ")
            continue
        fl = (tile + 1) << 8
        us = timed(lambda: call("mopa_conv2d_igemm_batched", ptr(V), ptr(U), ptr(M), ctypes.addressof(g1), NP, T * cin, cin * cout, T * cout, fl, stream()))
        row.append(f"{us:6.1f}us {gf / us * 1e3:5.1f}TF")
    us = timed(lambda: call("mopa_conv2d_igemm_batched", ptr(V), ptr(U), ptr(M), ctypes.addressof(g1), NP, T * cin, cin * cout, T * cout, 0, stream()))
    print(f"{name:22s} T={T:6d} " + " | ".join(f"{n}: {r}" for n, r in zip(names, row)) + f" | auto {us:6.1f}us")
