#!/usr/bin/env python3
"""Tile sweep of mopa_conv2d_igemm_batched on the 36-point GEMMs of the F(4x4) layers (M = tiles per point, Cin -> Cout), at the
tile counts of 8 and 16 images.  Usage: python profiles/bench_igemm_batched.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d
from mopa_amd._lib import call, ptr, stream

NP = 36
for B in (4, 8, 16):
    for name, cin, cout, H, W in (("L2 128->128 76x120", 128, 128, 76, 120), ("L3 256->256 38x60", 256, 256, 38, 60),
                                  ("L4 512->512 19x30", 512, 512, 19, 30), ("D4 512->256 38x60", 512, 256, 38, 60),
                                  ("D3 256->128 76x120", 256, 128, 76, 120), ("L2b 128->256 38x60", 128, 256, 38, 60)):
        T = B * ((H + 3) // 4) * ((W + 3) // 4)
        V = torch.randn(NP * T * cin, device="cuda")
        U = torch.randn(NP * cin * cout, device="cuda") * 0.05
        M = torch.empty(NP * T * cout, device="cuda")
        g = dense2d._geom(B=1, IH=1, IW=T, OHl=1, OWl=T, OHa=1, OWa=T, TH=1, TW=1, KWF=1, Cin=cin, Cout=cout, ld_in=cin, ld_out=cout)
        flops = 2.0 * NP * T * cin * cout
        res = []
        for tile in (0, 1, 2, 3, -1):
            if tile >= 0 and cout % (64, 128, 64, 64)[tile]:
                res.append("    - ")
                continue
            flags = ((tile + 1) << 8) if tile >= 0 else 0
            for _ in range(2):
                call("mopa_conv2d_igemm_batched", ptr(V), ptr(U), ptr(M), ctypes.addressof(g), NP, T * cin, cin * cout, T * cout, flags, stream())
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                call("mopa_conv2d_igemm_batched", ptr(V), ptr(U), ptr(M), ctypes.addressof(g), NP, T * cin, cin * cout, T * cout, flags, stream())
            e.record()
            torch.cuda.synchronize()
            res.append("%6.1f" % (flops * 10 / (s.elapsed_time(e) * 1e-3) / 1e12))
        print("B=%-2d %-22s T=%-6d TF/s  256x64 %s | 128x128 %s | 128x64 %s | 64x64 %s | auto %s" % (B, name, T, *res), flush=True)
