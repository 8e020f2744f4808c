#!/usr/bin/env python3
"""A few launches of the fused F(4x4) GEMM + output kernel on one shape (PMC / trace runs).  args: cin cout H W"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd._lib import call, ptr, stream
a = [int(x) for x in sys.argv[1:]] + [64, 64, 152, 240][len(sys.argv) - 1:]
B, cin, cout, H, W = 8, a[0], a[1], a[2], a[3]
T = B * ((H + 3) // 4) * ((W + 3) // 4)
V = torch.randn(36, T, cin, device="cuda")
Ut = torch.randn(36, cout, cin, device="cuda") * 0.05
out = torch.zeros(B * H * W, cout, device="cuda")
for _ in range(4):
    call("mopa_wino4_gemm_output", ptr(V), ptr(Ut), None, ptr(out), cout, B, H, W, cin, cout, 0, stream())
torch.cuda.synchronize()
