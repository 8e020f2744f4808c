#!/usr/bin/env python3
"""The BatchNorm passes of the image branch alone (mopa_bn_act_fwd_groups / mopa_bn_act_bwd_groups, 2 groups, training): us per call and
the HBM rate over the bytes each pass has to move (forward: statistics read x, apply reads x [+ residual] and writes y; backward: sums
read dy, x [, y]; apply reads dy, x [, y] and writes dx [, dres]), at the shapes of the joint step (16 images).  Usage: python profiles/bench_bn.py"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d
from mopa_amd.dense2d import Img, bn_bwd_groups, bn_fwd_groups, new_img


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


def main():
    G = 2
    print(f"{'rows':>9} {'C':>4} {'variant':>22} {'fwd us':>8} {'TB/s':>6} {'bwd us':>8} {'TB/s':>6}")
    for B, H, W, C in ((16, 304, 480, 64), (16, 152, 240, 64), (16, 76, 120, 128), (16, 38, 60, 256), (16, 19, 30, 512)):
        rows = B * H * W
        x, y, dy, dx, res, dres = (new_img(B, H, W, C, "cuda") for _ in range(6))
        for t in (x, dy, res):
            t.t.normal_()
        P = {"bn.weight": torch.rand(C, device="cuda") + 0.5, "bn.bias": torch.randn(C, device="cuda"),
             "bn.running_mean": torch.zeros(C, device="cuda"), "bn.running_var": torch.ones(C, device="cuda")}
        stats = torch.empty(G, 4, C, device="cuda")
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        tb = rows * C * 4
        for name, r, ym, dr in (("bn + relu", None, None, None), ("bn + residual + relu", res, y, dres)):
            tf = timed(lambda: bn_fwd_groups(x, y, P, "bn", 1, r, True, stats, G))
            bf = tb * (3 + (1 if r is not None else 0))
            tw = timed(lambda: bn_bwd_groups(dy, x, dx, stats, 1, ym, dr, False, True, dg, db, G))
            bw = tb * (5 + (2 if ym is not None else 0) + (1 if dr is not None else 0))
            print(f"{rows:>9} {C:>4} {name:>22} {tf:>8.1f} {bf / tf / 1e6:>6.2f} {tw:>8.1f} {bw / tw / 1e6:>6.2f}", flush=True)


if __name__ == "__main__":
    main()
