#!/usr/bin/env python3
"""The one-kernel F(4x4) convolution, both forms, per layer shape: mopa_wino4_conv (all 36 points per wave, 16x16x4 MFMAs; by shape
k_wino4_conv or k_wino4_conv32) against mopa_wino4_conv9 (nine points per wave, 32x32x2 MFMAs, csrc/wino4c9.hip).  us per
convolution, TF/s as executed (36 T Cin Cout 2), max difference relative to the output's scale.
Usage: python profiles/bench_conv9.py [images=16] [reps=10]"""
import sys
import torch

sys.path.insert(0, ".")
from mopa_amd._lib import call, ptr, stream  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
print(f"{B} images; us per convolution")
print(f"{'layer':>22} {'T':>7} | {'conv':>8} {'TF/s':>6} | {'conv9':>8} {'TF/s':>6} {'x conv':>7} {'rel diff':>9}")
g = torch.Generator(device="cuda").manual_seed(1)
for name, cin, cout, H, W in (("layer1 64->64", 64, 64, 152, 240), ("dec2 128->64", 128, 64, 152, 240), ("dec2 dgrad 64->128", 64, 128, 152, 240),
                              ("dec1 128->64", 128, 64, 304, 480), ("dec1 dgrad 64->128", 64, 128, 304, 480), ("layer2 128->128", 128, 128, 76, 120),
                              ("layer3 256->256", 256, 256, 38, 60)):
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    x = torch.randn(B * H * W, cin, device="cuda", generator=g)
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05
    Uf, Uq = torch.empty(36, cin, cout, device="cuda"), torch.empty(36, cin, cout, device="cuda")
    call("mopa_wino4_weight_f", ptr(w), cout, cin, 0, ptr(Uf), stream())
    call("mopa_wino4_weight_q", ptr(w), cout, cin, 0, ptr(Uq), stream())
    o1, o2 = torch.empty(B * H * W, cout, device="cuda"), torch.empty(B * H * W, cout, device="cuda")
    fl = 36 * T * cin * cout * 2
    t1 = timed(lambda: call("mopa_wino4_conv", ptr(x), cin, ptr(Uf), None, ptr(o1), cout, B, H, W, cin, cout, 0, None, 1, 0, None, stream()), reps)
    t2 = timed(lambda: call("mopa_wino4_conv9", ptr(x), cin, ptr(Uq), None, ptr(o2), cout, B, H, W, cin, cout, 0, None, 1, 0, stream()), reps)
    rel = float((o1 - o2).abs().max() / o1.abs().max())
    print(f"{name:>22} {T:>7} | {t1:>8.1f} {fl / t1 / 1e6:>6.1f} | {t2:>8.1f} {fl / t2 / 1e6:>6.1f} {t2 / t1:>7.2f} {rel:>9.1e}", flush=True)
