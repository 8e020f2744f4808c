#!/usr/bin/env python3
"""The transform-domain weight gradient of the deep layers (mopa_wino4_bwd_weight: dU[p] = V[p]^T dM[p] + the slab reduction), per
layer shape at 16 images: us per call, TF/s as executed.  MOPA_WGEMM=0 = k_conv2d_wgrad_mfma, default = the ring-buffered GEMM
(csrc/wgemm.hip).  Under rocprofv3 --kernel-trace --stats the two kernels of a call separate.  Usage: python profiles/bench_wgemm.py [reps=20]"""
import sys
import torch

sys.path.insert(0, ".")
from mopa_amd._lib import call, ptr, query, stream  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print(f"{'layer':>16} {'T':>6} {'cin':>4} {'cout':>4} {'us':>8} {'TF/s':>6}")
for name, T, cin, cout in (("layer2", 9120, 128, 128), ("layer3", 2400, 256, 256), ("layer4", 640, 512, 512), ("dec4", 2400, 512, 256),
                           ("dec3", 9120, 256, 128)):
    V = torch.randn(36 * T * cin, device="cuda")
    dM = torch.randn(36 * T * cout, device="cuda")
    dw = torch.zeros(cout, cin, 3, 3, device="cuda")
    ws = torch.empty(query("mopa_wino4_wgrad_workspace_bytes", T, cin, cout), dtype=torch.uint8, device="cuda")
    t = timed(lambda: call("mopa_wino4_bwd_weight", ptr(V), ptr(dM), T, cin, cout, ptr(dw), 2, ptr(ws), ws.numel(), stream()), reps)
    print(f"{name:>16} {T:>6} {cin:>4} {cout:>4} {t:>8.1f} {36 * T * cin * cout * 2 / t / 1e6:>6.1f}   slabs {ws.numel() / 1e6:.1f} MB", flush=True)
