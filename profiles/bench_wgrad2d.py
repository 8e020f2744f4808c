#!/usr/bin/env python3
"""The F(4x4) weight gradient of the 2D network's stride-1 3x3 layers, per layer shape: the two-operand form (mopa_wino4_dout +
mopa_wino4_bwd_weight on the V the forward pass kept; `+in` = with the input transform a V-free forward pass would have to repeat)
against the one-kernel form (mopa_wino4_wgrad_fused, csrc/wino4wg.hip: x and dY in, dW out).  us per weight gradient, TF/s as
executed (36 T Cin Cout 2 flops), GB/s of the algorithmic bytes (x + dY read once), max difference relative to the gradient's scale.
Usage: python profiles/bench_wgrad2d.py [images=16] [reps=10]"""
import sys
import torch

sys.path.insert(0, ".")
from mopa_amd import dense2d  # noqa: E402
from mopa_amd._lib import call, ptr, query, stream  # noqa: E402


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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    shapes = [("layer1 64->64", 64, 64, 152, 240), ("dec2 128->64", 128, 64, 152, 240), ("dec1 128->64", 128, 64, 304, 480),
              ("layer2 128->128", 128, 128, 76, 120), ("layer1 @225x400", 64, 64, 120, 200), ("dec1 @225x400", 128, 64, 240, 400)]
    print(f"{B} images; us per weight gradient")
    print(f"{'layer':>18} {'T':>7} | {'dout':>7} {'gemm+dw':>8} {'two-op':>8} {'TF/s':>6} {'+in':>8} | {'one':>8} {'TF/s':>6} {'GB/s':>6} "
          f"{'x two-op':>8} {'rel diff':>9} {'slab MB':>8}")
    g = torch.Generator(device="cuda").manual_seed(1)
    for name, cin, cout, H, W in shapes:
        T = B * ((H + 3) // 4) * ((W + 3) // 4)
        x = dense2d.Img(torch.randn(B * H * W, cin, device="cuda", generator=g), B, H, W)
        dy = dense2d.Img(torch.randn(B * H * W, cout, device="cuda", generator=g), B, H, W)
        V = torch.empty(36 * T * cin, device="cuda")
        dM = torch.empty(36 * T * cout, device="cuda")
        dw1, dw2 = torch.zeros(cout, cin, 3, 3, device="cuda"), torch.zeros(cout, cin, 3, 3, device="cuda")
        ws = torch.empty(query("mopa_wino4_wgrad_workspace_bytes", T, cin, cout), dtype=torch.uint8, device="cuda")
        t_in = timed(lambda: call("mopa_wino4_input", x.p, x.ld, B, H, W, cin, ptr(V), stream()), reps)
        t_do = timed(lambda: call("mopa_wino4_dout", dy.p, dy.ld, B, H, W, cout, ptr(dM), stream()), reps)
        t_gm = timed(lambda: call("mopa_wino4_bwd_weight", ptr(V), ptr(dM), T, cin, cout, ptr(dw1), 2, ptr(ws), ws.numel(), stream()), reps)
        fl = 36 * T * cin * cout * 2
        line = f"{name:>18} {T:>7} | {t_do:>7.1f} {t_gm:>8.1f} {t_do + t_gm:>8.1f} {fl / (t_do + t_gm) / 1e6:>6.1f} {t_do + t_gm + t_in:>8.1f} | "
        if query("mopa_wino4_wgrad_fused_ok", B, H, W, cin, cout):
            wsb = query("mopa_wino4_wgrad_fused_workspace_bytes", B, H, W, cin, cout)
            ws2 = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            t_one = timed(lambda: call("mopa_wino4_wgrad_fused", x.p, x.ld, None, 1, 0, dy.p, dy.ld, B, H, W, cin, cout, ptr(dw2), 2,
                                       ptr(ws2), ws2.numel(), stream()), reps)
            rel = float((dw1 - dw2).abs().max() / dw1.abs().max())
            line += (f"{t_one:>8.1f} {fl / t_one / 1e6:>6.1f} {B * H * W * (cin + cout) * 4 / t_one / 1e3:>6.0f} {t_one / (t_do + t_gm):>8.2f} "
                     f"{rel:>9.1e} {wsb / 1e6:>8.1f}")
        else:
            line += "       -"
        print(line, flush=True)


if __name__ == "__main__":
    main()
