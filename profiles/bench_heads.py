#!/usr/bin/env python3
"""Stand-alone timing of the streaming kernels around the 2D heads at the joint step's shape (16 images of 304x480 padded, 64 channels,
558,080 points, 5 classes): full-image head forward, point-head backward (dense d(feat) over all pixels + head weight gradients),
max-pool backward -- us per call and GB/s of their algorithmic bytes.  Usage: python profiles/bench_heads.py [B=16] [reps=20]"""
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = "cuda"
    Hp, Wp, H, W, M, C = 304, 480, 302, 480, 64, 5
    N = 34880 * B
    g = torch.Generator(device=dev).manual_seed(0)
    feat = torch.randn(B * Hp * Wp, M, device=dev, generator=g)
    w1, b1 = torch.randn(C, M, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
    w2 = torch.randn(C, M, device=dev, generator=g)
    pred = torch.empty(B, H, W, C, device=dev)
    rows = []

    def head_fwd():
        call("mopa_pixel_head_fwd", ptr(feat), M, B, Hp, Wp, H, W, M, C, ptr(w1), ptr(b1), ptr(pred), stream())
    t = timed(head_fwd, reps)
    rows.append(("mopa_pixel_head_fwd", t, (B * H * W * (M + C)) * 4))

    # point heads backward: CSR of points per pixel, then dense d(feat) + the two heads' weight gradients
    pix = torch.randint(0, B * Hp * Wp, (N,), device=dev, dtype=torch.int32, generator=g)
    nrow = B * Hp * Wp
    row_start = torch.empty(nrow + 1, dtype=torch.int32, device=dev)
    row_points = torch.empty(N, dtype=torch.int32, device=dev)
    ws = torch.empty(max(query("mopa_points_csr_workspace_bytes", nrow), 256), dtype=torch.uint8, device=dev)
    call("mopa_points_csr", ptr(pix), N, nrow, ptr(row_start), ptr(row_points), ptr(ws), ws.numel(), stream())
    feats = torch.randn(N, M, device=dev, generator=g)
    dfe, dl1, dl2 = torch.randn(N, M, device=dev, generator=g), torch.randn(N, C, device=dev, generator=g), torch.randn(N, C, device=dev, generator=g)
    dfeat = torch.empty(nrow, M, device=dev)
    dw1, db1, dw2, db2 = torch.zeros(C, M, device=dev), torch.zeros(C, device=dev), torch.zeros(C, M, device=dev), torch.zeros(C, device=dev)
    ws2 = torch.empty(max(query("mopa_output_layer_heads_bwd_workspace_bytes", N, M, C), 256), dtype=torch.uint8, device=dev)

    def heads_bwd():
        call("mopa_output_layer_heads_bwd", ptr(dfe), ptr(dl1), ptr(dl2), ptr(feats), ptr(w1), ptr(w2), ptr(row_start), ptr(row_points),
             nrow, N, M, C, ptr(dfeat), M, ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), 0, ptr(ws2), ws2.numel(), stream())
    t = timed(heads_bwd, reps)
    rows.append(("mopa_output_layer_heads_bwd (rows + 2 x wgrad)", t, (nrow * M + N * (2 * M + 4 * C) + nrow) * 4))

    # max-pool backward at the stem's shape: dx (B, 304, 480, 64) from dy (B, 152, 240, 64)
    OH, OW = Hp // 2, Wp // 2
    x = torch.randn(B * Hp * Wp, M, device=dev, generator=g)
    y = torch.empty(B * OH * OW, M, device=dev)
    amax = torch.empty(B * OH * OW * M, dtype=torch.uint8, device=dev)
    call("mopa_maxpool3x3s2_fwd", ptr(x), M, B, Hp, Wp, M, ptr(y), M, ptr(amax), stream())
    dy = torch.randn(B * OH * OW, M, device=dev, generator=g)
    dx = torch.empty(B * Hp * Wp, M, device=dev)

    def pool_fwd():
        call("mopa_maxpool3x3s2_fwd", ptr(x), M, B, Hp, Wp, M, ptr(y), M, ptr(amax), stream())

    def pool_bwd():
        call("mopa_maxpool3x3s2_bwd", ptr(dy), M, ptr(amax), B, Hp, Wp, M, ptr(dx), M, 0, stream())
    rows.append(("mopa_maxpool3x3s2_fwd", timed(pool_fwd, reps), B * (Hp * Wp + OH * OW) * M * 4 + B * OH * OW * M))
    rows.append(("mopa_maxpool3x3s2_bwd", timed(pool_bwd, reps), B * (Hp * Wp + OH * OW) * M * 4 + B * OH * OW * M))
    chk = float(pred.double().sum() + dfeat.double().sum() + dw1.double().sum() + dw2.double().sum() + dx.double().sum() + y.double().sum())
    for name, t, nbytes in rows:
        print(f"{name:50s} {t:9.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / t / 1e3:7.0f} GB/s")
    print(f"checksum {chk:.6e}")


if __name__ == "__main__":
    main()
