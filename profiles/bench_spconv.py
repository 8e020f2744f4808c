#!/usr/bin/env python3
"""Per-level timing of the sparse-conv forward kernels at the bench geometry (8 synthetic nuScenes-shape scans).

For every UNet level and the layer shapes the network uses there (C->C, 2C->C, and the K=8 down/up convs) this runs
the dense-table wave kernel (mopa_spconv_fwd) and the grouped-rulebook entry point (mopa_spconv_fwd_grouped: the
pipelined kernels on packed weights, the 4-wave block kernel on the shortest levels), checks that both give the same bits,
and prints us per launch, the time 40 % of the HBM roofline would take at SURVEY 8d's algorithmic bytes, and the fraction reached.

Usage: python profiles/bench_spconv.py [levels=7] [reps=20]     (MOPA_SPCONV_PATH=1|2 forces pipe|block kernel)
"""
import os
import sys
import torch

sys.path.insert(0, ".")
from mopa_amd import sparse3d as s3, synth  # noqa: E402
from mopa_amd._lib import call, ptr, stream  # noqa: E402


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
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    scans = int(sys.argv[3]) if len(sys.argv) > 3 else 8     # 16 = the joint step's merged source + target batch
    b = synth.make_batch(scans, H=16, W=16)
    g = s3.Geometry3D(b["x"][0], L, 4096, "cuda")
    torch.manual_seed(0)
    m = 16
    print(f"{'level':>5} {'table':>6} {'rows':>8} {'rules':>9} {'cin':>4} {'cout':>4} {'wave us':>9} {'grouped us':>10} "
          f"{'t40 us':>7} {'frac':>6} {'TF/s':>6} same  max|diff|  checksum(grouped) | run us (gemm + reduce)  TF/s  rel.err  run2==run")
    tot_t, tot_b, tot_r, tot_best = 0.0, 0.0, 0.0, 0.0
    bwd = os.environ.get("BENCH_BWD", "0") != "0"     # backward-data shapes: Cout -> Cin on the reversed table
    runs = {}

    def run_book(tab):
        key = tab.data_ptr()
        if key not in runs:
            K, Ao = tab.shape
            buf = torch.empty(s3.query("mopa_rulebook_runs_bytes", K, Ao) // 4, dtype=torch.int32, device="cuda")
            desc = torch.tensor([tab.data_ptr(), K, Ao, buf.data_ptr()], dtype=torch.int64)
            call("mopa_rulebook_runs_build_batched", desc.data_ptr(), 1, stream())
            runs[key] = buf
        return runs[key]
    for l in range(L):
        C = m * (l + 1)
        cases = [("subm", g.nbr27[l], C, C), ("subm", g.nbr27[l], 2 * C, C)]
        if l + 1 < L:
            cases.append(("down", g.ch[l], C, C + m))      # Convolution k2s2: out rows = level l+1
            cases.append(("up", g.up[l], C + m, C))        # Deconvolution k2s2: out rows = level l
        if bwd:   # the convolution backward-data runs: transposed weight on the reversed table
            cases = [(n, (t if n == "subm" else (g.up[l] if n == "down" else g.ch[l])), co, ci) for n, t, ci, co in cases]
        for name, tab, cin, cout in cases:
            K, Ao = tab.shape
            Ain = int(tab.max().item()) + 1
            rules = int((tab >= 0).sum().item())
            x = torch.randn(Ain, cin, device="cuda")
            w = torch.randn(K, cin, cout, device="cuda") * 0.1
            o1, o2 = s3.new_view(Ao, cout, "cuda"), s3.new_view(Ao, cout, "cuda")
            xv = s3.View(x)
            rb = g.rulebook(tab)
            gs, go, gi, gout = rb
            ws = torch.empty(max(1, s3.query("mopa_spconv_grouped_workspace_bytes", K, Ao, cout)), dtype=torch.uint8, device="cuda")

            def wave():
                call("mopa_spconv_fwd", ptr(tab), K, Ao, xv.p, xv.ld, cin, ptr(w), cout, 0, o1.p, o1.ld, stream())

            packed = s3.query("mopa_spconv_grouped_wants_packed", K, Ao, cin, cout)
            wp = torch.empty_like(w)
            if packed:
                call("mopa_spconv_pack_weight", ptr(w), K, cin, cout, 0, packed, ptr(wp), stream())

            def grouped():
                call("mopa_spconv_fwd_grouped", ptr(gs), ptr(go), ptr(gi), ptr(gout), K, Ao, xv.p, xv.ld, cin,
                     ptr(wp if packed else w), cout, 2 if packed else 0, o2.p, o2.ld, ptr(ws), ws.numel(), stream())

            tw = timed(wave, reps) if cin <= 192 else float('nan')   # the dense-table kernel stops at 192 input channels
            tg = timed(grouped, reps)
            same = torch.equal(o1.t, o2.t)
            err = float((o1.t - o2.t).abs().max())
            alg = rules * cin * 4 + Ao * cout * 4 + rules * 8 + K * cin * cout * 4     # SURVEY 8d algorithmic bytes
            t40 = alg / 3.2e6                                                           # us at 40 % of 8 TB/s
            bits = o2.t.contiguous().view(torch.int32).to(torch.int64)
            chk = int((bits * (torch.arange(bits.numel(), device="cuda").view_as(bits) % 8191 + 1)).sum().item()) & 0xffffffffffff
            tot_t += tg
            tot_b += alg
            line = (f"{l:>5} {name:>6} {Ao:>8} {rules:>9} {cin:>4} {cout:>4} {tw:>9.1f} {tg:>10.1f} {t40:>7.1f} {alg / tg / 8e6:>6.3f} "
                    f"{2 * rules * cin * cout / tg / 1e6:>6.1f} {same} {err:.2e} {chk:012x}")
            # offset-major path (sprun.hip): one rule per output row <=> the deconvolution-shaped table (every fine row has one parent)
            one = 1 if (K == 8 and rules == Ao) else 0
            tr = float("nan")
            if cin % 16 == 0 and cout % 16 == 0 and Ao * 8 * xv.ld * 4 < (1 << 32):
                rbuf = run_book(tab)
                wr = torch.empty_like(w)
                call("mopa_spconv_run_pack_weight", ptr(w), K, cin, cout, 0, ptr(wr), stream())
                wsr = torch.empty(max(256, 0 if one else s3.query("mopa_spconv_run_workspace_bytes", K, Ao, cout)), dtype=torch.uint8, device="cuda")
                o3 = s3.new_view(Ao, cout, "cuda")

                def run():
                    call("mopa_spconv_fwd_run", ptr(rbuf), K, Ao, xv.p, xv.ld, cin, ptr(wr), cout, 0, o3.p, o3.ld, one, ptr(wsr), wsr.numel(), stream())

                tr = timed(run, reps)
                first = o3.t.clone()
                o3.t.fill_(float("nan"))
                run()
                rel = float((o3.t - o1.t).abs().max() / o1.t.abs().max())
                line += f" | {tr:>8.1f} {2 * rules * cin * cout / tr / 1e6:>6.1f} {rel:.1e} {torch.equal(first, o3.t)}"
                tot_r += tr
            tot_best += min(tg, tr) if tr == tr else tg
            print(line, flush=True)
    print(f"forward family: {tot_t:.1f} us for {tot_b / 1e6:.1f} MB algorithmic = {tot_b / tot_t / 8e6:.3f} of 8 TB/s"
          f" | best of (grouped, run) per layer: {tot_best:.1f} us = {tot_b / tot_best / 8e6:.3f}")


if __name__ == "__main__":
    main()
