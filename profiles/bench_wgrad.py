#!/usr/bin/env python3
"""Per-level timing of the sparse weight-gradient entry point (mopa_spconv_bwd_weight = k_spconv_wgrad2 + k_reduce_slabs) at the bench
geometry (8 synthetic nuScenes-shape scans by default), every layer shape the network uses; us per launch, the algorithmic-bytes roofline
fraction (R (Cin + Cout) 4 + R 8 + K Cin Cout 4) and the MFMA rate -- and, beside it, the same gradient on the run-major
rulebook (mopa_spconv_bwd_weight_run = k_wgrad_run + k_wgrad_run_reduce, csrc/sprun.hip; the stride-2 convolution's on its deconvolution
table with the index lists swapped), its rate, the ratio, the largest difference between the two relative to the gradient's scale, and
whether the dispatcher (mopa_spconv_wgrad_run_wanted) picks it.  Usage: python profiles/bench_wgrad.py [levels=7] [reps=20] [scans=8]"""
import sys
import torch

sys.path.insert(0, ".")
from mopa_amd import sparse3d as s3, synth  # noqa: E402
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
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    scans = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    b = synth.make_batch(scans, H=16, W=16)
    g = s3.Geometry3D(b["x"][0], L, 4096, "cuda")
    m = 16
    print(f"{'level':>5} {'table':>6} {'rows':>8} {'rules':>9} {'cin':>4} {'cout':>4} {'us':>8} {'alg MB':>8} {'frac':>6} {'TF/s':>6} {'slab MB':>8}"
          f" | {'run us':>8} {'TF/s':>6} {'ratio':>6} {'rel diff':>9} {'picked':>6}")
    tot_t = tot_b = tot_r = tot_pick = 0.0
    runs = {}

    def run_book(tab):
        key = tab.data_ptr()
        if key not in runs:
            K, Ao = tab.shape
            buf = torch.empty(query("mopa_rulebook_runs_bytes", K, Ao) // 4, dtype=torch.int32, device="cuda")
            desc = torch.tensor([tab.data_ptr(), K, Ao, buf.data_ptr()], dtype=torch.int64)
            call("mopa_rulebook_runs_build_batched", desc.data_ptr(), 1, stream())
            runs[key] = buf
        return runs[key]
    for l in range(L):
        C = m * (l + 1)
        cases = [("subm", g.nbr27[l], C, C), ("subm", g.nbr27[l], 2 * C, C)]
        if l == 0:
            cases.insert(0, ("stem", g.nbr27[0], 1, 16))
        if l + 1 < L:
            cases += [("down", g.ch[l], C, C + m), ("up", g.up[l], C + m, C)]
        for name, tab, cin, cout in cases:
            K, Ao = tab.shape
            Ain = int(tab.max().item()) + 1
            rules = int((tab >= 0).sum().item())
            x, dy = torch.randn(Ain, cin, device="cuda"), torch.randn(Ao, cout, device="cuda")
            dw = torch.zeros(K, cin, cout, device="cuda")
            wsb = query("mopa_spconv_wgrad_workspace_bytes", K, Ao, cin, cout)
            ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device="cuda")

            def run():
                call("mopa_spconv_bwd_weight", ptr(tab), K, Ao, ptr(x), cin, cin, ptr(dy), cout, cout, ptr(dw), 0, ptr(ws), ws.numel(), stream())

            t = timed(run, reps)
            alg = rules * (cin + cout) * 4 + rules * 8 + K * cin * cout * 4
            tot_t += t; tot_b += alg
            line = (f"{l:>5} {name:>6} {Ao:>8} {rules:>9} {cin:>4} {cout:>4} {t:>8.1f} {alg / 1e6:>8.1f} {alg / t / 8e6:>6.3f} "
                    f"{2 * rules * cin * cout / t / 1e6:>6.1f} {wsb / 1e6:>8.1f}")
            # the run-list kernel: the table's own rulebook, or (stride-2 convolution) the deconvolution table's with the lists swapped
            if name == "down":
                rtab, one, swap = g.up[l], 1, 1
            else:
                rtab, one, swap = tab, int(name == "up"), 0
            Kr, Ar = rtab.shape
            tr = None
            if cin % 16 == 0 and query("mopa_spconv_wgrad_run_workspace_bytes", Kr, Ar, cin, cout, one) > 0:
                rb = run_book(rtab)
                wsr = torch.empty(query("mopa_spconv_wgrad_run_workspace_bytes", Kr, Ar, cin, cout, one), dtype=torch.uint8, device="cuda")
                dw2 = torch.zeros(K, cin, cout, device="cuda")

                def run2():
                    call("mopa_spconv_bwd_weight_run", ptr(rb), Kr, Ar, one, swap, ptr(x), cin, cin, ptr(dy), cout, cout, ptr(dw2), 0, ptr(wsr), wsr.numel(), stream())

                tr = timed(run2, reps)
                err = float((dw - dw2).abs().max()) / (float(dw.abs().max()) + 1e-30)
                pick = query("mopa_spconv_wgrad_run_wanted", Kr, Ar, cin, cout, one)
                line += f" | {tr:>8.1f} {2 * rules * cin * cout / tr / 1e6:>6.1f} {tr / t:>6.2f} {err:>9.1e} {'run *' if pick else '':>6}"
                tot_pick += tr if pick else t
            else:
                tot_pick += t
            tot_r += tr if tr is not None else t
            print(line, flush=True)
    print("total us %.1f  alg MB %.1f  -> frac %.3f | run everywhere it exists %.1f us, dispatcher %.1f us" % (tot_t, tot_b / 1e6, tot_b / tot_t / 8e6, tot_r, tot_pick))


if __name__ == "__main__":
    main()
