#!/usr/bin/env python3
"""Per-layer A/B of the sparse-conv kernels at the bench geometry (8 synthetic nuScenes-shape scans): the grouped-rulebook
kernels of spconv.hip (k_spconv_t4 / _pipe / _blk on 64-row tiles) against the column-slice kernel of spconv_cs.hip on
128- and 256-row tiles, forward shapes and the backward-data shapes of the decoder convolutions (Cin = P, Cout = 2P).

Prints us per launch, the fraction of the HBM roofline at SURVEY 8d's algorithmic bytes, and the max deviation of the
column-slice result from the dense-table kernel (summation order differs: rounding only).

Usage: python profiles/bench_spconv_cs.py [levels=7] [reps=20]
"""
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
    b = synth.make_batch(8, H=16, W=16)
    g = s3.Geometry3D(b["x"][0], L, 4096, "cuda")
    torch.manual_seed(0)
    m = 16
    VARIANTS = ((128, 1), (128, 2), (256, 1), (256, 2))       # (tile rows, 16-column tiles per wave)
    print(f"{'level':>5} {'table':>6} {'rows':>8} {'rules':>9} {'cin':>4} {'cout':>4} {'grouped':>8} " +
          " ".join(f"{'cw%d/%d' % v:>8}" for v in VARIANTS) + f" {'t40 us':>7} {'f_grp':>6} {'f_best':>6} {'maxerr':>9}")
    tot = {"grp": 0.0, "best": 0.0, "t40": 0.0}
    for l in range(L):
        C = m * (l + 1)
        cases = [("subm", g.nbr27[l], C, C), ("subm", g.nbr27[l], 2 * C, C)]
        if l + 1 < L:
            cases.append(("bwd", g.nbr27[l], C, 2 * C))   # backward-data of the decoder block's 2C -> C convolution
            cases.append(("down", g.ch[l], C, C + m))
            cases.append(("up", g.up[l], C + m, C))
        for name, tab, cin, cout in cases:
            K, Ao = tab.shape
            Ain = int(tab.max().item()) + 1
            rules = int((tab >= 0).sum().item())
            x = torch.randn(Ain, cin, device="cuda")
            w = torch.randn(K, cin, cout, device="cuda") * 0.1
            o0, o1, o2 = (s3.new_view(Ao, cout, "cuda") for _ in range(3))
            xv = s3.View(x)
            gs, go, gi, gout = g.rulebook(tab)
            ws = torch.empty(max(1, query("mopa_spconv_grouped_workspace_bytes", K, Ao, cout)), dtype=torch.uint8, device="cuda")
            packed = query("mopa_spconv_grouped_wants_packed", K, Ao, cin, cout)
            wp = torch.empty_like(w)
            if packed:
                call("mopa_spconv_pack_weight", ptr(w), K, cin, cout, 0, packed, ptr(wp), stream())

            def grouped():
                call("mopa_spconv_fwd_grouped", ptr(gs), ptr(go), ptr(gi), ptr(gout), K, Ao, xv.p, xv.ld, cin,
                     ptr(wp if packed else w), cout, 2 if packed else 0, o0.p, o0.ld, ptr(ws), ws.numel(), stream())

            tg = timed(grouped, reps)
            ref = torch.empty(Ao, cout, device="cuda")
            if cin <= 192:
                call("mopa_spconv_fwd", ptr(tab), K, Ao, xv.p, xv.ld, cin, ptr(w), cout, 0, ptr(ref), cout, stream())
            else:
                ref.copy_(o0.t)
            scale = float(ref.abs().max())
            res = {}
            w1 = torch.empty_like(w)
            call("mopa_spconv_pack_weight", ptr(w), K, cin, cout, 0, 1, ptr(w1), stream())
            for TM, ntw in VARIANTS:
                o = o1
                if not query("mopa_spconv_cs_supported", cin, cout) or (cout // 16) % ntw:
                    res[(TM, ntw)] = (float("nan"), 0.0)
                    continue
                cgs, cgo, cgi, cgout = g.rulebook_cs(tab, TM)

                def cs():
                    call("mopa_spconv_fwd_cs", ptr(cgs), ptr(cgo), ptr(cgi), ptr(cgout), K, Ao, TM, xv.p, xv.ld, cin, ptr(w1), cout, ntw << 8,
                         o.p, o.ld, stream())

                try:
                    o.t.fill_(float("nan"))
                    t = timed(cs, reps)
                    err = float((o.t - ref).abs().max()) / scale
                    if err != err:
                        err = float("inf")
                except RuntimeError:
                    t, err = float("nan"), 0.0
                res[(TM, ntw)] = (t, err)
            alg = rules * cin * 4 + Ao * cout * 4 + rules * 8 + K * cin * cout * 4
            t40 = alg / 3.2e6
            f = lambda t: alg / t / 8e6
            ts = [res[v][0] for v in VARIANTS]
            best = min(x for x in [tg] + ts if x == x)
            tot["grp"] += tg; tot["best"] += best; tot["t40"] += t40
            print(f"{l:>5} {name:>6} {Ao:>8} {rules:>9} {cin:>4} {cout:>4} {tg:>8.1f} " + " ".join(f"{t:>8.1f}" for t in ts) +
                  f" {t40:>7.1f} {f(tg):>6.3f} {f(best):>6.3f} {max(res[v][1] for v in VARIANTS):>9.2e}", flush=True)
    print("totals us: grouped %.1f  best-of %.1f  t40 %.1f  -> frac grouped %.3f best %.3f" % (
        tot["grp"], tot["best"], tot["t40"], 0.4 * tot["t40"] / tot["grp"], 0.4 * tot["t40"] / tot["best"]))


if __name__ == "__main__":
    main()
