#!/usr/bin/env python3
"""Which algorithm runs each stride-1 3x3 convolution of the image branch, per batch and resolution -- measured, next to what the
dispatcher picks (round-4 review, item 6: the thresholds WINO4_DIRECT_MIN_TILES, WINOGRAD_F4_PIXELS, WINO4_FUSED_MIN_BLOCKS had been
tuned at ONE shape, 16 x 302 x 480).

For every (images per pass in {4, 8, 16} = 2+2 / 4+4 / 8+8 merged, resolution in {225 x 400 -> 240 x 400, 302 x 480 -> 304 x 480})
and every distinct stride-1 3x3 layer shape of UNetResNet34 (mopa/models/resnet34_unet.py:97-110,115-129) the forward
convolution is timed as
    direct     mopa_conv2d_igemm (f32-MFMA implicit GEMM)
    F2         Winograd F(2x2,3x3): input transform + 16 batched GEMMs + output transform
    F4         Winograd F(4x4,3x3): input transform + 36 batched GEMMs + output transform
    F4 fused   input transform + mopa_wino4_gemm_output (36 GEMMs with the output transform in the epilogue)
    F4 one     mopa_wino4_conv (input transform, GEMMs, output transform in one kernel; no V in HBM)
    F4 one9    mopa_wino4_conv9 (round 6: the same in its second form -- nine transform points per wave on 32x32x2 MFMAs, raw patches
               staged by LDS-DMA; csrc/wino4c9.hip)
and the dispatcher's choice for the roles is listed beside the fastest: `fwd` (a training forward pass; since round 6 the layers whose
weight gradient runs in one kernel from x and dY keep no V, and their training forward is dispatched like `fwd_eval` --
dense2d.forward_role), `fwd_eval` / `dgrad` (nothing is kept: the one-kernel form where it is eligible).  Writes
profiles/r6_algo_table.md and .json (the CPU test
tests/test_host_logic.py::test_dispatcher_follows_the_measured_algorithm_table asserts the dispatcher still makes these choices).
Usage: python profiles/algo_table.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d  # noqa: E402
from mopa_amd._lib import ptr  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def layer_shapes(Hp, Wp):
    """(name, cin, cout, H, W) of the distinct stride-1 3x3 convolutions (forward orientation) at a padded resolution."""
    h2, w2 = Hp // 2, Wp // 2
    return [("layer1 64->64", 64, 64, h2, w2), ("layer2 128->128", 128, 128, h2 // 2, w2 // 2), ("layer3 256->256", 256, 256, h2 // 4, w2 // 4),
            ("layer4 512->512", 512, 512, h2 // 8, w2 // 8), ("dec4 512->256", 512, 256, h2 // 4, w2 // 4), ("dec3 256->128", 256, 128, h2 // 2, w2 // 2),
            ("dec2 128->64", 128, 64, h2, w2), ("dec1 128->64", 128, 64, Hp, Wp)]


class patched:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: getattr(dense2d, k) for k in self.kw}
        for k, v in self.kw.items():
            setattr(dense2d, k, v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            setattr(dense2d, k, v)


def choice(cin, cout, B, H, W, role):
    """The dispatcher's pick under the shipped thresholds (mopa_amd/dense2d.py: wino_tile, wino4_direct, wino4_fused)."""
    F = dense2d.wino_tile(cin, cout, 3, 1, 1, B, H, W, "fwd" if role == "fwd_eval" else role)
    if F == 0:
        return "direct"
    if F == 2:
        return "F2"
    lay = dense2d.wino4_layout(cin, cout, B, H, W, role)
    return {3: "F4 one9", 2: "F4 one", 1: "F4 fused", 0: "F4"}[lay]


def training_fwd_choice(cin, cout, B, H, W):
    """The training forward pass: dispatched as "fwd_eval" where the weight gradient needs no V (dense2d.forward_role)."""
    return choice(cin, cout, B, H, W, dense2d.forward_role(cin, cout, 3, 1, 1, B, H, W, True)[1])


def main():
    rows = []
    for (H0, W0) in ((225, 400), (302, 480)):
        Hp, Wp = (H0 + 15) // 16 * 16, (W0 + 15) // 16 * 16
        for B in (4, 8, 16):
            shapes = []
            for name, cin, cout, H, W in layer_shapes(Hp, Wp):
                shapes.append((name, cin, cout, H, W, False))
                if cin != cout:   # backward-data runs the transposed channel counts
                    shapes.append((name.split()[0] + f" dgrad {cout}->{cin}", cout, cin, H, W, True))
            for name, cin, cout, H, W, is_dgrad in shapes:
                x = dense2d.new_img(B, H, W, cin, "cuda")
                x.t.normal_()
                out = dense2d.new_img(B, H, W, cout, "cuda")
                w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
                op = dense2d.ConvOp(w, None, 3, 1, 1)
                t = {}
                with patched(WINOGRAD=False):
                    t["direct"] = timed(lambda: op.forward(x, out))

                def wino(F, role="fwd_eval"):
                    lay = dense2d.wino4_layout(cin, cout, B, H, W, role) if F == 4 else 0
                    U = dense2d.wino_weight_cached(w, False, F, lay)
                    return timed(lambda: dense2d.wino_conv(x.p, x.ld, B, H, W, cin, cout, U, None, out.p, out.ld, F=F, role=role, want_v=False))

                t["F2"] = wino(2)
                with patched(WINO4_DIRECT=False, WINO4_FUSED_MIN_BLOCKS=1 << 62):
                    t["F4"] = wino(4)
                if cin % 64 == 0 and cout % 32 == 0 and (B * ((H + 3) // 4) * ((W + 3) // 4)) * cin * 4 < (1 << 32):
                    with patched(WINO4_DIRECT=False, WINO4_FUSED_MIN_BLOCKS=0):
                        t["F4 fused"] = wino(4)
                if cin % 64 == 0 and cout % 64 == 0 and cin <= dense2d.WINO4_DIRECT_MAX_CIN:
                    with patched(WINO4_DIRECT=True, WINO4_DIRECT_MIN_TILES=0, WINO4_DIRECT_ROLES=("fwd_eval", "dgrad", "fwd"), WINO4_CONV9=False):
                        t["F4 one"] = wino(4)
                    with patched(WINO4_DIRECT=True, WINO4_DIRECT_MIN_TILES=0, WINO4_DIRECT_ROLES=("fwd_eval", "dgrad", "fwd"), WINO4_CONV9=True):
                        t["F4 one9"] = wino(4)
                best = min(t, key=t.get)
                if is_dgrad:   # (this row IS the backward-data convolution)
                    ch = {"fwd": "-", "fwd_eval": "-", "dgrad": choice(cin, cout, B, H, W, "dgrad")}
                else:
                    ch = {"fwd": training_fwd_choice(cin, cout, B, H, W), "fwd_eval": choice(cin, cout, B, H, W, "fwd_eval")}
                    ch["dgrad"] = choice(cout, cin, B, H, W, "dgrad") if cin == cout else "(next row)"
                rows.append(dict(res=f"{H0}x{W0}", B=B, layer=name, cin=cin, cout=cout, H=H, W=W, us={k: round(v, 1) for k, v in t.items()}, best=best,
                                 chosen=ch, chosen_over_best={r: round(t[c] / t[best], 3) for r, c in ch.items() if c in t}))
                print(rows[-1], flush=True)
                del x, out
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)   # (what comes back from the GPU box; copy into profiles/ afterwards)
    for d in ("gpurun_out", "profiles"):
        json.dump(rows, open(os.path.join(ROOT, d, "r6_algo_table.json"), "w"), indent=1)
    cols = ["direct", "F2", "F4", "F4 fused", "F4 one", "F4 one9"]
    md = ["# Round 6: algorithm per stride-1 3x3 layer, batch and resolution -- measured (us per forward convolution) against the dispatcher's choice",
          "", "`python profiles/algo_table.py` on one MI355X, one stream.  **bold** = fastest; `fwd` = training forward (V kept only where the weight gradient is the two-operand form),",
          "`fwd_eval` = a forward pass that keeps nothing; rows named `dgrad a->b` are the backward-data convolutions of the layers with cin != cout (the",
          "transposed channel counts; a symmetric layer's backward-data is its own row).",
          "`x best` = the time of the dispatcher's pick over the fastest alternative's (1.00 = it picks the fastest).", "",
          "| resolution | images | layer | map | " + " | ".join(cols) + " | fwd picks | fwd_eval picks | dgrad picks | fwd x best | fwd_eval x best | dgrad x best |", "|---|---|---|---|" + "---|" * (len(cols) + 6)]
    for r in rows:
        cells = [("**%.1f**" % r["us"][c] if c == r["best"] else "%.1f" % r["us"][c]) if c in r["us"] else "-" for c in cols]
        md.append(f"| {r['res']} | {r['B']} | {r['layer']} | {r['H']}x{r['W']} | " + " | ".join(cells) +
                  f" | {r['chosen']['fwd']} | {r['chosen']['fwd_eval']} | {r['chosen']['dgrad']} | " +
                  " | ".join(("%.2f" % r["chosen_over_best"][k]) if k in r["chosen_over_best"] else "-" for k in ("fwd", "fwd_eval", "dgrad")) + " |")
    worst = max(max(r["chosen_over_best"].values()) for r in rows)
    md += ["", f"Worst pick over the {len(rows)} rows: {worst:.2f} x the fastest alternative."]
    for d in ("gpurun_out", "profiles"):
        open(os.path.join(ROOT, d, "r6_algo_table.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md[-3:]))


if __name__ == "__main__":
    main()
