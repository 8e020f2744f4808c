#!/usr/bin/env python3
"""Benchmark of the MoPA hot path on MI355X (contract: see the task statement / DESIGN.md section "Measurement").

    python bench.py --gpus N --steps K --warmup W [--workload 3d|joint] [--batch B]

One "step" = one training pass of the hot path over one batch of B synthetic nuScenes-shape scans per GPU
(geometry build -> forward -> losses -> backward -> gradient all-reduce -> Adam).  Inputs are resident in HBM
before the timed region.  value = scans/s over all ranks (weak scaling: per-GPU batch fixed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
CLASS_WEIGHTS = [2.68678412, 4.36182969, 5.47896839, 3.89026883, 1.0]  # configs/nuscenes/usa_singapore yaml:54


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="3d", choices=["3d"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


class ConvTimer:
    """HIP-event brackets around every sparse-conv forward/backward-data launch (the dominant kernel family),
    recorded on the stream the kernel runs on (torch's current stream), plus its algorithmic byte count."""

    def __init__(self):
        self.records = []  # (start, end, bytes, flops)
        self.enabled = False
        self.rules = {}    # id(table) -> number of rules (set up front; tables are rebuilt with equal content)

    def install(self):
        from mopa_amd import sparse3d
        inner = sparse3d.spconv_fwd
        timer = self

        def wrapped(nbr, x, w, out, w_flip=False):
            if not timer.enabled:
                return inner(nbr, x, w, out, w_flip)
            K, A_out = nbr.shape
            R = timer.rules.get((K, A_out, x.rows))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            inner(nbr, x, w, out, w_flip)
            e.record()
            if R is not None:
                # gather R*Cin*4 + each output row once A*Cout*4 + int32 rule pair R*8 + weights
                nbytes = R * x.C * 4 + A_out * out.C * 4 + R * 8 + K * x.C * out.C * 4
                timer.records.append((s, e, nbytes, 2 * R * x.C * out.C))

        sparse3d.spconv_fwd = wrapped

    def summary(self):
        if not self.records:
            return None
        ms = sum(s.elapsed_time(e) for s, e, _, _ in self.records)
        nbytes = sum(r[2] for r in self.records)
        flops = sum(r[3] for r in self.records)
        n = len(self.records)
        return dict(launches=n, avg_us=1e3 * ms / n, bytes_per_launch=nbytes / n, gbs=nbytes / (ms * 1e-3) / 1e9,
                    tflops=flops / (ms * 1e-3) / 1e12)


def cpu_baseline_3d(model, seconds_budget=25.0):
    """The oracle (a CPU restatement, kind 'port') timed on this box's host cores: 3D branch fwd+bwd, one scan/step."""
    from mopa_amd import synth
    from oracle import scn3d
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 32)))  # more threads than that only adds contention on these op sizes
    pts = synth.lidar_points(12345)
    coords = np.concatenate([synth.voxelize(pts), np.zeros((len(pts), 1), np.int64)], 1)
    P = {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items()}
    for k, v in P.items():
        if "running" not in k:
            v.requires_grad_(True)
    feats = torch.ones(len(pts), 1)
    lab = torch.randint(0, 5, (len(pts),))
    def one_pass():
        t0 = time.time()
        geom = scn3d.Geometry(coords, 7)
        out = scn3d.net3dseg_forward(P, geom, feats, training=True)
        ce = torch.nn.functional.cross_entropy
        (ce(out["seg_logit"], lab) + ce(out["seg_logit2"], lab)).backward()
        return time.time() - t0

    warm = one_pass()
    n, t_total = 0, 0.0
    while n < 5 and t_total + warm < seconds_budget:
        t_total += one_pass()
        n += 1
    if n == 0:
        n, t_total = 1, warm
    return dict(value=n / t_total, unit="scans/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} x (1 synthetic 34,880-pt scan, Net3DSeg geometry+fwd+bwd, torch-CPU fp32 oracle)")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from mopa_amd import synth
    from mopa_amd.common.utils.loss import seg_ce
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_3d
    from mopa_amd.optim import FlatAdam
    from mopa_amd.sparse3d import Geometry3D

    torch.manual_seed(1 + rank)
    cfg = default_cfg(num_classes=5, dual_head=True)
    model3d, _ = build_model_3d(cfg)
    model3d = model3d.to(dev).train()
    if world > 1:  # identical initial weights on every rank
        for p in model3d.parameters():
            dist.broadcast(p.data, 0)
    opt3d = FlatAdam(model3d.parameters(), lr=1e-3)
    cw = torch.tensor(CLASS_WEIGHTS, device=dev)

    # ---- synthetic batches, resident in HBM before timing (two distinct batches, alternated)
    B = args.batch
    batches = []
    for j in range(2):
        scans = []
        for i in range(B):
            pts = synth.lidar_points(1000 * rank + j * B + i)
            rng = np.random.Generator(np.random.PCG64(99 + 1000 * rank + j * B + i))
            lab = rng.integers(0, 5, len(pts)).astype(np.int64)
            lab[rng.random(len(pts)) < 0.1] = -100
            scans.append((synth.voxelize(pts), lab))
        locs = torch.cat([torch.cat([torch.from_numpy(c), torch.full((len(c), 1), i, dtype=torch.int64)], 1)
                          for i, (c, _) in enumerate(scans)])
        batches.append(dict(locs=locs.to(dev), feats=torch.ones(locs.shape[0], 1, device=dev),
                            label=torch.cat([torch.from_numpy(l) for _, l in scans]).to(dev)))

    timer = ConvTimer()
    timer.install()
    for b in batches:  # rule counts for the algorithmic-bytes model (one-off, outside the timed region)
        g = Geometry3D(b["locs"], 7, 4096, dev)
        for l in range(7):
            timer.rules[(27, g.num_active[l], g.num_active[l])] = g.num_rules[l]
        for l in range(6):
            timer.rules[(8, g.num_active[l + 1], g.num_active[l])] = g.num_active[l]   # down / deconv-dgrad
            timer.rules[(8, g.num_active[l], g.num_active[l + 1])] = g.num_active[l]   # up / conv-dgrad
        del g

    def step(i):
        b = batches[i % 2]
        opt3d.zero_grad()
        geom = Geometry3D(b["locs"], 7, 4096, dev)
        out = model3d({"x": [b["locs"], b["feats"]], "geometry_3d": geom})
        loss = seg_ce(out["seg_logit"], b["label"], cw) + seg_ce(out["seg_logit2"], b["label"], cw)
        loss.backward()
        opt3d.all_reduce()
        opt3d.step(1.0 / world)
        return loss

    t_setup = time.perf_counter()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    print(f"[bench] rank {rank}: warmup {args.warmup} steps in {time.perf_counter() - t_setup:.2f}s", file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert torch.isfinite(loss).item(), "loss is not finite"

    if rank == 0:
        ks = timer.summary()
        roof = None
        if ks:
            roof = {"bound": "hbm", "achieved": round(ks["gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ks["gbs"] / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_spconv_fwd (fwd + bwd-data)",
                    "launches_per_step": ks["launches"] // args.steps, "avg_launch_us": round(ks["avg_us"], 2),
                    "algorithmic_bytes_per_launch": round(ks["bytes_per_launch"]), "mfma_tflops": round(ks["tflops"], 2)}
        line = {
            "metric": "scans/sec (joint 2D+3D train step) at 1/2/4/8 MI355X; sparse-conv HBM GB/s",
            "value": round(world * B * args.steps / elapsed, 3), "unit": "scans/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Net3DSeg SCN-UNet only (BASELINE configs[1]): geometry+fwd+CE+bwd+Adam, "
                                   f"bs={B} synthetic nuScenes-shape scans/GPU (34,880 pts each)",
                       "global_batch": B * world, "points_per_scan": 34880, "parallelism": f"dp{world}"},
            "roofline": roof,
        }
        print(f"[bench] timed {args.steps} steps in {elapsed:.3f}s", file=sys.stderr, flush=True)
        if not args.no_cpu_baseline:
            t_cpu = time.perf_counter()
            line["cpu_baseline"] = cpu_baseline_3d(model3d)
            print(f"[bench] cpu baseline took {time.perf_counter() - t_cpu:.1f}s", file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
