#!/usr/bin/env python3
"""Benchmark of the MoPA hot path on MI355X (contract: see the task statement / DESIGN.md section "Measurement").

    python bench.py --gpus N --steps K --warmup W [--workload 3d|joint|mopa] [--batch B]

`--gpus N` with N > 1, started as a plain process (no RANK / WORLD_SIZE in the environment), launches N ranks itself:
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`, one rank
per GPU over RCCL; rank 0 prints the one JSON line.  Started by a launcher already (the driver's torchrun form), it is a rank.

One "step" = one training pass of the hot path over one batch of B synthetic nuScenes-shape scans per GPU
(geometry build -> forward -> losses -> backward -> gradient all-reduce -> Adam).  Inputs are resident in HBM
before the timed region.  value = scans/s over all ranks (weak scaling: per-GPU batch fixed).

A/B switches (environment; the defaults are what the numbers in DESIGN.md were measured with):
  MOPA_BENCH_EVENT_STRIDE=5   HIP-event brackets (roofline figures) on every n-th timed step; 0 = none, 1 = every step
  MOPA_BENCH_STEP_TIMES=1     per-step host clock + caching-allocator counters + cyclic-GC passes of the timed region on stderr
  MOPA_BENCH_TIMELINE=1       HIP-event marks on the streams (ends of the forwards / losses / backwards of both halves), mean offsets on stderr
  MOPA_BENCH_GC_FREEZE=0      skip mopa_amd.step.freeze_host_heap() after the warm-up (then one 70-100 ms full GC pass lands in
                              the timed region: -20 % on the launch-bound 3D-only workload, nothing on the joint one)
  MOPA_BENCH_GEOM_AHEAD=0     voxel geometry built behind the main stream again;  MOPA_BENCH_REORDER=0  3D forward enqueued
                              first, 3D losses on the main stream;  MOPA_BENCH_BWD3_FIRST=0  (with REORDER=0) 2D backward first
  MOPA_BENCH_NO_SIDE=1        3D branch on the main stream;  MOPA_WGRAD_STREAM=0  2D weight gradients on the main stream
  MOPA_CONV2D_MFMA=0          fp32 vector-pipe conv kernels;  MOPA_WINOGRAD=0 / MOPA_WINOGRAD_F4=0 / MOPA_WINOGRAD_F4_ROLES=dgrad,wgrad (exact-product forward)
  MOPA_BENCH_BACKEND=gloo     lets several ranks share one GPU (RCCL refuses that): plumbing test only
  MOPA_BENCH_DRY=1            launcher / process-group plumbing only (no GPU work): every rank joins the group, exchanges its
                              scan seeds and point counts, rank 0 prints a JSON line with n_gpus = world (CPU test of --gpus N)
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
F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector == FP32 matrix (dense)
CLASS_WEIGHTS = [2.68678412, 4.36182969, 5.47896839, 3.89026883, 1.0]  # configs/nuscenes/usa_singapore yaml:54
CLASS_WEIGHTS_KITTI = [1.89090012, 2.0585112, 3.1970535, 3.1111633, 1., 2.93751704, 1.92053733,
                       1.47886874, 1.04654198, 1.78266561]              # configs/a2d2_semantic_kitti/xmuda.yaml:42-43


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="joint", choices=["joint", "3d", "mopa", "kitti"],
                    help="joint = BASELINE configs[2] (default), 3d = configs[1], mopa = configs[3] per-GPU step, "
                         "kitti = configs[4] per-GPU step (A2D2->SemanticKITTI shape: 120,000-pt scans, 10 classes, joint 2D+3D)")
    ap.add_argument("--batch", type=int, default=None, help="scans per domain per GPU (default 8; 4 for mopa; 2 for kitti)")
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
        inner = sparse3d.spconv_launch
        timer = self

        def wrapped(nbr, x, w, out, w_flip, rb, packed):
            if not timer.enabled:
                return inner(nbr, x, w, out, w_flip, rb, packed)
            K, A_out = nbr.shape
            R = timer.rules.get((K, A_out, x.rows))
            if R is None:   # a geometry built inside the step (the VGI batch of the mopa workload): count its rules once
                R = max(A_out, x.rows) if K == 8 else int((nbr >= 0).sum().item())   # K = 8: one rule per fine row
                timer.rules[(K, A_out, x.rows)] = R
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            inner(nbr, x, w, out, w_flip, rb, packed)
            e.record()
            if R is not None:
                # gather R*Cin*4 + each output row once A*Cout*4 + int32 rule pair R*8 + weights
                nbytes = R * x.C * 4 + A_out * out.C * 4 + R * 8 + K * x.C * out.C * 4
                timer.records.append((s, e, nbytes, 2 * R * x.C * out.C))

        sparse3d.spconv_launch = wrapped
        inner_run = sparse3d.spconv_launch_run

        def wrapped_run(runs, K, x, w, out, w_flip):   # the offset-major path of the same family (gather-GEMM + ordered reduce: csrc/sprun.hip)
            if not timer.enabled:
                return inner_run(runs, K, x, w, out, w_flip)
            A_out = out.rows
            R = timer.rules.get((K, A_out, x.rows))
            if R is None:
                R = max(A_out, x.rows) if K == 8 else int(runs[0][32:32 + K].sum().item())   # (the rulebook's header holds the rule counts)
                timer.rules[(K, A_out, x.rows)] = R
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            inner_run(runs, K, x, w, out, w_flip)
            e.record()
            nbytes = R * x.C * 4 + A_out * out.C * 4 + R * 8 + K * x.C * out.C * 4   # the same algorithmic bytes: the slab is not work
            timer.records.append((s, e, nbytes, 2 * R * x.C * out.C))

        sparse3d.spconv_launch_run = wrapped_run

    def summary(self):
        if not self.records:
            return None
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        nbytes = sum(r[2] for r in self.records)
        flops = sum(r[3] for r in self.records)
        direct = sum(r[4] if len(r) > 4 else r[3] for r in self.records)   # flops of the direct convolutions the launches stand for
        n = len(self.records)
        # per launch the tighter of the two ceilings: its algorithmic bytes at the HBM peak or its flops at the dense fp32 MFMA peak
        ideal_us = sum(max(r[2] / (HBM_PEAK_GBS * 1e9), r[3] / (F32_PEAK_TFLOPS * 1e12)) for r in self.records) * 1e6
        mfma_bound = sum(1 for r in self.records if r[3] / (F32_PEAK_TFLOPS * 1e12) > r[2] / (HBM_PEAK_GBS * 1e9))
        return dict(launches=n, avg_us=1e3 * ms / n, bytes_per_launch=nbytes / n, gbs=nbytes / (ms * 1e-3) / 1e9,
                    tflops=flops / (ms * 1e-3) / 1e12, tflops_direct=direct / (ms * 1e-3) / 1e12,
                    ideal_us_per_launch=ideal_us / n, mfma_bound_launches=mfma_bound)


class Conv2dTimer:
    """HIP-event brackets around every launch of the dense implicit-GEMM kernel in the 2D branch (fwd, dgrad, convT classes,
    the batched GEMMs of the Winograd layers)."""

    def __init__(self):
        self.records = []
        self.wgrad_records = []
        self.enabled = False

    def install(self):
        from mopa_amd import dense2d
        inner = dense2d.igemm
        timer = self

        def wrapped(x_ptr, w, bias, out_ptr, geom, accumulate=False):
            if not timer.enabled:
                return inner(x_ptr, w, bias, out_ptr, geom, accumulate)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            inner(x_ptr, w, bias, out_ptr, geom, accumulate)
            e.record()
            g = list(geom)
            M = g[0] * g[3] * g[4]  # B * OHl * OWl
            taps, cin, cout = g[15] * g[16], g[21], g[22]
            # bytes: activations read once + outputs written once + weights once (ideal reuse)
            timer.records.append((s, e, 4 * (M * cin + M * cout + taps * cin * cout), 2 * M * cout * taps * cin, 2 * M * cout * taps * cin))

        dense2d.igemm = wrapped
        inner_b = dense2d.igemm_batched

        def wrapped_b(x_ptr, w_ptr, out_ptr, geom, nbatch, in_stride, w_stride, out_stride):   # the 16 GEMMs of a Winograd conv
            if not timer.enabled:
                return inner_b(x_ptr, w_ptr, out_ptr, geom, nbatch, in_stride, w_stride, out_stride)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            inner_b(x_ptr, w_ptr, out_ptr, geom, nbatch, in_stride, w_stride, out_stride)
            e.record()
            g = list(geom)
            M, cin, cout = g[0] * g[3] * g[4], g[21], g[22]
            # (the direct 3x3 convolution these GEMMs replace: 16 or 36 points stand for 4 or 16 output pixels x 9 taps)
            timer.records.append((s, e, 4 * nbatch * (M * cin + M * cout + cin * cout), 2 * nbatch * M * cout * cin,
                                  2 * M * cout * cin * 9 * (4 if nbatch == 16 else 16 if nbatch == 36 else nbatch / 9)))

        dense2d.igemm_batched = wrapped_b
        inner_call = dense2d.call

        WGRAD = ("mopa_conv2d_bwd_weight", "mopa_stem_bwd_weight_bn", "mopa_wino4_bwd_weight", "mopa_wino_bwd_weight", "mopa_wino4_wgrad_fused")

        def wrapped_call(name, *a):   # the fused F(4x4) GEMM + output-transform kernel belongs to the same family
            if timer.enabled and name in WGRAD:
                # the weight-gradient family (roofline_wgrad): one entry point = one weight gradient = its MFMA kernel + the ordered
                # slab reduction (k_conv2d_wgrad_mfma / k_wino4_wgrad / k_stem_wgrad_mfma + k_reduce_slabs2 / k_wino4_dw), flops as
                # executed; the brackets sit on the stream the call runs on (the weight-gradient stream: torch's current stream there)
                import ctypes
                if name in ("mopa_conv2d_bwd_weight", "mopa_stem_bwd_weight_bn"):
                    g = (ctypes.c_int32 * 25).from_address(int(a[3] if name == "mopa_conv2d_bwd_weight" else a[10]))
                    fl = 2 * g[0] * g[3] * g[4] * g[15] * g[16] * g[21] * g[22]
                elif name == "mopa_wino4_wgrad_fused":
                    B_, H_, W_, ci_, co_ = a[7], a[8], a[9], a[10], a[11]
                    fl = 2 * 36 * B_ * ((H_ + 3) // 4) * ((W_ + 3) // 4) * ci_ * co_
                else:
                    fl = 2 * (36 if name == "mopa_wino4_bwd_weight" else 16) * a[2] * a[3] * a[4]
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = inner_call(name, *a)
                e.record()
                timer.wgrad_records.append((s, e, 0, fl, fl))
                return r
            if not timer.enabled or name not in ("mopa_wino4_gemm_output", "mopa_wino4_conv", "mopa_wino4_conv9"):
                return inner_call(name, *a)
            one = name != "mopa_wino4_gemm_output"   # the one-kernel convolutions: input transform + GEMMs + output transform (no V)
            B, H, W, cin, cout = (a[6], a[7], a[8], a[9], a[10]) if one else (a[5], a[6], a[7], a[8], a[9])
            T = B * ((H + 3) // 4) * ((W + 3) // 4)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = inner_call(name, *a)
            e.record()
            in_bytes = B * H * W * cin if one else 36 * T * cin
            timer.records.append((s, e, 4 * (in_bytes + B * H * W * cout + 36 * cin * cout), 2 * 36 * T * cin * cout, 2 * 16 * T * cin * cout * 9))
            return r

        dense2d.call = wrapped_call

    summary = ConvTimer.summary

    def wgrad_summary(self):
        keep, self.records = self.records, self.wgrad_records
        try:
            return ConvTimer.summary(self)
        finally:
            self.records = keep


def gradient_error_vs_fp64():
    """config.gradient_error_vs_fp64: what Winograd F(4x4) in all three passes costs in gradient accuracy at the bench shape -- the
    committed output of profiles/f4_gradient_noise.py (max |g - fp64| / max |fp64| per parameter tensor; median / p90 / max over the
    tensors), beside the exact-product kernels' own fp32-vs-fp64 noise.  Not measured in this run."""
    path = os.path.join(ROOT, "profiles", "r6_f4_gradient_noise.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    d["source"] = os.path.relpath(path, ROOT)
    return d


def f4_roles():
    from mopa_amd import dense2d
    return tuple(dense2d.F4_ROLES)


def one_kernel_roles():
    """Passes whose eligible F(4x4) layers run as one kernel (mopa_wino4_conv: no V / M in HBM; "fwd": the training forward pass of
    the layers whose weight gradient runs from x and dY -- mopa_wino4_wgrad_fused -- and so needs no V; "wgrad": that kernel)."""
    from mopa_amd import dense2d
    r = tuple(dense2d.WINO4_DIRECT_ROLES) if dense2d.WINO4_DIRECT else ()
    if dense2d.WINO4_WGRAD_FUSED:
        r += (("fwd",) if dense2d.WINO4_DIRECT and "fwd_eval" in r else ()) + ("wgrad",)
    return r


def dense2d_streams():
    """True when the 2D weight gradients run on their own stream in this process (mopa_amd/dense2d.py::wgrad_stream)."""
    from mopa_amd import dense2d
    return dense2d.wgrad_stream(torch.cuda.current_device()) is not None


def _cpu_passes(one_pass, seconds_budget, max_passes, what):
    """Bounded CPU sample: up to 3 warm-up passes (fewer only if one pass alone eats a third of the time
    budget), then up to `max_passes` timed passes inside the budget; value = 1 / median pass time (min and count in `sample`)."""
    warm = one_pass()
    n_warm = 1
    while n_warm < 3 and (n_warm + 1) * warm < seconds_budget / 3:
        warm = one_pass()
        n_warm += 1
    times = []
    while len(times) < max_passes and sum(times) + n_warm * warm < seconds_budget:
        times.append(one_pass())
    if not times:
        times = [warm]
    med = float(np.median(times))
    return dict(value=1.0 / med, unit="scans/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(times)} timed passes after {n_warm} warm-up pass(es) within a {seconds_budget:.0f} s budget, each = {what}; "
                       f"median {med:.2f} s, min {min(times):.2f} s per scan")


def cpu_baseline_joint(model2d, model3d, seconds_budget=25.0, shape=None):
    """CPU restatement (oracle, kind 'port') of one scan of the joint step: 2D + 3D forward/backward + CE/KL losses."""
    from mopa_amd import synth
    from oracle import losses as ol
    from oracle import net2d, scn3d
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 32)))
    shape = shape or synth.NUSCENES
    s = synth.make_scan(4242, shape=shape)
    coords = np.concatenate([s["coords"], np.zeros((len(s["coords"]), 1), np.int64)], 1)
    P2 = {k: v.detach().cpu().clone() for k, v in model2d.state_dict().items()}
    P3 = {k: v.detach().cpu().float().clone() for k, v in scn3d.fold_state_dict(model3d.state_dict()).items()}
    for P in (P2, P3):
        for k, v in P.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)
    img = torch.from_numpy(s["img"])[None]
    lab = torch.from_numpy(s["seg_label"])
    feats = torch.ones(len(coords), 1)

    def one_pass():
        t0 = time.time()
        o2 = net2d.net2dseg_forward(P2, img, [s["img_indices"]], training=True, dropout_p=0.4)
        o3 = scn3d.net3dseg_forward(P3, scn3d.Geometry(coords, 7), feats, training=True)
        l2 = ol.seg_ce(o2["seg_logit"], lab) + ol.xm_kl(o2["seg_logit2"], o3["seg_logit"])
        l3 = ol.seg_ce(o3["seg_logit"], lab) + ol.xm_kl(o3["seg_logit2"], o2["seg_logit"])
        l2.backward()
        l3.backward()
        return time.time() - t0

    return _cpu_passes(one_pass, seconds_budget, 10,
                       f"1 synthetic scan: 302x480 image + {len(coords):,} pts, Net2DSeg+Net3DSeg fwd+bwd+CE+KL, torch-CPU fp32 oracle")


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
    P = {k: v.detach().cpu().float().clone() for k, v in scn3d.fold_state_dict(model.state_dict()).items()}
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

    return _cpu_passes(one_pass, seconds_budget, 10, "1 synthetic 34,880-pt scan, Net3DSeg geometry+fwd+bwd, torch-CPU fp32 oracle")


def launch_ranks(args) -> int:
    """--gpus N from a plain process: start N ranks (one per GPU) under torch.distributed.run and wait for them.
    Nothing in this process has touched the GPU (no HIP call is made by importing torch), so the children own the devices."""
    import socket
    import subprocess
    # preflight (device_count() does not initialise the GPU): RCCL wants one device per rank, and a launcher that starts more ranks
    # than there are devices fails late and obscurely inside init_process_group
    if os.environ.get("MOPA_BENCH_DRY") != "1" and os.environ.get("MOPA_BENCH_BACKEND", "nccl") == "nccl":
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            print(f"[bench] --gpus {args.gpus} needs {args.gpus} visible devices, this node has {ndev} "
                  "(set MOPA_BENCH_BACKEND=gloo to run several ranks on one device as a plumbing test)", file=sys.stderr, flush=True)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this platform (RCCL / xGMI peer mappings)
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """MOPA_BENCH_DRY=1: the multi-process plumbing of `--gpus N` without a GPU (gloo): group setup, per-rank scan seeds,
    the point-count exchange behind the global loss mean, barrier + max-over-ranks timing, one JSON line from rank 0."""
    from mopa_amd.step import global_mean_weight
    if world > 1:
        dist.init_process_group(os.environ.get("MOPA_BENCH_BACKEND", "gloo"))
    B = args.batch or 8
    seeds = [1000 * rank + i for i in range(2 * B)]
    n_local = 34880 * B
    w = global_mean_weight(n_local)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    allseeds = [None] * world
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_gather_object(allseeds, seeds)
    else:
        allseeds = [seeds]
    if rank == 0:
        flat = [s for r in allseeds for s in r]
        print(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "unit": "scans/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "scaling": "weak", "dry": True, "distinct_scans": len(set(flat)) == len(flat),
                          "loss_weight_rank0": w, "config": {"parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # MOPA_FORCE_COLLECTIVES=1: run the process group, the gradient all-reduces (RCCL, the 3D one asynchronously on the side
    # stream) and the data-parallel stream configuration even with ONE rank -- the only way to exercise the N-GPU code path
    # on a 1-GPU box with the real backend (python -m torch.distributed.run --nproc-per-node 1 bench.py)
    multi = world > 1 or os.environ.get("MOPA_FORCE_COLLECTIVES") == "1"
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus = {world}", file=sys.stderr)
    if os.environ.get("MOPA_BENCH_DRY") == "1":
        return dry_run(args, rank, world)
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)  # (only differs from LOCAL_RANK in the single-GPU plumbing test below)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI by default; MOPA_BENCH_BACKEND=gloo lets two ranks share one GPU to test the multi-process
        # plumbing on a 1-GPU box (RCCL refuses two ranks on one device).
        backend = os.environ.get("MOPA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mopa_amd import synth
    from mopa_amd.common.utils.loss import mask_cons_loss, seg_ce, softmax_lastdim, xm_kl
    from mopa_amd.config import default_cfg
    from mopa_amd.models.build import build_model_2d, build_model_3d
    from mopa_amd import vgi
    from mopa_amd.optim import FlatAdam
    from mopa_amd.sparse3d import Geometry3D

    joint = args.workload in ("joint", "mopa", "kitti")
    mopa = args.workload == "mopa"
    kitti = args.workload == "kitti"
    if args.batch is None:
        args.batch = 4 if mopa else 2 if kitti else 8
    shape = synth.KITTI if kitti else synth.NUSCENES     # 64 beams x 1875 azimuths = 120,000 pts vs 32 x 1090 = 34,880
    NC = shape["classes"]
    lam_src, lam_trg = (0.1, 0.01) if kitti else (1.0, 0.1)   # lambda_xm_src / _trg: a2d2_semantic_kitti yaml :54-55, nuscenes :56-57
    torch.manual_seed(1 + rank)
    cfg = default_cfg(num_classes=NC, dual_head=True)
    model3d, _ = build_model_3d(cfg)
    model3d = model3d.to(dev).train()
    models = [model3d]
    if joint:
        model2d, _ = build_model_2d(cfg)
        model2d = model2d.to(dev).train()
        models.append(model2d)
    if multi:  # identical initial weights on every rank
        for m in models:
            for p in m.parameters():
                dist.broadcast(p.data, 0)
    opts = [FlatAdam(m.parameters(), lr=1e-3) for m in models]
    # 2D network under a process group: the 94.5 MB flat gradient goes out as 4 contiguous buckets in backward order, each all-reduced
    # on a communication stream from the point of the backward pass where its last gradient kernel is enqueued (decoder + heads and
    # layer4 are final ~20 ms before the stem).  MOPA_BENCH_BUCKETS=0: one collective behind the backward pass.
    n_buckets = int(os.environ.get("MOPA_BENCH_BUCKETS", "4"))
    bucket_bytes = None
    if joint and multi and n_buckets > 1:
        from mopa_amd import dense2d as _d2
        bucket_bytes = opts[1].enable_buckets(n_buckets, extra_streams=lambda: [_d2.wgrad_stream(dev)])
    cw = torch.tensor(CLASS_WEIGHTS_KITTI if kitti else CLASS_WEIGHTS, device=dev)
    H, W = 302, 480

    # ---- synthetic batches, resident in HBM before timing.  joint: [source, target] of one xMUDA iteration;
    #      3d: two batches alternated.
    B = args.batch
    batches = []
    for j in range(2):
        scans = []
        for i in range(B):
            seed = 1000 * rank + j * B + i
            pts = synth.lidar_points(seed, shape)
            rng = np.random.Generator(np.random.PCG64(99 + seed))
            lab = rng.integers(0, NC, len(pts)).astype(np.int64)
            lab[rng.random(len(pts)) < 0.1] = -100
            scans.append((synth.voxelize(pts), lab, rng))
        locs = torch.cat([torch.cat([torch.from_numpy(c), torch.full((len(c), 1), i, dtype=torch.int64)], 1)
                          for i, (c, _, _) in enumerate(scans)])
        bt = dict(locs=locs.to(dev), feats=torch.ones(locs.shape[0], 1, device=dev),
                  label=torch.cat([torch.from_numpy(l) for _, l, _ in scans]).to(dev))
        if joint:
            bt["img"] = torch.stack([torch.from_numpy(r.random((3, H, W), dtype=np.float32)) for _, _, r in scans]).to(dev)
            idx = [np.stack([r.integers(0, H, len(c)), r.integers(0, W, len(c))], 1) for c, _, r in scans]
            bt["pix"] = model2d.pack_indices(idx, H, W, dev)
            bt["idx_host"] = idx
        if mopa and j == 1:
            # target-domain extras of the MoPA iteration: pseudo labels (train_xmuda_mopa.py:450-469), SAM masks (:472-480)
            # and the VGI-style third 3D batch: each scan + one 500-point object cluster, re-voxelised (:483-576)
            n = locs.shape[0]
            r0 = scans[0][2]
            for key in ("pl2d", "pl3d"):
                pl = r0.integers(0, 5, n).astype(np.int64)
                pl[r0.random(n) < 0.5] = -100
                bt[key] = torch.from_numpy(pl).to(dev)
            bt["sam"] = [torch.from_numpy(synth.sam_mask(r, H, W)).to(dev) for _, _, r in scans]
            # Valid Ground-based Insertion inputs (train_xmuda_mopa.py:483-515): per target scan the raw points (metres, lidar
            # frame with y forward like nuScenes), its pseudo labels, a ground mask (the reference's offline g_indices), one
            # 500-point car-sized object cluster + labels, the lidar -> image projection.  The insertion itself runs inside the
            # timed step, per iteration, like the reference's (mopa_amd/vgi.py on the device).
            bt["vgi"] = []
            for i in range(B):
                pts = synth.lidar_points(1000 * rank + j * B + i)[:, [1, 0, 2]].copy()
                r = scans[i][2]
                obj = ((r.random((500, 3)) - 0.5) * np.array([1.8, 4.2, 1.5]) + np.array([r.uniform(-6, 6), r.uniform(6, 14), -1.0])).astype(np.float32)
                pl = r.integers(0, 5, len(pts)).astype(np.int64)
                pl[r.random(len(pts)) < 0.5] = -100
                bt["vgi"].append(dict(ori_pc=torch.from_numpy(np.concatenate([pts, np.ones((len(pts), 1), np.float32)], 1)).to(dev),
                                      g_mask=torch.from_numpy((pts[:, 2] < -1.7).astype(np.uint8)).to(dev), pslabel=torch.from_numpy(pl).to(dev),
                                      objs=[np.concatenate([obj, np.ones((500, 1), np.float32)], 1)], obj_labels=[np.full(500, 1, np.int64)]))
            bt["vgi_proj"] = np.array([[1266.0, 800.0, 0, 0], [0, 450.0, -1266.0, 0], [0, 1, 0, 0]], np.float64)
        batches.append(bt)

    # what an empty event bracket reads on this box (median of 100, idle stream): context for the launch times below
    torch.cuda.synchronize()
    _ev = []
    for _ in range(100):
        _s, _e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _s.record(); _e.record(); _ev.append((_s, _e))
    torch.cuda.synchronize()
    bracket_overhead_us = round(sorted(a.elapsed_time(b) * 1e3 for a, b in _ev)[50], 2)
    from mopa_amd import sparse3d as sparse3d_mod
    native_default = sparse3d_mod.NATIVE
    from mopa_amd import dense2d as dense2d_mod
    graph2d_default = dense2d_mod.GRAPH_2D
    native2d_default = dense2d_mod.NATIVE_2D
    timer = ConvTimer()
    timer.install()
    timer2d = Conv2dTimer()
    if joint:
        timer2d.install()
    for b in batches:  # rule counts for the algorithmic-bytes model (one-off, outside the timed region)
        g = Geometry3D(b["locs"], 7, 4096, dev)
        for l in range(7):
            timer.rules[(27, g.num_active[l], g.num_active[l])] = g.num_rules[l]
        for l in range(6):
            timer.rules[(8, g.num_active[l + 1], g.num_active[l])] = g.num_active[l]   # down / deconv-dgrad
            timer.rules[(8, g.num_active[l], g.num_active[l + 1])] = g.num_active[l]   # up / conv-dgrad
        del g

    from mopa_amd.step import DualStream
    reorder = os.environ.get("MOPA_BENCH_REORDER", "1") != "0"
    dual = DualStream(dev, order_2d_first=reorder)  # 3D branch on a second HIP stream, overlapping the 2D GEMMs
    if os.environ.get("MOPA_BENCH_NO_SIDE") == "1":   # A/B only: the 3D branch on the main stream
        dual.side = torch.cuda.current_stream(dev)

    tl = None
    if os.environ.get("MOPA_BENCH_TIMELINE") == "1" and joint:   # diagnostics: HIP-event marks on the streams, printed on stderr
        from mopa_amd.step import Timeline
        tl = dual.timeline = Timeline()
    resident = torch.cuda.Event()
    resident.record()           # the synthetic batches are in HBM from here on
    geom_ahead = os.environ.get("MOPA_BENCH_GEOM_AHEAD", "1") != "0"
    bwd3_first = os.environ.get("MOPA_BENCH_BWD3_FIRST", "1") != "0"

    # The 3D network's optimizer lives on the side stream with the rest of the 3D branch (its gradient clear, all-reduce and Adam
    # update are ordered behind the 3D backward there), so the main stream never waits for the 3D backward: measured with
    # MOPA_BENCH_TIMELINE=1 the 3D backward of the target half ended 3.8 ms after the 2D backward -- it is enqueued behind it and
    # shares the chip with it -- and `dual.join()` in front of the optimizer steps left the main stream idle for that long every
    # step.  The next 3D forward follows on the same stream, the cross-modal losses order the streams as before.
    # MOPA_BENCH_DECOUPLE_3D_OPT=0: the joined form (both updates on the main stream behind dual.join()).
    decoupled = joint and os.environ.get("MOPA_BENCH_DECOUPLE_3D_OPT", "1") != "0"

    def vgi_batch(b):
        """Valid Ground-based Insertion (train_xmuda_mopa.py:516-555) on the device for the target batch -> (3D input of the
        augmented batch, its pseudo labels).  The per-scan loop of the reference as one batched call: same draws, same results,
        two host round trips per batch instead of four per scan (mopa_amd/vgi.py::point_mixmatch_batch; MOPA_BENCH_VGI_LOOP=1: the
        loop).  Depends on the batch only (points, pseudo labels, ground mask, object bank), not on any network output."""
        if os.environ.get("MOPA_BENCH_VGI_LOOP") == "1":
            res = [vgi.point_mixmatch(v["ori_pc"], v["pslabel"], v["objs"], v["obj_labels"], insert_mode="ground",
                                      search_voxel_size=0.5, search_range=[25.0, 25.0], search_z_min=-2.0,
                                      proj_matrix=b["vgi_proj"], image_size=(1600, 900), g_indices=v["g_mask"], front_axis="y")
                   for v in b["vgi"]]
        else:
            res = vgi.point_mixmatch_batch([dict(ori_pc=v["ori_pc"], ori_label=v["pslabel"], obj_pc_ls=v["objs"],
                                                 obj_label_ls=v["obj_labels"], g_indices=v["g_mask"]) for v in b["vgi"]],
                                           search_voxel_size=0.5, search_range=[25.0, 25.0], search_z_min=-2.0,
                                           proj_matrix=b["vgi_proj"], image_size=(1600, 900), front_axis="y")
        cat_pc, cat_lab, cat_mask = [r[0] for r in res], [r[1] for r in res], [r[2] for r in res]
        aug = {"noisy_rot": 0.1, "flip_x": 0.5, "rot_z": 6.2831, "transl": True}   # nuScenes target augmentation (yaml)
        cat_input, cat_ps, _, _ = vgi.post_process(cat_pc, cat_lab, cat_mask, 20, 4096, aug, proj_W=1080, proj_H=32)
        return cat_input, cat_ps

    def loss_3d_of(o2, o3, b, lam_xm, supervised, third=None):
        """3D losses of one domain (train_xmuda_mopa.py:354-363,389-398 source; :440-445,456-465,516-567 target).  third: (3D
        predictions on the VGI batch, its pseudo labels) when the caller ran that pass already."""
        l3 = lam_xm * xm_kl(o3["seg_logit2"], o2["seg_logit"])
        if supervised:
            l3 = l3 + seg_ce(o3["seg_logit"], b["label"], cw)
        elif mopa:
            l3 = l3 + seg_ce(o3["seg_logit"], b["pl3d"])
            if third is None:   # the VGI, then the third 3D pass on its output
                cat_input, cat_ps = vgi_batch(b)
                third = (model3d(cat_input), cat_ps)
            l3 = l3 + seg_ce(third[0]["seg_logit"], third[1])
        return l3

    def loss_2d_of(o2, o3, b, lam_xm, supervised):
        l2 = lam_xm * xm_kl(o2["seg_logit2"], o3["seg_logit"])
        if supervised:
            l2 = l2 + seg_ce(o2["seg_logit"], b["label"], cw)
        elif mopa:
            l2 = l2 + seg_ce(o2["seg_logit"], b["pl2d"])                      # lambda_pl = 1.0, ignore rows skipped in-kernel
            l2 = l2 + 0.01 * mask_cons_loss(softmax_lastdim(o2["seg_logit_all"]), b["sam"], True)   # lambda_sam_cons (yaml :66)
        return l2

    def half(b, lam_xm, supervised, ready=None):
        """One domain of the xMUDA iteration (train_xmuda_mopa.py:342-418 source, :426-449,:578-579 target)."""
        ready = ready or resident
        o2, o3 = dual.forward(model2d, model3d, {"img": b["img"], "point_pix_2d": b["pix"], "img_indices": None},
                              {"x": [b["locs"], b["feats"]]}, inputs_ready=ready if geom_ahead else None)

        def loss_3d():
            return loss_3d_of(o2, o3, b, lam_xm, supervised)

        if tl is not None:
            tl.mark("fwd_joined")
        l2 = loss_2d_of(o2, o3, b, lam_xm, supervised)
        if reorder and mopa and not supervised:
            # MoPA target half: the VGI inside loss_3d() reads a few numbers back per scan (cell counts for numpy's RNG draws) --
            # each read waits for the side stream.  The 2D backward is enqueued FIRST (the losses are independent graphs), so the
            # main stream has ~12 ms of work while the host sits in those round trips; the side stream is ordered behind the 2D
            # losses only (event), not behind that backward.
            ready = torch.cuda.Event()
            ready.record()
            opts[1].arm_buckets()   # the last 2D backward of the iteration: its gradient buckets are reduced as they complete
            l2.backward()
            with dual.on_side(o2["seg_logit"], after=ready):
                l3 = loss_3d()
                l3.backward()   # called ON the side stream: see below
            return l2.detach(), l3.detach()
        if reorder:
            # the 3D losses on the side stream: their backward -- the whole 3D backward -- is then queued there
            # and runs beside the 2D backward without the host having to enqueue it first while the main stream waits
            with dual.on_side(o2["seg_logit"]):
                l3 = loss_3d()
            if tl is not None:
                tl.mark("losses_done")
            if not supervised:
                opts[1].arm_buckets()   # target half = the last 2D backward of the iteration
            l2.backward()
            if tl is not None:
                tl.mark("bwd2d_end")
            # backward() ends by making the stream that is CURRENT AT THE CALL wait for every stream the backward pass ran on
            # (autograd's stream semantics): called from the main stream, `l3.backward()` was a hidden join -- the main stream
            # sat idle until the 3D backward had finished, 3.8 ms in each half (MOPA_BENCH_TIMELINE=1).  Called with the side
            # stream current, nothing waits; the optimizer of the 3D network is on that stream too (step()).
            if decoupled:
                dual.backward_on_side(l3)
            else:
                l3.backward()
            if tl is not None:
                tl.mark("bwd3d_end", dual.side)
            return l2.detach(), l3.detach()
        l3 = loss_3d()
        if bwd3_first:   # the 3D backward (side stream) starts as soon as its loss gradient exists, beside the 2D backward
            l3.backward()
            l2.backward()
        else:
            l2.backward()
            l3.backward()
        return l2.detach(), l3.detach()

    # ---- both domains with the 2D branch in ONE pass.  The reference sends the source batch and then the target batch through
    # the 2D network (train_xmuda_mopa.py:342,426) with no weight update in between; the two passes are independent except for the
    # order in which BatchNorm updates its running statistics.  Net2DSeg's "bn_groups": 2 keeps exactly that (statistics, running
    # updates and dropout masks per half, in order: tests/test_gpu_2d.py) while every convolution sees both halves at once: half
    # the launches, twice the rows per launch -- 2 x 22.0 ms -> 40.5 ms for forward + backward at 8 + 8 images, 2 x 13.1 -> 22.0 at
    # 4 + 4, 2 x 8.8 -> 13.1 at 2 + 2 (2D branch alone, profiles/graph_probe.py).  The 3D branch still runs per domain (its rows
    # are not grouped by scan), both forwards on the side stream beside the 2D forward, both backwards beside the 2D backward.
    # MOPA_BENCH_PAIR=0: the two-call form (half() twice).
    pair_mode = joint and os.environ.get("MOPA_BENCH_PAIR", "1") != "0" and reorder and (decoupled or mopa)

    # The same for the 3D network (Net3DSeg "bn_group_points"): both domains' scans as one sparse tensor, BatchNorm per domain on row
    # ranges (the source's rows come first at every level).  The 3D branch is bound by per-launch costs, not by rows: one pass over
    # 16 scans takes 9.3 ms where two passes over 8 take 11.2, 8 scans 5.6 where two of 4 take 7.9 (`--workload 3d --batch`).
    # MOPA_BENCH_PAIR_3D=0: one 3D pass per domain.
    pair_3d = os.environ.get("MOPA_BENCH_PAIR_3D", "1") != "0"
    pair_3d_three = os.environ.get("MOPA_BENCH_PAIR_3D_THREE", "1") != "0"   # MoPA: the VGI batch as a third group of that pass
    vgi_stream = torch.cuda.Stream(device=dev) if (mopa and os.environ.get("MOPA_BENCH_VGI_STREAM", "1") != "0") else None

    def pair_batch_of(bs, bt):
        from mopa_amd.step import merge_domains_2d, merge_domains_3d
        Bs = bs["img"].shape[0]
        p2 = merge_domains_2d(bs["img"], bt["img"], bs["pix"], bt["pix"])
        p3 = merge_domains_3d([(bs["locs"], bs["feats"]), (bt["locs"], bt["feats"])], [Bs, Bs]) if pair_3d else None
        return p2, p3

    mode = {"pair": pair_mode}   # (mutable: the two-calls-per-domain figure after the timed region switches it off)
    pair2d = pair_batch_of(batches[0], batches[1]) if pair_mode else None
    if pair_mode and pair2d[1] is not None:   # rule counts of the joint geometry for the algorithmic-bytes model (as above)
        g = Geometry3D(pair2d[1]["x"][0], 7, 4096, dev, group_points=pair2d[1]["bn_group_points"])
        for l in range(7):
            timer.rules[(27, g.num_active[l], g.num_active[l])] = g.num_rules[l]
        for l in range(6):
            timer.rules[(8, g.num_active[l + 1], g.num_active[l])] = g.num_active[l]
            timer.rules[(8, g.num_active[l], g.num_active[l + 1])] = g.num_active[l]
        del g
    if pair_mode:
        torch.cuda.synchronize()
        resident.record()

    def pair(bs, bt, p2d, ready=None):
        ready = ready or resident
        main = torch.cuda.current_stream(dev)
        # MoPA: the VGI needs the batch only.  Its two host round trips wait for whatever is queued in front of its kernels: on the
        # side stream that is the tail of the previous step's 3D backward (5-6 ms into the step), and the host cannot enqueue the 3D
        # forwards until they are through.  On a stream of its own they return as soon as the VGI kernels have run
        # (MOPA_BENCH_VGI_STREAM=0: on the side stream, after the 2D forward's enqueue).
        vin, vgi_done = None, None
        if mopa and vgi_stream is not None:
            with torch.cuda.stream(vgi_stream):
                vgi_stream.wait_event(ready)
                vin = vgi_batch(bt)
                vgi_done = torch.cuda.Event()
                vgi_done.record()
        dual.side.wait_stream(main)
        # the long queue first; the 3D launches are enqueued while it runs.  (Measured alternatives, no gain: the 3D passes enqueued
        # first -- 238 -> 220 scans/s MoPA, 191 -> 176 kitti; the side stream not ordered behind the main stream at the step
        # boundary -- unchanged: at 4 + 4 and 2 + 2 images the 3D chain fills its stream for the whole step)
        o2 = model2d(p2d[0])
        if tl is not None:
            tl.mark("fwd2d_end", main)
        third = None
        with torch.cuda.stream(dual.side):
            dual.side.wait_event(ready)
            if mopa and vin is None:
                vin = vgi_batch(bt)
            if vin is not None and vgi_done is not None:
                dual.side.wait_event(vgi_done)
                for t in list(vin[0]["x"]) + [vin[1]]:
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(dual.side)
            if p2d[1] is not None:
                p3 = p2d[1]
                locs3, feats3, cuts = p3["x"][0], p3["x"][1], p3["bn_group_points"]
                if vin is not None and pair_3d_three:   # MoPA: the VGI batch as a third group of the same pass (scan indices behind)
                    from mopa_amd.step import merge_domains_3d
                    nbs = bs["img"].shape[0]
                    m3 = merge_domains_3d([(bs["locs"], bs["feats"]), (bt["locs"], bt["feats"]), (vin[0]["x"][0], vin[0]["x"][1])],
                                          [nbs, nbs, nbs])
                    (locs3, feats3), cuts = m3["x"], m3["bn_group_points"]
                o3m = model3d({"x": [locs3, feats3], "bn_group_points": cuts,
                               "geometry_3d": model3d.net_3d.geometry(locs3, group_points=cuts)})
                n3 = p3["bn_group_points"]
                n4 = int(p3["x"][0].shape[0])
                o3 = [{k: v[:n3] for k, v in o3m.items()}, {k: v[n3:n4] for k, v in o3m.items()}]
                if vin is not None and pair_3d_three:
                    third = ({k: v[n4:] for k, v in o3m.items()}, vin[1])
            else:
                o3 = []
                for b in (bs, bt):
                    g = model3d.net_3d.geometry(b["locs"])
                    o3.append(model3d({"x": [b["locs"], b["feats"]], "geometry_3d": g}))
            if vin is not None and third is None:
                third = (model3d(vin[0]), vin[1])
        if tl is not None:
            tl.mark("fwd3d_end", dual.side)
        main.wait_stream(dual.side)
        for o in o3:
            for t in o.values():
                if torch.is_tensor(t):
                    t.record_stream(main)
        ns, nb = bs["pix"].numel(), bs["img"].shape[0]
        o2s = {k: (v[:nb] if k == "seg_logit_all" else v[:ns]) for k, v in o2.items()}
        o2t = {k: (v[nb:] if k == "seg_logit_all" else v[ns:]) for k, v in o2.items()}
        l2 = loss_2d_of(o2s, o3[0], bs, lam_src, True) + loss_2d_of(o2t, o3[1], bt, lam_trg, False)
        ev = torch.cuda.Event()
        ev.record()
        opts[1].arm_buckets()                   # (no-op without a process group)
        l2.backward()                           # main stream: the 2D backward of both halves
        if tl is not None:
            tl.mark("bwd2d_end", main)
        # the 3D losses (for MoPA: with the VGI and its host round trips) and both 3D backwards on the side stream, ordered behind
        # the 2D losses only -- they run beside the 2D backward
        with dual.on_side(o2["seg_logit"], o2["seg_logit2"], after=ev):
            l3 = loss_3d_of(o2s, o3[0], bs, lam_src, True) + loss_3d_of(o2t, o3[1], bt, lam_trg, False, third)
            l3.backward()
        if tl is not None:
            tl.mark("bwd3d_end", dual.side)
        return l2.detach(), l3.detach()

    # Loss means are per rank; the reference's single-process mean over the global batch weights rank r by N_r / sum N
    # (SURVEY 8e).  The point counts are known on the host when the batch is built, so the one-float exchange happens here,
    # outside the timed region; with equal counts (this synthetic workload) every weight is exactly 1.
    from mopa_amd.step import global_mean_weight
    rank_weight = global_mean_weight(sum(int(b["locs"].shape[0]) for b in batches))
    # (RCCL only: the gloo plumbing mode stages device tensors through host memory inside the collective -- nothing to overlap)
    overlap_3d = os.environ.get("MOPA_BENCH_OVERLAP_AR", "1") != "0" and multi and dist.get_backend() == "nccl"

    # ---- the reference boundary's real hand-off (mopa/data/collate.py:183-186,233-235): coords int64 / feats / labels / images as
    # HOST tensors and img_indices as numpy arrays.  A copy stream uploads the next half's inputs (pageable copies, see
    # _lib.upload) while the current half computes; an event orders the consumers.  Timed AFTER the main region and reported as
    # `value_with_host_inputs` beside `value` (which by contract has its inputs resident in HBM).
    host_batches = None
    shared_gpu = multi and dist.get_backend() != "nccl"   # plumbing mode: several ranks on one device, keep the stream count down
    if joint and not mopa and not shared_gpu and os.environ.get("MOPA_BENCH_HOST_INPUTS", "1") != "0":
        host_batches = []
        for bt in batches:
            host_batches.append(dict(locs=bt["locs"].cpu(), feats=bt["feats"].cpu(), label=bt["label"].cpu(), img=bt["img"].cpu(),
                                     idx=[np.ascontiguousarray(a) for a in bt.pop("idx_host")]))
    copy_stream = torch.cuda.Stream(device=dev) if host_batches is not None else None

    def stage(hb):
        """Upload one half's inputs on the copy stream -> (device batch, event)."""
        with torch.cuda.stream(copy_stream):
            d = dict(locs=hb["locs"].to(dev), feats=hb["feats"].to(dev), label=hb["label"].to(dev), img=hb["img"].to(dev),
                     pix=model2d.pack_indices(hb["idx"], H, W, dev))
            ev = torch.cuda.Event()
            ev.record()
        for t in d.values():   # allocated from the copy stream's pool, consumed on the main and the side stream
            t.record_stream(torch.cuda.current_stream(dev))
            t.record_stream(dual.side)
        return d, ev

    host_delay = float(os.environ.get("MOPA_BENCH_HOST_DELAY_MS", "0")) * 1e-3   # diagnostics: is the step host-paced?

    steps_run = [0]

    def step(i, host_fed=False):
        steps_run[0] += 1
        if host_delay:
            time.sleep(host_delay)
        if tl is not None:
            tl.begin()
            tl.mark("step_start")
        if decoupled:
            with torch.cuda.stream(dual.side):
                opts[0].zero_grad()
            opts[1].zero_grad()
        else:
            for o in opts:
                o.zero_grad()
        work3 = None
        if joint and host_fed and mode["pair"]:
            d0, e0 = stage(host_batches[0])
            d1, e1 = stage(host_batches[1])
            torch.cuda.current_stream(dev).wait_event(e0)
            torch.cuda.current_stream(dev).wait_event(e1)
            parts = pair(d0, d1, pair_batch_of(d0, d1), ready=e1)   # (the copy stream is in order: e1 covers both uploads)
        elif joint and host_fed:
            d0, e0 = stage(host_batches[0])
            d1, e1 = stage(host_batches[1])   # in flight while the source half computes
            torch.cuda.current_stream(dev).wait_event(e0)
            pa = half(d0, lam_src, True, ready=e0)
            torch.cuda.current_stream(dev).wait_event(e1)
            parts = pa + half(d1, lam_trg, False, ready=e1)
        elif joint and mode["pair"]:
            parts = pair(batches[0], batches[1], pair2d)
            if multi and overlap_3d:
                with torch.cuda.stream(dual.side):
                    work3 = opts[0].all_reduce(async_op=True)
        elif joint:
            parts = half(batches[0], lam_src, True) + half(batches[1], lam_trg, False)   # source: CE + lambda_xm_src * KL, target: lambda_xm_trg * KL (yaml :56-57)
            if multi and overlap_3d and not host_fed:
                # the 3D network's gradients are complete when the side stream drains: reduce them there, under the tail of the
                # 2D backward on the main stream (RCCL orders its own stream behind the stream current at the call)
                with torch.cuda.stream(dual.side):
                    work3 = opts[0].all_reduce(async_op=True)
        else:
            b = batches[i % 2]
            # the voxel geometry depends on the coordinates only: built beside the previous step's backward (loader-side work)
            geom = dual.geometry_ahead(model3d, b["locs"], resident) if geom_ahead else Geometry3D(b["locs"], 7, 4096, dev)
            out = model3d({"x": [b["locs"], b["feats"]], "geometry_3d": geom})
            loss = seg_ce(out["seg_logit"], b["label"], cw) + seg_ce(out["seg_logit2"], b["label"], cw)
            loss.backward()
        if decoupled:
            with torch.cuda.stream(dual.side):   # 3D: reduce + update behind its own backward
                if work3 is not None:
                    work3.wait()                 # the side stream waits for the collective; the host does not block
                else:
                    opts[0].all_reduce()
                opts[0].step(rank_weight / world)
            opts[1].all_reduce()                 # 2D: on the main stream behind the 2D backward
            opts[1].step(rank_weight / world)
            if tl is not None:
                tl.mark("step_end")
            return parts                         # (l2, l3) x 2 halves, summed after the final synchronisation
        if joint:
            dual.join()  # 3D backward done before its gradients are reduced / applied
            for t in parts:   # the 3D parts live in the side stream's pool
                t.record_stream(torch.cuda.current_stream())
            loss = sum(parts[1:], parts[0])
        for k, o in enumerate(opts):
            if k == 0 and work3 is not None:
                continue
            o.all_reduce()
        if work3 is not None:
            work3.wait()   # the current stream waits for the collective; the host does not block
        for o in opts:
            o.step(rank_weight / world)
        if tl is not None:
            tl.mark("step_end")
        return loss

    scans_per_step = 2 * B if joint else B
    t_setup = time.perf_counter()
    # Setup steps before the W warm-up steps.  On a box that has never run this process (kernel code objects loaded from disk at their
    # first launch, first hipMalloc calls) the first two or three steps take 100-250 ms: the host, not the device, paces them, so
    # few tensors are in flight and the caching allocator's pool stays smaller than the steady state needs -- the pool then grows
    # INSIDE the timed region (hipMalloc of GB-sized segments, 30-100 ms each: measured 284-295 scans/s instead of 328-331 on such a
    # box, 19 new segments, reserved 37 -> 47 GB; on a box that had run the process before: 13 small ones, 43 -> 45 GB).  Two extra
    # untimed steps and a synchronisation put that one-off cost in front of the warm-up; reported as `setup_steps` in the line.
    setup_steps = int(os.environ.get("MOPA_BENCH_SETUP_STEPS", "2"))
    for i in range(setup_steps):
        step(i)
    torch.cuda.synchronize()
    paced_step = step
    for i in range(args.warmup):
        paced_step(i)
    torch.cuda.synchronize()
    print(f"[bench] rank {rank}: warmup {args.warmup} steps in {time.perf_counter() - t_setup:.2f}s", file=sys.stderr, flush=True)
    if os.environ.get("MOPA_BENCH_GC_FREEZE", "1") != "0":
        # what a trainer does once before its iteration loop (mopa_amd.step.freeze_host_heap): without it one full pass of
        # Python's cyclic collector (70-100 ms over torch's ~2 M module-level objects) lands somewhere in the timed region
        from mopa_amd.step import freeze_host_heap
        freeze_host_heap()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP-event brackets for the roofline figures on every EVENT_STRIDE-th timed step: on every step they cost 2.8 % (joint) /
    # 4.4 % (3d) of `value` (MOPA_BENCH_EVENT_STRIDE=1 / =0 measure that: every step / never)
    ev_stride = int(os.environ.get("MOPA_BENCH_EVENT_STRIDE", "5"))
    n_ev_steps = 0
    t0 = time.perf_counter()
    per_step, step_times = [], os.environ.get("MOPA_BENCH_STEP_TIMES")
    ms0 = torch.cuda.memory_stats() if step_times else None
    gc_log = []
    if step_times:   # diagnostics: cyclic-GC passes inside the timed region (generation, duration)
        import gc

        def _gc_cb(phase, info, _t=[0.0]):
            if phase == "start":
                _t[0] = time.perf_counter()
            else:
                gc_log.append((info["generation"], time.perf_counter() - t0, time.perf_counter() - _t[0]))
        gc.callbacks.append(_gc_cb)
    for i in range(args.steps):
        timer.enabled = timer2d.enabled = ev_stride > 0 and i % ev_stride == 0
        n_ev_steps += int(timer.enabled)
        # a bracketed step walks the 3D layer program from Python (same kernels, same order, bit-identical: tests/test_gpu_3d.py) so
        # that every sparse-conv launch can be bracketed; all other steps run it as one native call per pass (csrc/scn_exec.hip)
        sparse3d_mod.NATIVE = native_default and not timer.enabled
        dense2d_mod.GRAPH_2D = graph2d_default and not timer2d.enabled   # same for the 2D backbone: brackets need the eager walk
        dense2d_mod.NATIVE_2D = native2d_default and not timer2d.enabled
        # the bracketed steps run in the SAME stream configuration as every other step (weight-gradient stream and 3D side stream
        # on): the brackets then time each launch as it runs inside `value`'s step, sharing the chip with the other streams --
        # which is also what `rocprofv3 --kernel-trace --stats` of this command reports (profiles/r3_final_*)
        loss = paced_step(i)
        if step_times:   # diagnostics only (stderr): host clock after each step, "sync" adds a device sync per step
            if step_times == "sync":
                torch.cuda.synchronize()
            per_step.append(time.perf_counter() - t0)
    if per_step:
        ms1 = torch.cuda.memory_stats()
        print("[bench] allocator in the timed region: segments malloc'd", ms1["segment.all.allocated"] - ms0["segment.all.allocated"],
              "freed", ms1["segment.all.freed"] - ms0["segment.all.freed"], "retries", ms1["num_alloc_retries"] - ms0["num_alloc_retries"],
              "reserved GB", round(ms1["reserved_bytes.all.current"] / 1e9, 2), "was", round(ms0["reserved_bytes.all.current"] / 1e9, 2),
              file=sys.stderr, flush=True)
        print("[bench] cyclic-GC passes >1 ms (generation, at s, lasted ms):",
              [(g, round(at, 3), round(1e3 * d, 1)) for g, at, d in gc_log if d > 1e-3], file=sys.stderr, flush=True)
        print("[bench] cumulative step times:", " ".join(f"{t:.3f}" for t in per_step), file=sys.stderr, flush=True)
    t_enqueued = time.perf_counter() - t0   # host time to enqueue the steps (stderr only): ~= elapsed means launch-bound
    if tl is not None:
        torch.cuda.synchronize()
        tl.steps = tl.steps[args.warmup:]
        print("[bench] timeline (mean ms from step start): " + ", ".join(f"{n} {t:.2f}" for n, t in tl.report()), file=sys.stderr, flush=True)
        tl = dual.timeline = None
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = timer2d.enabled = False
    sparse3d_mod.NATIVE = native_default
    dense2d_mod.GRAPH_2D = graph2d_default
    dense2d_mod.NATIVE_2D = native2d_default
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if isinstance(loss, tuple):   # decoupled streams: the four loss parts, read after the synchronisation above
        loss = sum(t.float() for t in loss)
    assert torch.isfinite(loss).item(), "loss is not finite"
    host_value = None
    if host_batches is not None:   # same step, inputs handed over as host tensors (not part of `value`)
        n_host = max(2, min(args.steps, 8))
        for i in range(2):
            paced_step(i, host_fed=True)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        th = time.perf_counter()
        for i in range(n_host):
            paced_step(i, host_fed=True)
        torch.cuda.synchronize()
        el_h = time.perf_counter() - th
        if multi:
            t = torch.tensor([el_h], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_h = t.item()
        host_value = (world * scans_per_step * n_host / el_h, n_host)

    def timed_extra(n, **kw):
        """n further steps after two untimed ones, max over ranks -> scans/s (figures beside `value`, never `value`)."""
        for i in range(2):
            paced_step(i, **kw)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        t_ = time.perf_counter()
        for i in range(n):
            paced_step(i, **kw)
        torch.cuda.synchronize()
        el = time.perf_counter() - t_
        if multi:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return world * scans_per_step * n / el

    # ---- the reference's loop as written: model(source) then model(target), two passes per network and iteration
    # (train_xmuda_mopa.py:342-343,426-427).  `value` merges them (bn_groups / bn_group_points: an extension of the data_batch contract,
    # INTEGRATION.md); this is the same step without the extension, a few steps after the timed region, reported beside `value`.
    two_call_value = None
    if joint and pair_mode and decoupled and os.environ.get("MOPA_BENCH_TWO_CALLS", "1") != "0":
        n_two = max(2, min(args.steps, 8))
        mode["pair"] = False
        two_call_value = (timed_extra(n_two), n_two)
        mode["pair"] = pair_mode

    # ---- the sparse-conv roofline figure of THIS run (joint workloads): the merged 3D pass (both domains' scans as one sparse tensor,
    # exactly the launches of the timed step) forward + backward with the 2D branch idle, the layer program walked from Python so that
    # every sparse-conv launch sits in a HIP-event bracket on its stream.  Inside the timed step the family shares the chip and a
    # lower-priority stream with the 2D GEMMs: a bracket there is mostly queue wait (kept as frac_event_bracket_with_queue_wait).
    ks_alone = None
    if joint and rank == 0 and pair_mode and pair2d[1] is not None and os.environ.get("MOPA_BENCH_SPARSE_ALONE", "1") != "0":
        p3 = pair2d[1]
        lab3 = torch.cat([batches[0]["label"], batches[1]["label"]])
        ks_main_records = timer.records
        timer.records = []
        sparse3d_mod.NATIVE = False

        def step3d_alone():
            opts[0].zero_grad()
            g3 = model3d.net_3d.geometry(p3["x"][0], group_points=p3["bn_group_points"])
            o = model3d({"x": p3["x"], "bn_group_points": p3["bn_group_points"], "geometry_3d": g3})
            l = seg_ce(o["seg_logit"], lab3, cw) + seg_ce(o["seg_logit2"], lab3, cw)
            l.backward()

        for _ in range(2):
            step3d_alone()
        torch.cuda.synchronize()
        timer.enabled = True
        n_alone = 5
        for _ in range(n_alone):
            step3d_alone()
        torch.cuda.synchronize()
        timer.enabled = False
        ks_alone = timer.summary()
        if ks_alone:
            ks_alone["steps"] = n_alone
        timer.records = ks_main_records
        sparse3d_mod.NATIVE = native_default
        opts[0].zero_grad()

    if rank == 0:
        ks = timer.summary()
        sp = None
        if ks:
            sp = {"bound": "hbm", "achieved": round(ks["gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(ks["gbs"] / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_spconv_t4 / k_spconv_pipe / k_spconv_blk / k_spconv_run + k_run_reduce / k_spconv_stem (sparse conv fwd + bwd-data)",
                  "launches_per_step": ks["launches"] // max(n_ev_steps, 1), "timed_steps_bracketed": n_ev_steps, "avg_launch_us": round(ks["avg_us"], 2),
                  "algorithmic_bytes_per_launch": round(ks["bytes_per_launch"]), "mfma_tflops": round(ks["tflops"], 2),
                  # the family is not HBM-bound on every level: from level 3 down (128 -> 64 channels and wider, few rows) a launch's
                  # flops at the fp32 MFMA peak take longer than its bytes at the HBM peak.  mixed = sum over launches of
                  # max(bytes / 8 TB/s, flops / 157.3 TF/s) / measured time -- the fraction of the tighter ceiling per launch
                  "mixed_roofline": {"ideal_us_per_launch": round(ks["ideal_us_per_launch"], 2),
                                     "mfma_bound_launches_per_step": ks["mfma_bound_launches"] // max(n_ev_steps, 1),
                                     "frac_event_bracket": round(ks["ideal_us_per_launch"] / ks["avg_us"], 4)}}
        # PMC traffic and rocprofv3 launch durations come from COMMITTED files (counters need rocprofv3 around the process): the
        # files carry the commit they were taken at (profiles/collect_final.py), the line says so
        wl_key = "3d" if not joint else ("kitti" if kitti else "mopa" if mopa else "joint")
        def _prof(name):   # committed profile summaries: this round's if present, else the previous round's
            for rnd in ("r6", "r5", "r4", "r3"):
                q = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
                if os.path.exists(q):
                    return q
            return os.path.join(ROOT, "profiles", f"r6_{name}")

        fam_path = _prof("rocprof_family.json")
        famj = json.load(open(fam_path)) if os.path.exists(fam_path) else {}
        # (every workload has its own PMC passes: the paired steps run both domains' scans as one sparse tensor -- other launches than
        #  `--workload 3d`'s)
        t3_key = wl_key
        t3 = _prof(f"{t3_key}_hbm_traffic.json")  # PMC passes of that workload's command (profiles/traffic.py)
        if sp and t3_key and os.path.exists(t3):
            d3 = json.load(open(t3))
            fam = [d3[k] for k in ("k_spconv_t4", "k_spconv_pipe", "k_spconv_fwd", "k_spconv_blk", "k_spconv_run", "k_spconv_stem") if k in d3]
            extra = [d3[k] for k in ("k_run_reduce",) if k in d3]   # (its bytes belong to the offset-major launches, it is not a launch of its own)
            if fam:
                sp["traffic"] = int(sum(f["hbm_bytes_per_launch"] * f["launches"] for f in fam + extra) / sum(f["launches"] for f in fam))
                sp["traffic_source"] = ("%s at commit %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                        "`bench.py --workload %s`, per launch)" % (os.path.relpath(t3, ROOT), d3.get("_commit", "?"), t3_key))
        if sp and joint:
            # inside the joint step the family shares the chip and its (lower-priority) side stream with the 2D GEMMs: an event
            # bracket there is mostly queue wait, not kernel time.  `frac` is therefore NOT reported from it; see frac_rocprof (kernel
            # durations of the same command) and `--workload 3d` for the uncontended figure
            sp["frac_event_bracket_with_queue_wait"] = sp.pop("frac")
            sp["achieved_event_bracket_with_queue_wait"] = sp.pop("achieved")
            sp["frac"], sp["achieved"] = None, None
            sp["avg_launch_us_event_bracket_with_queue_wait"] = sp.pop("avg_launch_us")
            if ks_alone:   # measured in THIS run: the same launches with the 2D branch idle
                sp["frac"], sp["achieved"] = round(ks_alone["gbs"] / HBM_PEAK_GBS, 4), round(ks_alone["gbs"], 1)
                sp["avg_launch_us"] = round(ks_alone["avg_us"], 2)
                sp["mfma_tflops"] = round(ks_alone["tflops"], 2)
                sp["launches_per_step"] = ks_alone["launches"] // ks_alone["steps"]
                sp["algorithmic_bytes_per_launch"] = round(ks_alone["bytes_per_launch"])
                sp["mixed_roofline"] = {"ideal_us_per_launch": round(ks_alone["ideal_us_per_launch"], 2),
                                        "mfma_bound_launches_per_step": ks_alone["mfma_bound_launches"] // ks_alone["steps"],
                                        "frac_event_bracket": round(ks_alone["ideal_us_per_launch"] / ks_alone["avg_us"], 4)}
                sp["frac_measured_in"] = (f"this run, after the timed region: {ks_alone['steps']} forward + backward passes of the merged 3D batch "
                                          "(source + target scans as one sparse tensor: the launches of the timed step) with the 2D branch idle, "
                                          "every launch of the family in a HIP-event bracket on its stream")
            sp["note"] = ("inside the timed step the 3D branch runs on a second stream beside the 2D branch: HIP-event brackets there include queue "
                          "wait (frac_event_bracket_with_queue_wait); frac / achieved are measured in this run with the 2D branch idle; frac_rocprof = "
                          "rocprofv3 kernel durations of this command from the committed profile")
        def _fresh(rf_, launches_per_step):
            """A committed rocprofv3 figure belongs to this code only if the family is launched as often per step as when it was taken
            (e.g. one 2D pass per iteration instead of two halves the launch count and doubles the launch size)."""
            cps = rf_.get("calls_per_step")
            return cps is None or abs(cps - launches_per_step) <= 0.06 * launches_per_step

        rf = famj.get(wl_key, {}).get("sparse_conv")
        if sp and rf and not _fresh(rf, sp["launches_per_step"]):
            sp["rocprof_note"] = f"{os.path.relpath(fam_path, ROOT)} was taken with another launch count per step: not used"
            rf = None
        if sp and rf:
            gbs = sp["algorithmic_bytes_per_launch"] / (rf["avg_us"] * 1e-6) / 1e9
            sp["frac_rocprof"], sp["avg_launch_us_rocprof"], sp["rocprof_commit"] = round(gbs / HBM_PEAK_GBS, 4), rf["avg_us"], famj.get("commit")
            sp["mixed_roofline"]["frac_rocprof"] = round(sp["mixed_roofline"]["ideal_us_per_launch"] / rf["avg_us"], 4)
            # (a committed figure is never `frac`: the file cannot know whether the kernels changed since -- ADVICE r3)
            sp["rocprof_source"] = os.path.relpath(fam_path, ROOT)
        roof = sp
        hbm_step = (None, None)
        tall = _prof(f"{wl_key}_hbm_traffic.json")
        if os.path.exists(tall):
            tdat = json.load(open(tall))
            if tdat.get("_hbm_GB_per_step") is not None:
                hbm_step = (tdat["_hbm_GB_per_step"], f"{os.path.relpath(tall, ROOT)} at commit {tdat.get('_commit', '?')} (rocprofv3 --pmc FETCH_SIZE / "
                            f"WRITE_SIZE passes of `bench.py --workload {wl_key}`, {tdat.get('_steps')} steps, guide's gfx950 corrections)")
        k2 = timer2d.summary() if joint else None
        traffic = None
        tpath = _prof(f"{wl_key}_hbm_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        tj = {}
        if joint and os.path.exists(tpath):                                      # of this same command (profiles/traffic.py)
            tj = json.load(open(tpath))
            fam = [tj[k] for k in ("k_conv2d_igemm_mfma", "k_wino4_gemm_out", "k_wino4_conv", "k_wino4_conv32", "k_wino4_conv9") if k in tj]
            if fam:
                traffic = int(sum(f["hbm_bytes_per_launch"] * f["launches"] for f in fam) / sum(f["launches"] for f in fam))
        if k2:  # the dominant kernel of the joint step is the dense implicit-GEMM conv: compute-bound fp32
            roof = {"bound": "mfma", "achieved": round(k2["tflops"], 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(k2["tflops"] / F32_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": f"{os.path.relpath(tpath, ROOT)} at commit {tj.get('_commit', '?')} (PMC passes of this command; HBM bytes per launch)",
                    "kernel": "k_conv2d_igemm_mfma + k_wino4_gemm_out + k_wino4_conv9 / k_wino4_conv / k_wino4_conv32 (f32-operand MFMA, exact fp32 products; conv fwd + bwd-data + convT + the Winograd layers' GEMMs, flops as executed)",
                    "launches_per_step": k2["launches"] // max(n_ev_steps, 1), "timed_steps_bracketed": n_ev_steps, "avg_launch_us": round(k2["avg_us"], 2),
                    "algorithmic_flops_per_launch": round(k2["tflops"] * 1e12 * k2["avg_us"] * 1e-6),
                    "algorithmic_bytes_per_launch": round(k2["bytes_per_launch"]),
                    "event_bracket_overhead_us": bracket_overhead_us,   # an EMPTY HIP-event bracket on the idle stream, measured at start-up
                    "note": ("HIP-event brackets measure from the retirement of the stream's previous packet to the end of the kernel: under "
                             "the three-stream overlap they include the launch gap in front of the kernel (command-processor arbitration, "
                             "waiting for free CUs), which rocprofv3's kernel durations of the same command do not "
                             "(profiles/r3_final_joint_kernel_stats.md; frac_rocprof below)"),
                    "direct_conv_equivalent_tflops": round(k2["tflops_direct"], 1),   # NOT the roofline figure: what a direct 3x3 conv would have to sustain for the same launch times (Winograd executes 2.25x / 4x fewer flops)
                    "stream_configuration": "as timed for `value`: 2D main stream + " +
                                            ("weight-gradient stream + " if dense2d_streams() else "") + "3D side stream"}
            rd = famj.get(wl_key, {}).get("dense_mfma")
            if rd and not _fresh(rd, roof["launches_per_step"]):
                roof["rocprof_note"] = f"{os.path.relpath(fam_path, ROOT)} was taken with another launch count per step: not used"
                rd = None
            if rd:   # the same flops over rocprofv3's kernel durations of this command (no launch gaps): the two figures bracket the truth
                roof["frac_rocprof"] = round(roof["algorithmic_flops_per_launch"] / (rd["avg_us"] * 1e-6) / 1e12 / F32_PEAK_TFLOPS, 4)
                roof["avg_launch_us_rocprof"], roof["rocprof_commit"] = rd["avg_us"], famj.get("commit")
        roof_wgrad = None
        kw = timer2d.wgrad_summary() if joint else None
        if kw:
            roof_wgrad = {"bound": "mfma", "achieved": round(kw["tflops"], 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(kw["tflops"] / F32_PEAK_TFLOPS, 4), "traffic": None,
                          "kernel": "weight gradients of the 2D network: k_conv2d_wgrad_mfma (direct and transform-domain batches) / k_wgemm_tn (transform-domain, "
                                    "128-aligned channels, operands by LDS-DMA) / k_wino4_wgrad "
                                    "(one-kernel F(4x4): x and dY in) / k_stem_wgrad_mfma + their ordered slab reductions (k_reduce_slabs2, "
                                    "k_wino4_dw), one bracket per weight gradient, flops as executed (36 T Cin Cout 2 for an F(4x4) layer)",
                          "launches_per_step": kw["launches"] // max(n_ev_steps, 1), "timed_steps_bracketed": n_ev_steps,
                          "avg_launch_us": round(kw["avg_us"], 2),
                          "algorithmic_flops_per_launch": round(kw["tflops"] * 1e12 * kw["avg_us"] * 1e-6),
                          "stream_configuration": "the weight-gradient stream beside the backward-data chain (as timed for `value`)"
                          if dense2d_streams() else "the main stream",
                          "note": "the transforms that feed the two-operand form (k_wino4_dout, the V kept or recomputed) are not in the brackets"}
            # HBM bytes per weight gradient from the same committed PMC passes: the MFMA kernels + their slab reductions, per MFMA launch
            wmain = [tj[k] for k in ("k_conv2d_wgrad_mfma", "k_wino4_wgrad", "k_stem_wgrad_mfma", "k_wgemm_tn") if k in tj]
            wextra = [tj[k] for k in ("k_reduce_slabs2", "k_wino4_dw", "k_wino_dw") if k in tj]
            if wmain:
                roof_wgrad["traffic"] = int(sum(f["hbm_bytes_per_launch"] * f["launches"] for f in wmain + wextra) / sum(f["launches"] for f in wmain))
                roof_wgrad["traffic_source"] = (f"{os.path.relpath(tpath, ROOT)} at commit {tj.get('_commit', '?')} (PMC passes of this command; HBM bytes of the "
                                                "MFMA kernels and their slab reductions per weight gradient)")
                roof_wgrad["algorithmic_bytes_per_launch_note"] = ("not priced: a weight gradient reads its layer's input and output gradient once "
                                                                   "(x + dY; V + dM = 2.25 x that for the two-operand transform-domain form) and writes 36 Cin Cout floats per split")
            rw = famj.get(wl_key, {}).get("wgrad_mfma")
            if rw and not _fresh(rw, roof_wgrad["launches_per_step"]):
                roof_wgrad["rocprof_note"] = f"{os.path.relpath(fam_path, ROOT)} was taken with another launch count per step: not used"
                rw = None
            if rw:
                roof_wgrad["frac_rocprof"] = round(roof_wgrad["algorithmic_flops_per_launch"] / (rw["avg_us"] * 1e-6) / 1e12 / F32_PEAK_TFLOPS, 4)
                roof_wgrad["avg_launch_us_rocprof"], roof_wgrad["rocprof_commit"] = rw["avg_us"], famj.get("commit")
        wl = ("MoPA iteration per GPU (BASELINE configs[3] shape): "
              f"{B} source + {B} target scans, CE + cross-modal KL + pseudo-label CE + SAM-mask consistency loss + "
              "Valid Ground-based Insertion of a 500-pt object per target scan on the device (overlap test, ground cells, "
              "range-image culling, re-voxelisation) + third 3D pass on that batch, backward, Adam") if mopa else (
              "A2D2->SemanticKITTI-shape joint step per GPU (BASELINE configs[4]): "
              f"{B} source + {B} target scans/GPU per step, 120,000 pts/scan (64 beams x 1875 azimuths, scale 20), 10 classes, "
              "Net2DSeg(UNetResNet34, 302x480) + Net3DSeg(SCN UNet), CE + cross-modal KL, backward, Adam") if kitti else (
              "Full xMUDA 2D+3D joint step + xModalKL (BASELINE configs[2]): "
              f"{B} source + {B} target scans/GPU per step, Net2DSeg(UNetResNet34, 302x480) + Net3DSeg(SCN UNet, 34,880 pts), "
              "CE + cross-modal KL, backward, Adam") if joint else (
              "Net3DSeg SCN-UNet only (BASELINE configs[1]): geometry+fwd+CE+bwd+Adam, "
              f"bs={B} synthetic nuScenes-shape scans/GPU (34,880 pts each)")
        line = {
            "metric": "scans/sec (joint 2D+3D train step) at 1/2/4/8 MI355X; sparse-conv HBM GB/s",
            "value": round(world * scans_per_step * args.steps / elapsed, 3), "unit": "scans/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "setup_steps": setup_steps, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "scans_per_step_per_gpu": scans_per_step, "global_batch": scans_per_step * world,
                       "points_per_scan": shape["beams"] * shape["azimuths"], "num_classes": NC, "image": "302x480",
                       "parallelism": f"dp{world}",
                       "collective": "one flat fp32 gradient all-reduce per network per step (RCCL), 3D one overlapped with the 2D backward"
                       if world > 1 else ("one-rank process group, same collectives (MOPA_FORCE_COLLECTIVES=1)" if multi else "none (1 rank)"),
                       # what actually ran: backend of the process group, all-reduces issued per timed + warm-up step (FlatAdam counts
                       # them), and which passes of the stride-1 3x3 convolutions use Winograd F(4x4) (fp32, ~1e-5 relative per
                       # layer; "dgrad,wgrad" = exact-product forward pass)
                       "backend": (dist.get_backend() if multi else None),
                       "allreduces_per_step": round(sum(o.n_collectives for o in opts) / max(1, steps_run[0]), 2),
                       "gradient_buckets_2d_bytes": bucket_bytes,
                       "winograd_f4_roles": ",".join(f4_roles()) if joint else None,
                       "winograd_f4_one_kernel_roles": (",".join(one_kernel_roles()) or None) if joint else None,
                       "gradient_error_vs_fp64": gradient_error_vs_fp64() if joint else None,
                       "scn_executor": "native (one C-ABI call per pass; bracketed steps walk the program from Python)" if native_default
                       else "python walk (MOPA_SCN_NATIVE=0)",
                       "net3d_pass": (None if not joint else "source + target scans as ONE sparse tensor (Net3DSeg bn_group_points: BatchNorm per domain "
                                      "on row ranges)" if (pair_mode and pair_3d) else "one pass per domain"),
                       "net2d_pass": (None if not joint else
                                      "source + target images in ONE pass of the 2D network (Net2DSeg bn_groups=2: BatchNorm statistics, "
                                      "running-statistics updates and dropout masks per domain, in the reference's call order)" if pair_mode
                                      else "one pass per domain (MOPA_BENCH_PAIR=0 or a stream configuration without the side stream)"),
                       "net2d_executor": (None if not joint else
                                          f"HIP-graph replay of the backbone ({dense2d_mod.GRAPH_STATS['forward_replays']} forward / "
                                          f"{dense2d_mod.GRAPH_STATS['backward_replays']} backward replays in this process; heads and "
                                          "bracketed steps eager)" if graph2d_default else
                                          f"native command list (csrc/exec2d.hip: the recorded backbone pass replayed in one C-ABI call; "
                                          f"{dense2d_mod.GRAPH_STATS['forward_replays']} forward / {dense2d_mod.GRAPH_STATS['backward_replays']} backward "
                                          "replays in this process; heads and bracketed steps walk it from Python)" if native2d_default else
                                          "python walk (MOPA_NATIVE_2D=0)")},
            "iterations_per_s": round(world * args.steps / elapsed, 3),
            "value_with_host_inputs": None if host_value is None else round(host_value[0], 3),
            "value_two_calls_per_domain": None if two_call_value is None else round(two_call_value[0], 3),
            "two_calls_note": None if two_call_value is None else (
                f"{two_call_value[1]} further steps in the reference's loop order -- model(source batch), then model(target batch), two passes per "
                "network and iteration (train_xmuda_mopa.py:342-343,426-427) -- without the bn_groups / bn_group_points extension `value` uses"),
            "host_inputs_note": None if host_value is None else (
                f"{host_value[1]} further steps with coords / feats / labels / images handed over as HOST tensors and img_indices as "
                "numpy arrays (the reference's collate output), uploaded on a copy stream beside the compute; not part of `value`"),
            "roofline": roof,
            "roofline_wgrad": roof_wgrad,
            "roofline_sparse_conv": sp,
            # fabric-side bytes of ONE step, all kernels (Infinity-Cache hits included): the committed PMC passes of this workload's
            # command (profiles/make_final.sh -> traffic.py), never measured in this run -- the file and its commit say which build
            "hbm_GB_per_step": hbm_step[0], "hbm_GB_per_step_source": hbm_step[1],
            "peak_device_memory_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),   # caching-allocator high-water mark of this rank
        }
        print(f"[bench] timed {args.steps} steps in {elapsed:.3f}s (host enqueue {t_enqueued:.3f}s)", file=sys.stderr, flush=True)
        if not args.no_cpu_baseline and world == 1:  # CPU baseline: rank 0 at N=1 only
            t_cpu = time.perf_counter()
            line["cpu_baseline"] = cpu_baseline_joint(model2d, model3d, shape=shape) if joint else cpu_baseline_3d(model3d)
            print(f"[bench] cpu baseline took {time.perf_counter() - t_cpu:.1f}s", file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
