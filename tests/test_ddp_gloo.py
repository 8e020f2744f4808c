"""Data-parallel path on CPU: world_size 2 over gloo (the GPU path is the same code over RCCL, backend 'nccl')."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mopa_amd import synth
    from mopa_amd.optim import FlatAdam
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 2))]
    before = [p.detach().clone() for p in params]
    opt = FlatAdam(params, lr=1e-3)
    assert all(torch.equal(p, b) for p, b in zip(params, before))            # flattening keeps the values
    assert all(p.grad.data_ptr() >= opt.grad.data_ptr() for p in params)     # .grad are views of the flat buffer
    opt.zero_grad()
    for p in params:
        (p * (rank + 1)).sum().backward()                                    # rank-dependent gradient
    opt.all_reduce()                                                         # ONE collective for the whole model
    expect = float(sum(r + 1 for r in range(world)))
    ok = all(torch.allclose(p.grad, torch.full_like(p, expect)) for p in params)
    # each rank owns different scans (weak scaling: no data-path collective)
    mine = synth.lidar_points(1000 * rank)[:16]
    gathered = [torch.zeros(16, 3) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(mine))
    distinct = not torch.equal(gathered[0], gathered[1])
    try:
        opt.step()
        cpu_step_refused = False
    except RuntimeError:
        cpu_step_refused = True                                              # no CPU fallback for the update kernel
    # under a multi-rank process group the 2D weight gradients stay on the caller's stream (no third stream per rank)
    from mopa_amd import dense2d
    no_third_stream = dense2d.wgrad_stream("cuda:0") is None if os.environ.get("MOPA_WGRAD_STREAM") != "1" else True
    q.put((rank, ok and no_third_stream, distinct, cpu_step_refused))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, distinct, refused in res:
        assert ok and distinct and refused, (rank, ok, distinct, refused)


def _model_worker(rank, world, port, q):
    """Two ranks, UNEQUAL point counts, a real network gradient: the 3D oracle network (eval-mode BN, so scans do not couple)
    with its parameters living in FlatAdam's flat buffers.  rank-weighted (global_mean_weight) flat all-reduce / world must
    equal the single-process gradient of the mean loss over the union batch -- the reference's single-GPU semantics."""
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mopa_amd.optim import FlatAdam
    from mopa_amd.step import global_mean_weight
    from oracle import scn3d
    from oracle.params import det_tensor
    torch.set_num_threads(2)

    def cloud(r, n):
        rng = np.random.Generator(np.random.PCG64(50 + r))
        return rng.integers(0, 40, (n, 3)).astype(np.int64), torch.from_numpy(rng.integers(0, 5, n).astype(np.int64))

    def params():
        shapes = {"net_3d." + k: s for k, s in scn3d.unet_param_shapes(num_planes=3).items()}
        shapes.update({"linear.weight": (5, 16), "linear.bias": (5,), "linear2.weight": (5, 16), "linear2.bias": (5,)})
        P = {k: det_tensor(k, s, torch.float32) for k, s in shapes.items()}
        return {k: (torch.nn.Parameter(v) if "running" not in k else v) for k, v in P.items()}

    def loss_of(P, clouds):
        c = np.concatenate([np.concatenate([xyz, np.full((len(xyz), 1), b)], 1) for b, (xyz, _) in enumerate(clouds)])
        lab = torch.cat([l for _, l in clouds])
        out = scn3d.net3dseg_forward(P, scn3d.Geometry(c, 3), torch.ones(len(c), 1), training=False, num_planes=3)
        return torch.nn.functional.cross_entropy(out["seg_logit"], lab) + torch.nn.functional.cross_entropy(out["seg_logit2"], lab)

    sizes = [300, 700]                      # rank 0 holds 300 points, rank 1 holds 700
    mine = cloud(rank, sizes[rank])
    P = params()
    train = [p for p in P.values() if isinstance(p, torch.nn.Parameter)]
    opt = FlatAdam(train)          # parameters and their .grad are now views of the flat fp32 buffers
    opt.zero_grad()
    w = global_mean_weight(sizes[rank])
    assert abs(w - sizes[rank] * world / sum(sizes)) < 1e-12
    loss_of(P, [mine]).backward()
    opt.grad.mul_(w)
    opt.all_reduce()
    got = opt.grad / world
    ok, err = True, 0.0
    if rank == 0:
        Q = params()
        loss_of(Q, [cloud(0, sizes[0]), cloud(1, sizes[1])]).backward()
        ref = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4)) for p in Q.values() if isinstance(p, torch.nn.Parameter)])
        err = float((got - ref).abs().max() / ref.abs().max())
        ok = err < 1e-4            # fp32 on both sides; only the summation order over points differs
    q.put((rank, ok, err if rank == 0 else 0.0))
    dist.destroy_process_group()


def test_model_gradient_two_ranks_equals_single_process_union_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a plain process must start two ranks itself (the flag was dead).  Dry mode:
    process-group plumbing over gloo, no GPU work; rank 0 prints the JSON line with n_gpus = 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOPA_BENCH_DRY="1", MOPA_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout        # exactly one JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry"] is True and line["distinct_scans"] is True and line["loss_weight_rank0"] == 1.0


def _bucket_worker(rank, world, port, q):
    """Gradient buckets (FlatAdam.enable_buckets): the flat buffer all-reduced as 4 contiguous ranges, each issued from inside the
    backward pass when GradSink reports its last gradient kernel enqueued -- must equal the single flat all-reduce bit for bit."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mopa_amd._lib import GradSink
    from mopa_amd.optim import FlatAdam

    def make():
        g = torch.Generator().manual_seed(3)
        names = [f"layer{i}.weight" for i in range(11)]
        shapes = [(7, 5), (13,), (3, 3, 3), (64,), (2, 9), (31,), (5, 5), (17,), (4, 4, 4), (6,), (10, 3)]
        return names, [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]

    def fill(params):   # rank-dependent gradients, written in place like the HIP kernels do
        g = torch.Generator().manual_seed(100 + rank)
        return [torch.randn(p.shape, generator=g) for p in params]

    names, pa = make()
    _, pb = make()
    oa, ob = FlatAdam(pa, lr=1e-3), FlatAdam(pb, lr=1e-3)
    sizes = ob.enable_buckets(4)
    oa.zero_grad(); ob.zero_grad()
    ga = fill(pa)
    for p, g in zip(pa, ga):
        p.grad.copy_(g)
    oa.all_reduce()                                   # reference: ONE collective
    # bucketed: a "backward pass" that hands the parameters out in reverse order, two at a time, through GradSink
    armed = ob.arm_buckets()
    sink = GradSink(dict(zip(names, pb)), names)
    issued_during_backward = []
    for i in range(len(pb) - 1, -1, -2):
        group = [names[j] for j in (i, i - 1) if j >= 0]
        gs, acc = sink.take(*group)
        assert acc
        for n_, gbuf in zip(group, gs):
            gbuf.copy_(ga[names.index(n_)])
        issued_during_backward.append(sum(ob._bucket_issued))
    sink.returned()
    early = sum(ob._bucket_issued)
    ob.all_reduce()
    same = torch.equal(oa.grad, ob.grad)
    # a second iteration re-arms cleanly; un-armed passes use the single collective
    n_before = ob.n_collectives
    ob.zero_grad()
    ob.all_reduce()
    q.put((rank, armed and same, len(sizes), early, issued_during_backward, ob.n_collectives - n_before, oa.n_collectives))
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_equals_the_flat_one_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, nb, early, trace, later, flat_calls in res:
        assert ok, f"rank {rank}: bucketed result differs from the flat all-reduce"
        assert nb == 4 and early == 4, (nb, early)                # every bucket went out from inside the backward pass ...
        assert trace[0] == 0 and trace == sorted(trace) and trace[-1] >= 2, trace   # ... progressively, not all at the end
        assert later == 1 and flat_calls == 1
