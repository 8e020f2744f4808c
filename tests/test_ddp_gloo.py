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
