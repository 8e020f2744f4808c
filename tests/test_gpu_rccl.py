"""The RCCL (`nccl` backend) path of the data-parallel step on ONE MI355X.

RCCL refuses two ranks on one device, so a 1-GPU box can only run a one-rank group -- which still goes through
`init_process_group("nccl", device_id=...)`, the flat gradient all-reduces (the 3D one asynchronously on the side stream), the
data-parallel stream configuration of bench.py and synchronised BatchNorm's collectives.  Both tests start CHILD processes
(never re-exec a process that has touched the GPU).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, timeout):
    env = dict(os.environ, MOPA_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    env.pop("MOPA_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("buckets", [4, 0])
def test_bench_joint_step_one_rank_over_rccl(buckets, monkeypatch):
    monkeypatch.setenv("MOPA_BENCH_BUCKETS", str(buckets))
    r = _launch([os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"], 900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])      # bench.py asserts a finite loss before it prints
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["backend"] == "nccl"                 # the process group is RCCL ...
    # ... and every step's gradients went through it: the 3D network's flat buffer as one collective (asynchronous, side stream), the 2D
    # network's as 4 buckets issued from inside its backward pass on a communication stream (or as one collective behind it)
    assert d["config"]["allreduces_per_step"] == (5.0 if buckets else 2.0), d["config"]["allreduces_per_step"]
    if buckets:
        bb = d["config"]["gradient_buckets_2d_bytes"]
        assert len(bb) == 4 and sum(bb) >= 4 * 23_614_794 and max(bb) < 2 * min(bb) + 8_000_000, bb
    assert d["roofline"] is not None and d["roofline_sparse_conv"] is not None


def test_syncbn_and_flat_allreduce_one_rank_over_rccl():
    r = _launch([os.path.join(ROOT, "tests", "_rccl_worker.py")], 900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
