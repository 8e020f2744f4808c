"""Synchronised BatchNorm (mopa_amd.syncbn): 2 ranks x 1 scan == 1 process x 2 scans, on the GPU, full networks.

Two processes share cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the collectives are backend-agnostic
torch.distributed calls, the HIP kernels on both sides of them are what is tested).  The worker exits non-zero on any mismatch.
"""
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


def test_syncbn_two_ranks_equal_one_process_full_batch():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_syncbn_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
