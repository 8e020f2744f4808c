"""Flat-buffer Adam + data-parallel gradient all-reduce.

All parameters of a network become views into ONE fp32 buffer and their ``.grad`` views into one flat gradient
buffer: one HIP launch updates the model (``mopa_adam_flat``) and the same flat buffer is what RCCL all-reduces
over xGMI -- one collective per network per iteration.  Semantics of ``torch.optim.Adam`` as the
reference configures it (``mopa/common/solver/build.py:7-21``, yaml ``OPTIMIZER: TYPE Adam, BASE_LR 1e-3``).
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist

from ._lib import WEIGHTS_EPOCH, call, ptr, stream


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam got no parameters")
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]  # keep every view 16-byte aligned
        self.n = sum(sizes)
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        off = 0
        with torch.no_grad():
            for p, sz in zip(self.params, sizes):
                view = self.flat[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                p.grad = self.grad[off:off + p.numel()].view_as(p)
                off += sz
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.t = 0

    def zero_grad(self):
        self.grad.zero_()  # hipMemsetAsync; the .grad views stay attached so autograd accumulates in place

    def all_reduce(self, async_op=False):
        """Sum the flat gradient over ranks (RCCL when the process group backend is 'nccl')."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, async_op=async_op)
        return None

    def step(self, grad_scale: float = 1.0):
        self.t += 1
        b1, b2 = self.betas
        if self.flat.device.type != "cuda":
            raise RuntimeError("FlatAdam.step needs the HIP extension on an MI355X (no CPU fallback)")
        call("mopa_adam_flat", ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n,
             self.lr, b1, b2, self.eps, self.weight_decay, 1.0 - b1 ** self.t, math.sqrt(1.0 - b2 ** self.t),
             grad_scale, stream())
        WEIGHTS_EPOCH[0] += 1   # cached weight re-layouts (dense2d) are stale now
