"""Synchronised BatchNorm across data-parallel ranks (an option, off by default).

The reference trains in one process: every BatchNorm of both networks normalises over the WHOLE batch -- 2D over ``B*H*W``
pixels (``mopa/models/resnet34_unet.py:131-191`` through torchvision's ``BatchNorm2d``), 3D over every active row of the batch
(``scn.BatchNormReLU`` inside ``mopa/models/scn_unet.py:27-29``).  With scans sharded over ranks the default here is rank-local
statistics (throughput; documented deviation).  ``enable()`` switches every BatchNorm of both networks to global statistics:

    forward    rank-local (mean, M2, n) per channel -> ``all_gather`` (2C+1 doubles) -> Chan combination in rank order
    backward   rank-local (sum dz, sum dz*xhat)     -> ``all_reduce``  (2C doubles)   -> apply with the global row count

so that N ranks x (B/N scans) computes what one rank x B scans computes (``tests/test_gpu_syncbn.py``: logits, running
statistics and -- after the gradient all-reduce -- parameter gradients agree to fp32 round-off).  Two small collectives per
BatchNorm layer and direction (69 layers): use it for equivalence checks and tiny per-rank batches, not for throughput.
Evaluation mode never communicates.  The collectives run on the current stream's order (RCCL: its own stream, ordered by
events; gloo: staged through the host).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from ._lib import call, ptr, query, stream, workspace

_STATE = {"enabled": False, "group": None}


def enable(group=None) -> None:
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("mopa_amd.syncbn.enable() needs an initialised torch.distributed process group")
    _STATE["enabled"], _STATE["group"] = True, group


def disable() -> None:
    _STATE["enabled"], _STATE["group"] = False, None


def active() -> bool:
    if not _STATE["enabled"]:
        return False
    return dist.get_world_size(_STATE["group"]) > 1 or os.environ.get("MOPA_FORCE_COLLECTIVES") == "1"


class local_statistics:
    """``with syncbn.local_statistics():`` -- rank-local BatchNorm inside the block (e.g. a reference run in the same process)."""

    def __enter__(self):
        self.prev = _STATE["enabled"]
        _STATE["enabled"] = False

    def __exit__(self, *exc):
        _STATE["enabled"] = self.prev
        return False


def _ws(nbytes, dev):
    return workspace.get(max(int(nbytes), 256), dev)


def fwd(x, y, gamma, beta, rmean, rvar, momentum, eps, leak, act, res, stats):
    """Training-mode forward of one BatchNorm(+residual)(+ReLU) with global statistics.  Returns the gathered moments
    ``[world][2C+1]`` (double) -- the backward pass reads the global row count from them."""
    dev = x.t.device
    C = x.C
    group = _STATE["group"]
    world = dist.get_world_size(group)
    gathered = torch.empty(world * (2 * C + 1), dtype=torch.float64, device=dev)   # [world][2C+1], flat for gloo's all_gather
    if x.rows == 0:
        # a rank without rows at this layer still takes part in the collective (moments with n = 0 are skipped by the Chan
        # combination); raising here would leave the other ranks blocked in all_gather for ever
        dist.all_gather_into_tensor(gathered, torch.zeros(2 * C + 1, dtype=torch.float64, device=dev), group=group)
        # ... and still owns running statistics: updated from the global moments like on every other rank (stats-only call, rows = 0),
        # so that whichever rank writes the checkpoint holds the same BatchNorm buffers
        call("mopa_bn_act_fwd_sync", x.p, x.ld, y.p, y.ld, 0, C, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), momentum, eps,
             leak, int(act), None, 0, ptr(gathered), world, ptr(stats), stream())
        return gathered
    ws = _ws(query("mopa_bnrelu_rows_workspace_bytes", x.rows, C), dev)
    mine = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
    call("mopa_bn_sync_moments", x.p, x.ld, x.rows, C, ptr(mine), ptr(ws), ws.numel(), stream())
    dist.all_gather_into_tensor(gathered, mine, group=group)
    call("mopa_bn_act_fwd_sync", x.p, x.ld, y.p, y.ld, x.rows, C, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), momentum, eps,
         leak, int(act), res.p if res is not None else None, res.ld if res is not None else 0, ptr(gathered), world, ptr(stats),
         stream())
    return gathered


def bwd(dy, x, dx, stats, leak, act, ymask, dres, acc_dres, dgamma, dbeta, acc_params, acc_dx, gathered):
    dev = x.t.device
    C = x.C
    group = _STATE["group"]
    if x.rows == 0:   # no rows here: contribute zeros to the sums, nothing to apply (see fwd)
        dist.all_reduce(torch.zeros(2 * C, dtype=torch.float64, device=dev), op=dist.ReduceOp.SUM, group=group)
        if not acc_params:
            dgamma.zero_()
            dbeta.zero_()
        return
    ws = _ws(query("mopa_bnrelu_rows_workspace_bytes", x.rows, C), dev)
    sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
    call("mopa_bn_sync_bwd_sums", dy.p, dy.ld, x.p, x.ld, x.rows, C, ptr(stats), leak, int(act),
         ymask.p if ymask is not None else None, ymask.ld if ymask is not None else 0, ptr(dgamma), ptr(dbeta), int(acc_params),
         ptr(sums), ptr(ws), ws.numel(), stream())
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    coef = torch.empty(2 * C, dtype=torch.float32, device=dev)
    call("mopa_bn_act_bwd_sync", dy.p, dy.ld, x.p, x.ld, dx.p, dx.ld, x.rows, C, ptr(stats), leak, int(act),
         ymask.p if ymask is not None else None, ymask.ld if ymask is not None else 0,
         dres.p if dres is not None else None, dres.ld if dres is not None else 0, int(acc_dres), ptr(sums), ptr(gathered),
         gathered.numel() // (2 * C + 1), int(acc_dx), ptr(coef), stream())
