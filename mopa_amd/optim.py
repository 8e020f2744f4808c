"""Flat-buffer Adam + data-parallel gradient all-reduce.

All parameters of a network become views into ONE fp32 buffer and their ``.grad`` views into one flat gradient
buffer: one HIP launch updates the model (``mopa_adam_flat``) and the same flat buffer is what RCCL all-reduces
over xGMI -- one collective per network per iteration.  Semantics of ``torch.optim.Adam`` as the
reference configures it (``mopa/common/solver/build.py:7-21``, yaml ``OPTIMIZER: TYPE Adam, BASE_LR 1e-3``).

``FlatAdam`` IS a ``torch.optim.Optimizer``: the reference hands its optimizer to ``build_scheduler`` (a
``MultiStepLR`` that rewrites ``param_groups[0]["lr"]``, ``mopa/common/solver/build.py:24-47``,
``train_xmuda_mopa.py:135-136``) and to ``CheckpointerV2`` which calls ``optimizer.state_dict()`` /
``load_state_dict()`` (``mopa/common/utils/checkpoint.py:48-49,76-77``).  ``state_dict()`` has exactly the layout of
``torch.optim.Adam`` (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``, one param group), so optimizer
checkpoints move between the reference and this build in both directions.
"""
from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist

from ._lib import GRAD_DONE_HOOKS, WEIGHTS_EPOCH, call, ptr, stream

# MOPA_FORCE_COLLECTIVES=1: issue the all-reduce in a one-rank process group too (exercises the RCCL path on a 1-GPU box)
_FORCE_COLLECTIVES = os.environ.get("MOPA_FORCE_COLLECTIVES") == "1"


class _Works:
    """Several asynchronous collectives behind the one `work.wait()` the callers use."""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, checkpoint_shapes=None):
        """``checkpoint_shapes``: optional list (one entry per parameter, ``None`` = the parameter's own shape) of the shapes
        the moment tensors take in ``state_dict()`` -- ``mopa_amd.models.scn_unet.checkpoint_shapes(model)`` gives
        SparseConvNet's 4-D conv-weight layout so that a reference-side ``torch.optim.Adam.load_state_dict`` of our
        checkpoint pairs every moment with a parameter of the same shape."""
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FlatAdam got no parameters")
        if isinstance(params[0], dict):
            raise ValueError("FlatAdam takes one flat list of parameters (the reference builds a single param group)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False))
        self.params = params
        dev = params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]  # keep every view 16-byte aligned
        self.n = sum(sizes)
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self._slices = []
        off = 0
        with torch.no_grad():
            for p, sz in zip(params, sizes):
                view = self.flat[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                self._slices.append((off, p.numel()))
                off += sz
        self._attach_grads()
        self.t = 0
        self.n_collectives = 0        # all-reduces issued so far, and the backend of the last one (bench.py / tests report them)
        self.collective_backend = None
        self._ckpt_shapes = list(checkpoint_shapes) if checkpoint_shapes is not None else [None] * len(params)
        if len(self._ckpt_shapes) != len(params):
            raise ValueError("checkpoint_shapes must have one entry per trainable parameter")

    # ---- reference-facing conveniences (kept from round 1)
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, v):
        self.param_groups[0]["lr"] = v

    def _grad_view(self, i):
        off, n = self._slices[i]
        return self.grad[off:off + n].view_as(self.params[i])

    def _attach_grads(self):
        self._gviews = [self._grad_view(i) for i in range(len(self.params))]
        for p, g in zip(self.params, self._gviews):
            p.grad = g

    def _check_grads(self):
        """Every ``p.grad`` must still be the view of the flat gradient buffer the HIP kernels accumulate into.
        ``model.zero_grad()`` (torch's default sets grads to None) or a hook that replaces ``.grad`` would otherwise make
        ``step`` and ``all_reduce`` read a stale flat buffer without any error.  A detached slice is STALE -- whoever dropped
        the view meant "zero", and nothing has accumulated into the slice since -- so the parameter's own ``.grad`` is the
        truth: its value replaces the slice (``None`` = no gradient this iteration = zeros), and the view is re-attached."""
        for i, (p, gv) in enumerate(zip(self.params, self._gviews)):
            g = p.grad
            if g is gv:      # autograd accumulates in place: the tensor object we attached is still the gradient (~0.1 us per check)
                continue
            view = self._gviews[i] = self._grad_view(i)
            if g is None:
                view.zero_()
            else:               # autograd attached a fresh tensor after the view was dropped: it holds everything accumulated since
                if g.shape != p.shape:
                    raise RuntimeError("FlatAdam: a parameter's .grad was replaced by a tensor of another shape")
                view.copy_(g)
            p.grad = view

    def zero_grad(self, set_to_none: bool = False):
        """In-place memset; the ``.grad`` views stay attached so the backward kernels accumulate into the flat buffer
        (``set_to_none`` is accepted for API compatibility and ignored: dropping the views would break that)."""
        self.grad.zero_()
        self._checked = False

    def _collectives_on(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_COLLECTIVES)

    def all_reduce(self, async_op=False):
        """Sum the flat gradient over ranks (RCCL when the process group backend is 'nccl').  With armed buckets (enable_buckets +
        arm_buckets): the buckets whose all-reduce the backward pass has issued already are only waited for (the current stream
        waits for the communication stream; the host does not block), the others are issued now."""
        if not self._collectives_on():
            return None
        self._check_grads()
        self._checked = True   # once per iteration (~0.1 ms of host time for 200 parameters)
        self.collective_backend = dist.get_backend()
        if self._armed:
            for b in range(len(self._buckets)):
                if not self._bucket_issued[b]:
                    self._issue_bucket(b)
            self._armed = False
            cur = torch.cuda.current_stream(self.grad.device) if self.grad.is_cuda else None
            works, self._bucket_works = self._bucket_works, []
            if async_op:
                return _Works(works)
            for w in works:
                w.wait()
            if cur is not None and self._comm is not None:
                cur.wait_stream(self._comm)
            return None
        self.n_collectives += 1
        return dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, async_op=async_op)

    # ---- gradient buckets: the flat buffer cut into contiguous ranges in BACKWARD order, each all-reduced as soon as the backward
    # pass has enqueued the last kernel that writes into it -- the collective of the layers that finish first (decoder, layer4)
    # runs on a communication stream under the rest of the backward pass.  One flat buffer stays the optimizer's view.
    _armed = False
    _buckets = ()
    _comm = None

    def enable_buckets(self, n_buckets=4, extra_streams=None):
        """Cut the flat gradient buffer into `n_buckets` contiguous ranges of about equal size; bucket 0 holds the LAST parameters
        (their gradients are final first).  `extra_streams`: callable -> streams besides the current one that gradient kernels
        run on (the 2D weight-gradient stream); a bucket's collective is ordered behind all of them.  Takes effect for a
        backward pass only after arm_buckets()."""
        n_par = len(self.params)
        target = self.n / max(1, n_buckets)
        cuts, acc = [n_par], 0
        for i in range(n_par - 1, -1, -1):     # walk the parameters in backward order
            acc += self._slices[i][1]
            if acc >= target and len(cuts) < n_buckets and i > 0:
                cuts.append(i)
                acc = 0
        cuts.append(0)
        self._buckets = []                     # (first parameter, one past the last, flat offset lo, hi)
        for hi_p, lo_p in zip(cuts[:-1], cuts[1:]):
            if lo_p == hi_p:
                continue
            lo = self._slices[lo_p][0]
            hi = self.n if hi_p == n_par else self._slices[hi_p][0]
            self._buckets.append((lo_p, hi_p, lo, hi))
        self._bucket_of = {}                   # id(parameter) -> (bucket, parameter index)
        for b, (lo_p, hi_p, _, _) in enumerate(self._buckets):
            for i in range(lo_p, hi_p):
                self._bucket_of[id(self.params[i])] = (b, i)
        self._extra_streams = extra_streams
        self._comm = torch.cuda.Stream(device=self.grad.device) if self.grad.is_cuda else None
        self._bucket_works, self._bucket_issued, self._bucket_left = [], [False] * len(self._buckets), [0] * len(self._buckets)
        return [(hi - lo) * 4 for _, _, lo, hi in self._buckets]

    def disable_buckets(self):
        """Back to one collective per iteration; drops this optimizer's entries from the GradSink hook table."""
        for p in self.params:
            if GRAD_DONE_HOOKS.get(id(p)) == self._grad_done:
                del GRAD_DONE_HOOKS[id(p)]
        self._buckets, self._armed = (), False

    def arm_buckets(self):
        """Call right before the LAST backward pass of the iteration (gradients of earlier passes accumulate first: the reference
        runs the source and the target half before the optimizer step, train_xmuda_mopa.py:342-427)."""
        if not self._buckets or not self._collectives_on():
            return False
        if os.environ.get("MOPA_DIRECT_GRADS", "1") == "0":
            # without in-place gradients autograd's AccumulateGrad adds a pass's gradients to .grad only AFTER Function.backward has
            # returned -- after the bucket's all-reduce would have been enqueued: the collective would race with that add
            raise RuntimeError("FlatAdam.arm_buckets needs the gradients written in place (MOPA_DIRECT_GRADS=0 is set): "
                               "use all_reduce() after backward instead")
        self._check_grads()
        self._checked = True
        self._armed = True
        self._bucket_works = []
        self._bucket_issued = [False] * len(self._buckets)
        self._bucket_left = [hi_p - lo_p for lo_p, hi_p, _, _ in self._buckets]
        self._done_ids = set()
        for p in self.params:
            GRAD_DONE_HOOKS[id(p)] = self._grad_done
        return True

    def _grad_done(self, p, final=False):
        """GradSink: the kernels writing p's gradient in this backward pass are enqueued (reported once per parameter: a parameter
        two kernels write -- the 2D head's point and pixel parts -- is reported after the first; its bucket also holds the decoder
        layers that follow both in stream order, and a bucket that went out before a late writer is refused below)."""
        if not self._armed:
            return
        b, i = self._bucket_of.get(id(p), (None, None))
        if b is None or self.params[i] is not p:   # (an id re-issued to another tensor after this optimizer's parameter died)
            return
        if id(p) in self._done_ids:
            if self._bucket_issued[b] and not final:
                raise RuntimeError("FlatAdam buckets: a gradient was written after its bucket's all-reduce had been issued")
            return
        self._done_ids.add(id(p))
        self._bucket_left[b] -= 1
        # buckets go out in order (every rank issues the same sequence of collectives): bucket b only once b - 1 is out
        while b < len(self._buckets) and self._bucket_left[b] <= 0 and not self._bucket_issued[b] and (b == 0 or self._bucket_issued[b - 1]):
            self._issue_bucket(b)
            b += 1

    def _issue_bucket(self, b):
        _, _, lo, hi = self._buckets[b]
        self._bucket_issued[b] = True
        self.n_collectives += 1
        self.collective_backend = dist.get_backend()
        view = self.grad[lo:hi]
        if self._comm is None:
            self._bucket_works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
            return
        self._comm.wait_stream(torch.cuda.current_stream(self.grad.device))
        for st in (self._extra_streams() if self._extra_streams is not None else ()):
            if st is not None:
                self._comm.wait_stream(st)
        with torch.cuda.stream(self._comm):
            self._bucket_works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0, closure=None):
        """``step()`` / ``step(grad_scale)`` / ``step(closure)``: torch's convention puts the closure first, so a callable in
        the first position is taken as the closure.  ``grad_scale`` multiplies the gradient inside the update kernel
        (``1 / world_size`` after a sum all-reduce, times the rank's loss weight)."""
        if callable(grad_scale):
            closure, grad_scale = grad_scale, 1.0
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        else:
            loss = None
        if not getattr(self, "_checked", False):
            self._check_grads()
        self._checked = False
        if self.flat.device.type != "cuda":
            raise RuntimeError("FlatAdam.step needs the HIP extension on an MI355X (no CPU fallback)")
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        self.t += 1
        call("mopa_adam_flat", ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n,
             float(g["lr"]), b1, b2, g["eps"], g["weight_decay"], 1.0 - b1 ** self.t, math.sqrt(1.0 - b2 ** self.t),
             grad_scale, stream())
        WEIGHTS_EPOCH[0] += 1   # cached weight re-layouts (dense2d / sparse3d) are stale now
        return loss

    # ---- checkpointing in torch.optim.Adam's layout (mopa/common/utils/checkpoint.py:48-49,76-77)
    def state_dict(self):
        state = {}
        if self.t > 0:
            for i, (off, n) in enumerate(self._slices):
                shape = self._ckpt_shapes[i] or self.params[i].shape
                state[i] = {"step": torch.tensor(float(self.t)),
                            "exp_avg": self.exp_avg[off:off + n].view(shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(shape).clone()}
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        groups[0]["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("FlatAdam.load_state_dict: expected one param group with %d parameters" % len(self.params))
        for k, v in groups[0].items():
            if k != "params":
                self.param_groups[0][k] = tuple(v) if k == "betas" else v
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for key, st in sd["state"].items():
            i = int(key)
            off, n = self._slices[i]
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FlatAdam.load_state_dict: parameters with different step counts (one fused update per model)")
        self.t = steps.pop() if steps else 0
