"""Step helpers for the joint 2D+3D iteration (the caller-side loop stays the user's, like the reference's train scripts).

``DualStream`` runs the two networks of one domain concurrently: they are independent until the losses (each
cross-modal KL detaches the other modality, ``mopa/train/train_xmuda_mopa.py:389-398``), the 3D branch is a chain of
small latency-bound kernels and the 2D branch of large FMA-bound ones, so a second HIP stream hides most of the 3D time.
Autograd replays every node on the stream it was recorded on, so the 3D backward overlaps the 2D backward too.
Results are bit-identical to sequential execution (same kernels, same per-network order; checked in tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def global_mean_weight(n_local: int, group=None) -> float:
    """Weight that turns this rank's per-rank loss MEAN into its share of the mean over the global batch.

    The reference is single-process: its losses are means over all points of the batch
    (``mopa/train/train_xmuda_mopa.py:354-363,389-398``).  Data-parallel ranks each average over their own N_r points and the
    gradients are summed and divided by the world size (``FlatAdam.step(grad_scale=w / world)``), which equals the global
    mean only when every rank holds the same number of points.  In general rank r must be weighted by
    ``w_r = N_r * world / sum_r N_r`` -- one float per rank per iteration, exchanged here.  The count is
    known on the host as soon as the batch is collated, so call this from the loader side, not between kernels."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 1.0
    t = torch.tensor([float(n_local)], dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    total = float(t.item())
    return float(n_local) * dist.get_world_size(group) / total if total > 0 else 1.0


def merge_domains_2d(img_s, img_t, pix_s, pix_t):
    """The source and the target batch of one iteration as ONE `Net2DSeg` batch (`bn_groups = 2`: BatchNorm statistics, running
    updates and dropout masks per domain in the reference's call order, train_xmuda_mopa.py:342,426; every convolution on both).
    img_*: (B,3,H,W) of equal size; pix_*: `Net2DSeg.pack_indices` of each domain's img_indices (int32 rows of the /16-padded
    feature map).  -> data_batch for `model_2d(...)`; its outputs hold the source's points / images first:
    `out["seg_logit"][:pix_s.numel()]`, `out["seg_logit_all"][:img_s.shape[0]]`."""
    if img_s.shape[1:] != img_t.shape[1:] or img_s.shape[0] != img_t.shape[0]:
        raise ValueError(f"the two domains need equal batch and image sizes, got {tuple(img_s.shape)} and {tuple(img_t.shape)}")
    B, _, H, W = img_s.shape
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    return {"img": torch.cat([img_s, img_t]), "point_pix_2d": torch.cat([pix_s, pix_t + B * Hp * Wp]), "img_indices": None,
            "bn_groups": 2}


def merge_domains_3d(batches, scans_per_batch):
    """Two or three batches of scans (source, target[, the VGI batch]) as ONE `Net3DSeg` batch: `batches` = [(locs (N_k,4) int64
    [x,y,z,scan], feats (N_k,C)), ...] with scan indices 0 .. scans_per_batch[k]-1 each; the k-th batch's scan indices are moved
    behind the earlier ones' and `bn_group_points` marks the boundaries (BatchNorm per batch on row ranges, in order).
    -> data_batch for `model_3d(...)`; outputs are per point in the order given (`out["seg_logit"][:N_0]`, `[N_0:N_0+N_1]`, ...)."""
    if not 2 <= len(batches) <= 3 or len(scans_per_batch) != len(batches):
        raise ValueError("merge_domains_3d takes two or three batches and one scan count per batch")
    locs, feats, cuts, first = [], [], [], 0
    for (lc, ft), nb in zip(batches, scans_per_batch):
        if first:
            lc = lc.clone()
            lc[:, 3] += first
        locs.append(lc)
        feats.append(ft)
        first += int(nb)
        cuts.append(int(lc.shape[0]) + (cuts[-1] if cuts else 0))
    return {"x": [torch.cat(locs), torch.cat(feats)], "bn_group_points": cuts[0] if len(cuts) == 2 else cuts[:-1]}


def freeze_host_heap() -> int:
    """Call once after the models, optimizers and the first iterations exist (the trainer: after building everything at
    ``train_xmuda_mopa.py:135-170``, before the iteration loop).  ``import torch`` and the model graph leave ~2 M tracked
    Python objects behind; every full (generation-2) pass of the cyclic collector walks all of them -- 70-100 ms on this
    host, during which nothing is enqueued and a launch-bound step (3D-only: 4.8 ms) idles the GPU for 15-20 steps' worth
    of time (``MOPA_BENCH_STEP_TIMES=1 python bench.py --workload 3d`` prints the passes).  ``gc.freeze()`` moves what
    exists now into the permanent generation: later passes only walk what the iterations allocate.  Nothing is disabled --
    cycles created afterwards are still collected.  Returns the number of objects frozen."""
    import gc
    gc.collect()
    gc.freeze()
    return gc.get_freeze_count()


class Timeline:
    """HIP-event marks on named streams (diagnostics): `mark(name, stream)` records, `report()` prints the mean offset of every
    mark from the first mark of its step in ms."""

    def __init__(self):
        self.steps = []

    def begin(self):
        self.steps.append([])

    def mark(self, name, stream=None):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        self.steps[-1].append((name, ev))

    def report(self):
        import collections
        acc = collections.OrderedDict()
        for marks in self.steps:
            occ = collections.Counter()
            for name, ev in marks:
                occ[name] += 1
                acc.setdefault((name, occ[name]), []).append(marks[0][1].elapsed_time(ev))
        return [(f"{n}#{k}", sum(v) / len(v)) for (n, k), v in acc.items()]


class DualStream:
    def __init__(self, device, order_2d_first: bool = False):
        self.device = torch.device(device)
        # (a high-priority side stream was measured in round 3: 325-329 scans/s either way -- the switch is gone)
        self.side = torch.cuda.Stream(device=self.device)
        self.order_2d_first = order_2d_first
        self.timeline = None

    def on_side(self, *tensors_from_main, after=None):
        """Context: run what follows (e.g. the 3D losses, so that their backward -- and with it the whole 3D backward -- is queued
        on the side stream and does not wait behind the 2D backward) on the side stream, ordered after everything queued on the
        current stream so far -- or, with ``after`` (an event recorded on the current stream), only after that point: work
        enqueued on the current stream behind the event (e.g. the 2D backward) then runs beside the block instead of in front of
        it, which matters when the block holds host round trips (the VGI of the MoPA iteration).  `tensors_from_main` are
        recorded as used by the side stream."""
        main = torch.cuda.current_stream(self.device)
        if after is not None:
            self.side.wait_event(after)
        else:
            self.side.wait_stream(main)
        for t in tensors_from_main:
            if torch.is_tensor(t):
                t.record_stream(self.side)
        return torch.cuda.stream(self.side)

    def forward(self, model_2d, model_3d, batch_2d: dict, batch_3d: dict, inputs_ready=None):
        """-> (preds_2d, preds_3d); both are safe to use on the current stream when this returns.

        ``inputs_ready``: an event after which the 3D coordinates are resident (e.g. recorded by the loader's copy stream).
        With it the voxel geometry -- which depends on the coordinates only and holds the 3D branch's two host syncs -- is
        built on the side stream BEFORE that stream is ordered behind the main stream, so the host does not stall on the
        previous half's 2D backward and keeps enqueueing ahead of the device.  Without it the build waits like the rest."""
        main = torch.cuda.current_stream(self.device)
        if self.order_2d_first:   # the main stream gets its (long) queue first; the 3D launches are enqueued while it runs
            self.side.wait_stream(main)
            preds_2d = model_2d(batch_2d)
        if inputs_ready is not None and batch_3d.get("geometry_3d") is None:
            with torch.cuda.stream(self.side):
                self.side.wait_event(inputs_ready)
                batch_3d = dict(batch_3d, geometry_3d=model_3d.net_3d.geometry(batch_3d["x"][0]))
        if not self.order_2d_first:
            self.side.wait_stream(main)
        if self.timeline is not None:   # diagnostics (bench.py MOPA_BENCH_TIMELINE=1): where the two forwards end
            self.timeline.mark("fwd2d_end", main)
        with torch.cuda.stream(self.side):
            preds_3d = model_3d(batch_3d)
        if not self.order_2d_first:
            preds_2d = model_2d(batch_2d)
        if self.timeline is not None:
            self.timeline.mark("fwd3d_end", self.side)
        main.wait_stream(self.side)
        for t in preds_3d.values():   # allocated from the side stream's pool, read by the loss kernels on the main stream:
            if torch.is_tensor(t):    # their memory must not return to the side stream before those reads are done
                t.record_stream(main)
        return preds_2d, preds_3d

    def geometry_ahead(self, model_3d, locs, inputs_ready, group_points=None):
        """Voxel geometry of `locs` for a 3D pass that will run on the CURRENT stream, built on the side stream behind
        `inputs_ready` only: its two host syncs then wait for the (short) side-stream queue instead of everything queued on
        the current stream.  Pass the result as ``batch["geometry_3d"]``."""
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.side):
            self.side.wait_event(inputs_ready)
            geom = model_3d.net_3d.geometry(locs, group_points=group_points)   # (group_points: Net3DSeg's "bn_group_points")
            built = torch.cuda.Event()
            built.record()
        cur.wait_event(built)
        geom.record_stream(cur)
        return geom

    def backward_on_side(self, loss, **kw):
        """``loss.backward()`` with the side stream current.  autograd ends a backward pass by making the stream that is current
        at the call wait for every stream the pass ran on; a 3D loss built under ``on_side()`` runs its backward on the side
        stream, so calling ``backward()`` from the main stream is a hidden join: the main stream idles until the whole 3D backward
        has finished (3.8 ms per half of the joint step, found with ``MOPA_BENCH_TIMELINE=1``; +7 % throughput once removed).
        Keep the 3D network's optimizer on the side stream as well (``with torch.cuda.stream(dual.side): opt3d.step()``), or
        ``join()`` before touching its gradients from the main stream."""
        with torch.cuda.stream(self.side):
            loss.backward(**kw)

    def join(self):
        """Call after the backward passes, before reducing / applying the 3D network's gradients."""
        torch.cuda.current_stream(self.device).wait_stream(self.side)
