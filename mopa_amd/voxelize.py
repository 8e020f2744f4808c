"""Device voxeliser + collate: points (metres) -> the ``[x, y, z, batch]`` int64 rows the 3D branch consumes.

SURVEY.md section 8(f) rank 1.  Mirrors the arithmetic of ``augment_and_scale_3d`` after its rotation
(``mopa/data/utils/augmentation_3d.py:48-59``) plus the dataset's int64 cast and in-range filter
(``mopa/data/nuscenes/nuscenes_dataloader.py:419-424``) and the collate layout (``mopa/data/collate.py:183-185``).
The random decisions (rotation matrix, scale factor, translation draws) stay with the caller, exactly as the reference
draws them from numpy's global RNG; the device part is deterministic and bit-exact with the reference (fixture G4).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ._lib import call, ptr, query, stream, workspace


def voxelize_scan(points: torch.Tensor, scale: float, full_scale: int = 4096, transl_u=None, batch_index: int = 0):
    """points (N,3) fp32 on the GPU -> (coords (N',4) int64 [x,y,z,b], keep (N,) bool).

    ``transl_u``: the three ``np.random.rand(3)`` draws of the random translation (None = no translation).
    Points whose voxel falls outside ``[0, full_scale)`` are dropped like the dataset does; ``keep`` tells the caller
    which rows of the per-point side arrays (labels, image indices) survive.
    """
    if points.device.type != "cuda":
        raise RuntimeError("voxelize_scan needs points on the GPU (no CPU fallback)")
    pts = points.contiguous().float()
    n = pts.shape[0]
    coords = torch.empty(n, 4, dtype=torch.int64, device=pts.device)
    keep = torch.empty(n, dtype=torch.uint8, device=pts.device)
    ws = workspace.get(query("mopa_voxelize_workspace_bytes"), pts.device)
    u = (ctypes.c_double * 3)(*([float(x) for x in transl_u] if transl_u is not None else [0.0, 0.0, 0.0]))
    call("mopa_voxelize", ptr(pts), n, float(scale), int(full_scale), ctypes.addressof(u), int(transl_u is not None),
         int(batch_index), ptr(coords), ptr(keep), ptr(ws), ws.numel(), stream())
    keep = keep.bool()
    if not bool(keep.all()):  # rare: compaction only when something fell outside the field
        coords = coords[keep]
    return coords, keep


def collate_scans(point_sets, scale: float, full_scale: int = 4096, transl_us=None):
    """List of (N_b,3) GPU point tensors -> data_batch['x'] = [locs (sumN,4) int64, feats (sumN,1) ones]."""
    locs = []
    for b, pts in enumerate(point_sets):
        c, _ = voxelize_scan(pts, scale, full_scale, None if transl_us is None else transl_us[b], b)
        locs.append(c)
    locs = torch.cat(locs, 0)
    return [locs, torch.ones(locs.shape[0], 1, device=locs.device)]
