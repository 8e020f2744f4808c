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


def draw_rotation(noisy_rot=0.0, flip_x=0.0, flip_y=0.0, rot_z=0.0):
    """The 3x3 float32 matrix of ``augment_and_scale_3d`` (``augmentation_3d.py:26-46``), drawn from numpy's global RNG in the
    reference's order (randn(3,3), randint for flip_x, randint for flip_y, rand for rot_z); None when no option is on."""
    if not (noisy_rot > 0 or flip_x > 0 or flip_y > 0 or rot_z > 0):
        return None
    r = np.eye(3, dtype=np.float32)
    if noisy_rot > 0:
        r += np.random.randn(3, 3) * noisy_rot
    if flip_x > 0:
        r[0][0] *= np.random.randint(0, 2) * 2 - 1
    if flip_y > 0:
        r[1][1] *= np.random.randint(0, 2) * 2 - 1
    if rot_z > 0:
        theta = np.random.rand() * rot_z
        z = np.array([[np.cos(theta), -np.sin(theta), 0], [np.sin(theta), np.cos(theta), 0], [0, 0, 1]], dtype=np.float32)
        r = r.dot(z)
    return r


def rotate_points(points: torch.Tensor, rot) -> torch.Tensor:
    """points (N,3) fp32 on the GPU @ rot (3,3) float32 -> (N,3) fp32: the rotation / flip stage (``augmentation_3d.py:48-50``)."""
    pts = points.contiguous().float()
    if rot is None:
        return pts
    r = np.ascontiguousarray(np.asarray(rot, np.float32))
    out = torch.empty_like(pts)
    call("mopa_rotate_points_f32", ptr(pts), pts.shape[0], r.ctypes.data, ptr(out), stream())
    return out


def voxelize_scan(points: torch.Tensor, scale: float, full_scale: int = 4096, transl_u=None, batch_index: int = 0, rot=None):
    """points (N,3) fp32 on the GPU -> (coords (N',4) int64 [x,y,z,b], keep (N,) bool).

    ``rot``: the augmentation's 3x3 matrix (``draw_rotation``), applied on the device first; None = points are used as given.
    ``transl_u``: the three ``np.random.rand(3)`` draws of the random translation (None = no translation).
    Points whose voxel falls outside ``[0, full_scale)`` are dropped like the dataset does; ``keep`` tells the caller
    which rows of the per-point side arrays (labels, image indices) survive.
    """
    if points.device.type != "cuda":
        raise RuntimeError("voxelize_scan needs points on the GPU (no CPU fallback)")
    pts = rotate_points(points, rot)
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
