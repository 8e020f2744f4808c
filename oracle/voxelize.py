"""Oracle: voxel coordinates as the reference computes them (numpy) -- TEST INFRASTRUCTURE ONLY.

Restates ``augment_and_scale_3d`` after the rotation (``mopa/data/utils/augmentation_3d.py:48-59``) and the dataset's
cast + filter (``mopa/data/nuscenes/nuscenes_dataloader.py:419-424``).  PINNED by fixture G4 (reference outputs, with
the reference's own rotated points and replayed translation draws).
"""
import numpy as np


def voxel_coords(aug_points, scale, full_scale=4096, transl_u=None):
    c = np.round(np.asarray(aug_points, np.float32) * np.float32(scale))
    c -= c.min(0)
    if transl_u is not None:
        t = (np.float32(full_scale) - c.max(0) - np.float32(0.001)).astype(np.float32)
        c = (c.astype(np.float64) + np.clip(t, 0, None).astype(np.float64) * np.asarray(transl_u, np.float64)).astype(np.float32)
    ci = c.astype(np.int64)
    keep = (ci.min(1) >= 0) & (ci.max(1) < full_scale)
    return ci, keep
