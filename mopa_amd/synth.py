"""Seeded synthetic scans in the reference's batch layout (no dataset needed).

Generator spec: SURVEY.md section 8(d).  The batch dict has exactly the keys
the reference's ``collate_scn_base`` produces for the hot path
(``mopa/data/collate.py:182-186,233-235,271-273``):

    x            [locs (sumN,4) int64 CPU [x,y,z,batch], feats (sumN,1) f32]
    img          (B,3,H,W) f32 in [0,1)
    img_indices  list of B (N_b,2) int64 numpy arrays [row v, col u]
    seg_label    (sumN,) int64 with -100 = ignore
    pseudo_label_2d / pseudo_label_3d  (sumN,) int64 with -100
    sam_mask_ls  list of B (H,W) int32 tensors, ids in {-100} U [0,255]

Voxelisation follows ``augment_and_scale_3d`` with no augmentation
(``mopa/data/utils/augmentation_3d.py:48-52``): round(points*scale) - min.
"""
from __future__ import annotations

import numpy as np
import torch

NUSCENES = dict(beams=32, azimuths=1090, el_lo=-30.0, el_hi=10.0, height=1.84, classes=5)
KITTI = dict(beams=64, azimuths=1875, el_lo=-24.8, el_hi=2.0, height=1.73, classes=10)


def lidar_points(seed: int, shape: dict = NUSCENES) -> np.ndarray:
    """One synthetic LiDAR sweep, (beams*azimuths, 3) float32, beam-major order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nb, na, h = shape["beams"], shape["azimuths"], shape["height"]
    el = np.deg2rad(np.linspace(shape["el_lo"], shape["el_hi"], nb))
    az = -np.pi + 2 * np.pi * np.arange(na) / na
    sector = np.clip(rng.lognormal(np.log(15.0), 0.6, 64), 3.0, 70.0)
    obstacle = sector[(np.arange(na) * 64) // na]  # (na,)
    tan_el = np.tan(el)[:, None]  # (nb,1)
    with np.errstate(divide="ignore"):
        ground = np.where(tan_el < 0, h / -tan_el, np.inf)  # (nb,1)
    r = np.minimum(np.minimum(ground, obstacle[None, :]), 70.0)  # (nb,na)
    on_ground = ground <= np.minimum(obstacle[None, :], 70.0)
    z = np.where(on_ground, -h, r * tan_el)
    r = r + rng.normal(0.0, 0.02, r.shape)
    z = z + rng.normal(0.0, 0.02, z.shape)
    x = r * np.cos(az)[None, :]
    y = r * np.sin(az)[None, :]
    return np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)


def voxelize(points: np.ndarray, scale: int = 20) -> np.ndarray:
    """augment_and_scale_3d without augmentation -> int64 (N,3) voxel coords."""
    coords = np.round(points.astype(np.float32) * scale)
    coords -= coords.min(0)
    return coords.astype(np.int64)


def sam_mask(rng, H: int, W: int, seeds: int = 60) -> np.ndarray:
    """Jittered Voronoi segmentation, ids 1..seeds; FOV rows and big masks -> -100."""
    sy = rng.integers(0, H, seeds)
    sx = rng.integers(0, W, seeds)
    yy, xx = np.mgrid[0:H, 0:W]
    yj = (yy // 8) * 8 + rng.integers(0, 8, ((H + 7) // 8, (W + 7) // 8)).repeat(8, 0).repeat(8, 1)[:H, :W]
    xj = (xx // 8) * 8 + rng.integers(0, 8, ((H + 7) // 8, (W + 7) // 8)).repeat(8, 0).repeat(8, 1)[:H, :W]
    d = (yj[..., None] - sy) ** 2 + (xj[..., None] - sx) ** 2
    ids = (np.argmin(d, -1) + 1).astype(np.int32)
    ids[: min(120, H // 3)] = -100
    area = np.bincount(ids[ids >= 0], minlength=seeds + 2)
    big = np.nonzero(area >= 0.1 * H * W)[0]
    ids[np.isin(ids, big)] = -100
    return ids


def make_scan(seed: int, H: int = 302, W: int = 480, shape: dict = NUSCENES, scale: int = 20,
              num_classes: int | None = None):
    rng = np.random.Generator(np.random.PCG64(seed + 7_000_000))
    C = num_classes or shape["classes"]
    pts = lidar_points(seed, shape)
    coords = voxelize(pts, scale)
    n = coords.shape[0]
    label = rng.integers(0, C, n).astype(np.int64)
    label[rng.random(n) < 0.1] = -100
    pl2 = rng.integers(0, C, n).astype(np.int64)
    pl2[rng.random(n) < 0.5] = -100
    pl3 = rng.integers(0, C, n).astype(np.int64)
    pl3[rng.random(n) < 0.5] = -100
    return dict(
        coords=coords,
        feats=np.ones((n, 1), np.float32),
        img=rng.random((3, H, W), dtype=np.float32),
        img_indices=np.stack([rng.integers(0, H, n), rng.integers(0, W, n)], 1).astype(np.int64),
        seg_label=label,
        pseudo_label_2d=pl2,
        pseudo_label_3d=pl3,
        sam_mask=sam_mask(rng, H, W),
    )


def collate(scans: list) -> dict:
    """Same layout as collate_scn_base (collate.py:182-186,233-235)."""
    locs, feats = [], []
    for b, s in enumerate(scans):
        c = torch.from_numpy(s["coords"])
        locs.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int64)], 1))
        feats.append(torch.from_numpy(s["feats"]))
    return {
        "x": [torch.cat(locs, 0), torch.cat(feats, 0)],
        "img": torch.stack([torch.from_numpy(s["img"]) for s in scans]),
        "img_indices": [s["img_indices"] for s in scans],
        "seg_label": torch.cat([torch.from_numpy(s["seg_label"]) for s in scans]),
        "pseudo_label_2d": torch.cat([torch.from_numpy(s["pseudo_label_2d"]) for s in scans]),
        "pseudo_label_3d": torch.cat([torch.from_numpy(s["pseudo_label_3d"]) for s in scans]),
        "sam_mask_ls": [torch.from_numpy(s["sam_mask"]) for s in scans],
    }


def make_batch(batch_size: int, rank: int = 0, first: int = 0, **kw) -> dict:
    """Scan i of rank r uses seed 1000*r + i."""
    return collate([make_scan(1000 * rank + first + i, **kw) for i in range(batch_size)])
