"""Valid Ground-based Insertion (VGI) on the device -- the host mirror of ``mopa/data/mixmatch_ss.py``.

Same call shapes as the reference's ``check_overlap`` / ``point_mixmatch(insert_mode="ground")`` / ``post_process``
(``mopa/data/mixmatch_ss.py:215-331,47-212,458-559``; caller ``mopa/train/train_xmuda_mopa.py:483-555``), with the scan
resident on the GPU: everything that touches the N scan points or the search grid is a HIP kernel (``csrc/vgi.hip``), the
host keeps what the reference also does on the host with a handful of numbers -- the object's extents, numpy's global
RNG (``np.random.choice`` / ``rand``: same draws in the same order, so a seeded run reproduces the reference), the 4x4
placement matrix.  Semantics and parity: ``oracle/vgi.py`` (fixture G8 = the reference's outputs).  No CPU fallback.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ._lib import call, ptr, query, stream, workspace


def _dev_f32(pc, device):
    t = torch.as_tensor(pc)
    if t.device.type != "cuda":
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class OverlapMap:
    """Device state of one scan's search region: first-point volume, ground columns, and (per object) the free cells."""

    def __init__(self, pc_scan, voxel_size, search_range, z_min, front_axis, g_mask=None, device="cuda"):
        self.pts = _dev_f32(pc_scan, device)
        if self.pts.dim() != 2 or self.pts.shape[1] < 3:
            raise RuntimeError(f"pc_scan must be (N, >=3), got {tuple(self.pts.shape)}")
        dev = self.pts.device
        self.vs = float(voxel_size)
        self.sr = [int(search_range[0] / voxel_size), int(search_range[1] / voxel_size)]
        self.zmin_v = float(np.floor(z_min / voxel_size))
        self.front = front_axis
        if front_axis == "x":
            self.ox, self.oy = 0, -self.sr[1]
        elif front_axis == "y":
            self.ox, self.oy = -self.sr[0], 0
        else:
            raise ValueError("front_axis must be 'x' or 'y'")
        self.X, self.Y = 2 * self.sr[0], 2 * self.sr[1]
        self.ZR = query("mopa_vgi_zslots")
        self.zlo = -self.ZR // 2
        self.first = torch.empty(self.X * self.Y * self.ZR, dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.g_mask = None
        if g_mask is not None:
            self.g_mask = torch.as_tensor(np.asarray(g_mask).astype(np.uint8) if not torch.is_tensor(g_mask) else g_mask.to(torch.uint8)).to(dev).contiguous()
            if self.g_mask.numel() != self.pts.shape[0]:
                raise RuntimeError("g_mask must hold one flag per scan point")
        call("mopa_vgi_first_points", ptr(self.pts), self.pts.shape[1], self.pts.shape[0], float(np.float32(self.vs)), self.ox, self.oy,
             self.X, self.Y, self.zlo, ptr(self.g_mask), ptr(self.first), ptr(self.status), stream())
        self._ground2d = None

    def free_cells(self, pc_obj, z_max=None):
        """The overlap test for one object -> dict(free (Xo,Yo,Zo) uint8 on the device or None, extent, offset, ori_range)."""
        obj = np.asarray(pc_obj)[:, :3]
        ov = np.floor(obj / self.vs)                      # numpy keeps the object's dtype, like the reference
        zmax_v = self.zmin_v if z_max is None else z_max
        extent_z = np.max(ov, axis=0)[2] - np.min(ov, axis=0)[2] + 2
        sr2 = int(extent_z + zmax_v)
        Z = int(sr2 - self.zmin_v)
        extent = np.max(ov, axis=0) - np.min(ov, axis=0) + 1
        extent[0:2] = np.ceil(np.sqrt(np.square(extent[0]) + np.square(extent[1])))
        box = extent.astype(np.int32)
        offset = np.array([self.ox, self.oy, self.zmin_v], np.float64)
        out = dict(extent=extent.astype(np.float64), offset=offset, box=box, free=None)
        gz0 = int(self.zmin_v)
        if Z <= 0 or box[0] > self.X or box[1] > self.Y or box[2] > Z or gz0 < self.zlo or gz0 + Z > self.zlo + self.ZR:
            return out
        Xo, Yo, Zo = self.X - int(box[0]) + 1, self.Y - int(box[1]) + 1, Z - int(box[2]) + 1
        free = torch.empty(Xo, Yo, Zo, dtype=torch.uint8, device=self.pts.device)
        ws = workspace.get(query("mopa_vgi_box_free_workspace_bytes", self.X, self.Y, Z), self.pts.device)
        call("mopa_vgi_box_free", ptr(self.first), self.X, self.Y, self.zlo, gz0, Z, int(box[0]), int(box[1]), int(box[2]), ptr(free),
             ptr(ws), ws.numel(), stream())
        out["free"] = free
        return out

    def ground2d(self):
        if self._ground2d is None:
            if self.g_mask is None:
                raise RuntimeError("a ground mask is needed (the reference's offline g_indices, mixmatch_ss.py:386-395)")
            self._ground2d = torch.empty(self.X * self.Y, dtype=torch.uint8, device=self.pts.device)
            call("mopa_vgi_ground_cells", ptr(self.first), ptr(self.g_mask), self.X, self.Y, ptr(self._ground2d), stream())
        return self._ground2d

    def ground_cells_enqueue(self, fc, pc_obj, proj_matrix, image_size):
        """Centre filters + ground lookup, enqueued only -> (cells (X*Y,2) int32, ncell (1,), counts (2,)) device tensors."""
        free = fc["free"]
        obj = np.asarray(pc_obj)[:, :3]
        oc = (np.max(obj, axis=0) + np.min(obj, axis=0)) / 2
        ori_range = float(np.sqrt(np.square(oc[0]) + np.square(oc[1])))
        params = np.concatenate([fc["extent"], fc["offset"], [self.vs, ori_range], np.asarray(proj_matrix, np.float64).astype(np.float32).reshape(-1),
                                 [float(image_size[0]), float(image_size[1])]]).astype(np.float64)
        dev = self.pts.device
        cand = torch.empty(self.X * self.Y, dtype=torch.uint8, device=dev)
        counts = torch.empty(2, dtype=torch.int32, device=dev)
        call("mopa_vgi_candidates", ptr(free), free.shape[0], free.shape[1], free.shape[2], params.ctypes.data, ptr(self.ground2d()),
             self.ox, self.oy, self.X, self.Y, ptr(cand), ptr(counts), stream())
        cells = torch.empty(self.X * self.Y, 2, dtype=torch.int32, device=dev)
        ncell = torch.empty(1, dtype=torch.int32, device=dev)
        call("mopa_vgi_compact_cells", ptr(cand), self.X, self.Y, self.ox, self.oy, ptr(cells), ptr(ncell), stream())
        return cells, ncell, counts

    def ground_cells(self, fc, pc_obj, proj_matrix, image_size):
        """Centre filters + ground lookup -> (cells (m,2) int64 numpy in voxel units, lexicographic; n_free; n_filtered)."""
        cells, ncell, counts = self.ground_cells_enqueue(fc, pc_obj, proj_matrix, image_size)
        n = int(ncell.item())                     # the one host sync of an anchor attempt
        c = counts.tolist()
        return cells[:n].cpu().numpy().astype(np.int64), c[0], c[1]

    def road_height_enqueue(self, cell, out):
        """(sum z, count) of the ground points of `cell` into out (2,) float64, enqueued only."""
        call("mopa_vgi_road_height", ptr(self.pts), self.pts.shape[1], self.pts.shape[0], float(np.float32(self.vs)), ptr(self.first), ptr(self.g_mask),
             self.ox, self.oy, self.X, self.Y, self.zlo, int(cell[0]), int(cell[1]), ptr(out), stream())

    def road_height(self, cell):
        out = torch.empty(2, dtype=torch.float64, device=self.pts.device)
        self.road_height_enqueue(cell, out)
        s, c = out.tolist()
        if c <= 0:
            raise RuntimeError("no ground point in the chosen cell")
        return np.float32(s / c)                 # the reference's mean is a float32 (ori_pc is float32)


def check_overlap(pc_scan, pc_obj, voxel_size=0.2, search_range=(25.0, 25.0), z_min=-2.0, z_max=None, front_axis="x"):
    """Same signature and return value as the reference (``mixmatch_ss.py:215-331``): (n,3) float64 numpy centres or None.
    (The training path below keeps the free-cell volume on the device instead of listing the centres.)"""
    m = OverlapMap(pc_scan, voxel_size, search_range, z_min, front_axis)
    fc = m.free_cells(pc_obj, z_max)
    if fc["free"] is None:
        return None
    idx = torch.nonzero(fc["free"]).cpu().numpy()
    if idx.shape[0] == 0:
        return None
    return (idx + (fc["extent"] - 1) / 2 + fc["offset"]) * voxel_size


def _cyl(c):
    out = np.array([np.sqrt(np.square(c[0]) + np.square(c[1])), np.arctan(c[1] / c[0])])
    if c[0] < 0 and c[1] < 0:
        out[1] -= np.pi
    if c[0] < 0 and c[1] > 0:
        out[1] += np.pi
    return out


def _place(o3, cell, road_z, jitter, voxel_size):
    """The reference's placement of one object at a ground cell (``obj_on_road``, mixmatch_ss.py:411-453): move its centre to the
    cell along its own azimuth, rotate by the azimuth difference, put its lowest point on the road (+ jitter).  -> moved (n,3)."""
    new_center = cell * voxel_size
    oc = (np.max(o3, axis=0) + np.min(o3, axis=0)) / 2
    occ, ncc = _cyl(oc), _cyl(new_center)
    d_r, d_theta = ncc - occ
    disc = np.array([d_r * np.cos(occ[1]), d_r * np.sin(occ[1]), 0])
    rot = np.array([[np.cos(d_theta), -np.sin(d_theta), 0, 0], [np.sin(d_theta), np.cos(d_theta), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    disc[2] = road_z - np.min(o3[:, 2], axis=0) + jitter
    t = np.eye(4)
    t[:3, 3] = disc
    tr = rot @ t
    h = np.concatenate((o3, np.ones((o3.shape[0], 1))), axis=1)
    return (tr @ h.T).T[:, :3]


def _assemble(m, lab, new_pc, new_lab):
    dev = m.pts.device
    n0 = m.pts.shape[0]
    obj_pc = torch.from_numpy(np.concatenate(new_pc, 0)).to(dev)
    cat_pc = torch.cat([m.pts[:, :3].double(), obj_pc], 0).contiguous()
    cat_label = torch.cat([lab, torch.from_numpy(np.concatenate(new_lab, 0)).to(dev).to(lab.dtype)], 0)
    mask = torch.zeros(cat_pc.shape[0], dtype=torch.bool, device=dev)
    mask[n0:] = True
    mask.n_obj = cat_pc.shape[0] - n0     # known on the host: post_process need not read it back from the device
    return cat_pc, cat_label, mask, mask.clone()


def point_mixmatch_batch(items, search_voxel_size=0.5, search_range=(50, 50), search_z_min=-2.0, proj_matrix=None, image_size=(),
                         front_axis="x"):
    """``[point_mixmatch(**it, insert_mode="ground", ...) for it in items]`` -- the loop of ``train_xmuda_mopa.py:516-555`` over the
    target scans of a batch -- with TWO host round trips per batch instead of four per scan: (A) the overlap test, centre filters
    and ground-cell compaction of every scan's first anchor are enqueued back to back and read together, (B) numpy's global RNG
    is drawn per scan IN SCAN ORDER exactly as the loop draws it (``choice`` then one ``rand`` per object; the draws need the cell
    counts only), the road-height lookups of all objects are enqueued and read together, (C) the placements are computed and the
    clouds assembled.  A scan whose first anchor admits no placement takes the sequential function at its turn (more round
    trips, same draws).  items: dicts with ori_pc, ori_label, obj_pc_ls, obj_label_ls, g_indices.  Same results bit for bit
    (``tests/test_gpu_vgi.py::test_batched_insertion_equals_the_per_scan_loop``).  Measured on the MoPA step of bench.py (4 + 4
    scans): 218-225 scans/s against 214-218 with the loop -- inside the run-to-run spread; that step is bound by its kernels."""
    common = dict(insert_mode="ground", search_voxel_size=search_voxel_size, search_range=search_range, search_z_min=search_z_min,
                  proj_matrix=proj_matrix, image_size=image_size, front_axis=front_axis)
    st = []
    for it in items:   # ---- (A) enqueue
        m = OverlapMap(it["ori_pc"], search_voxel_size, search_range, search_z_min, front_axis, g_mask=it["g_indices"])
        objs = it["obj_pc_ls"]
        ext = np.array([np.linalg.norm(np.max(o, axis=0)[0:2] - np.min(o, axis=0)[0:2]) for o in objs])
        anchor = np.asarray(objs[int(np.argsort(ext)[::-1][0])])
        fc = m.free_cells(anchor[:, :3])
        st.append(dict(m=m, dev=None if fc["free"] is None else m.ground_cells_enqueue(fc, anchor, proj_matrix, image_size)))
    host = [None if e["dev"] is None else (e["dev"][1].cpu(), e["dev"][2].cpu(), e["dev"][0].cpu()) for e in st]   # first .cpu() waits, the rest are ready
    heights = torch.zeros(max(1, sum(len(it["obj_pc_ls"]) for it in items)), 2, dtype=torch.float64, device=st[0]["m"].pts.device)
    out, k = [None] * len(items), 0
    for i, (it, e, h) in enumerate(zip(items, st, host)):   # ---- (B) draws in scan order, road heights enqueued
        n = 0 if h is None else int(h[0][0])
        if h is None or n == 0 or int(h[1][0]) == 0 or int(h[1][1]) == 0:
            out[i] = point_mixmatch(it["ori_pc"], it["ori_label"], it["obj_pc_ls"], it["obj_label_ls"], g_indices=it["g_indices"], **common)
            continue
        cells = h[2][:n].numpy().astype(np.int64)
        pick = np.random.choice(cells.shape[0], len(it["obj_pc_ls"]))
        e["placed"] = []
        for j in range(len(it["obj_pc_ls"])):
            cell = cells[pick[j]]
            e["m"].road_height_enqueue(cell, heights[k])
            e["placed"].append((cell, np.random.rand() * 0.1, k))
            k += 1
    hz = heights.cpu().numpy()
    for i, (it, e) in enumerate(zip(items, st)):   # ---- (C) placements
        if out[i] is not None:
            continue
        new_pc, new_lab = [], []
        for j, (cell, jitter, kk) in enumerate(e["placed"]):
            if hz[kk, 1] <= 0:
                raise RuntimeError("no ground point in the chosen cell")
            o3 = np.asarray(it["obj_pc_ls"][j])[:, :3]
            new_pc.append(_place(o3, cell, np.float32(hz[kk, 0] / hz[kk, 1]), jitter, search_voxel_size))
            new_lab.append(np.asarray(it["obj_label_ls"][j]))
        m = e["m"]
        out[i] = _assemble(m, torch.as_tensor(it["ori_label"]).to(m.pts.device), new_pc, new_lab)
    return out


def point_mixmatch(ori_pc, ori_label, obj_pc_ls, obj_label_ls, z_disc=-0.324, obj_aug=None, insert_mode="ground",
                   search_voxel_size=0.5, search_range=(50, 50), search_z_min=-2.0, proj_matrix=None, image_size=(),
                   g_indices=None, front_axis="x"):
    """``point_mixmatch`` of the reference (``mixmatch_ss.py:42-212``), both insert modes ("ground": the shipped MoPA configs; "fv").

    ``ori_pc`` (N,>=3) float32, host or device; returns ``(cat_pc, cat_label, obj_mask, obj_ps_mask)`` with ``cat_pc`` a
    DEVICE float64 (N+M,3) tensor (scan points first), the other three device tensors of length N+M -- what
    ``post_process`` below consumes.  When no object fits: the scan alone and all-False masks, like the reference."""
    if insert_mode == "fv":
        return _point_mixmatch_fv(ori_pc, ori_label, obj_pc_ls, obj_label_ls, z_disc)
    if insert_mode != "ground":
        raise ValueError(f"insert_mode must be 'ground' or 'fv', got {insert_mode!r}")
    m = OverlapMap(ori_pc, search_voxel_size, search_range, search_z_min, front_axis, g_mask=g_indices)
    dev = m.pts.device
    n0 = m.pts.shape[0]
    lab = torch.as_tensor(ori_label).to(dev)
    ext = np.array([np.linalg.norm(np.max(o, axis=0)[0:2] - np.min(o, axis=0)[0:2]) for o in obj_pc_ls])
    # ignore_idx_ls as the reference writes it (mixmatch_ss.py:123-196): a failed anchor attempt records its POSITION in the
    # extent order; Step 4 skips the object whose LIST index equals a recorded position.  The random draws still cover every object.
    ignore = []
    for idx_i, obj_idx in enumerate(np.argsort(ext)[::-1]):
        obj = np.asarray(obj_pc_ls[obj_idx])
        fc = m.free_cells(obj[:, :3])
        if fc["free"] is None:
            ignore.append(idx_i)
            continue
        cells, n_free, n_kept = m.ground_cells(fc, obj, proj_matrix, image_size)
        if n_free == 0 or n_kept == 0 or cells.shape[0] == 0:
            ignore.append(idx_i)
            continue
        pick = np.random.choice(cells.shape[0], len(obj_pc_ls))          # mixmatch_ss.py:409
        new_pc, new_lab = [], []
        for i, o in enumerate(obj_pc_ls):
            o3 = np.asarray(o)[:, :3]
            cell = cells[pick[i]]
            moved = _place(o3, cell, m.road_height(cell), np.random.rand() * 0.1, search_voxel_size)   # :444-446
            if i in ignore:
                continue
            new_pc.append(moved)
            new_lab.append(np.asarray(obj_label_ls[i]))
        return _assemble(m, lab, new_pc, new_lab)
    none = torch.zeros(n0, dtype=torch.bool, device=dev)
    none.n_obj = 0
    return m.pts[:, :3].double().contiguous(), lab, none, none.clone()


def _point_mixmatch_fv(ori_pc, ori_label, obj_pc_ls, obj_label_ls, z_disc):
    """``insert_mode="fv"`` (``mixmatch_ss.py:83-105``): lift every object by the sensors' height discrepancy and, when its mean x
    is negative, rotate it about z by twice its azimuth so that it lands in the front half space.  No search, no random draws.
    The objects are small host arrays (a few hundred points): the angle and the rotation are the reference's own numpy
    expressions in the objects' dtype (bit-identical), INCLUDING its in-place edit of the caller's arrays (:87,:99); the scan
    stays on the device and the result has ``point_mixmatch``'s device layout (float64 xyz, scan points first)."""
    pts = ori_pc if torch.is_tensor(ori_pc) else torch.from_numpy(np.ascontiguousarray(ori_pc))
    if pts.device.type != "cuda":
        pts = pts.cuda()
    dev = pts.device
    lab = torch.as_tensor(ori_label).to(dev)
    new_pc, new_lab = [], []
    for i in range(len(obj_pc_ls)):
        obj_pc = obj_pc_ls[i]
        obj_pc[:, 2] = obj_pc[:, 2] - z_disc
        ctr = np.average(obj_pc, axis=0)
        if ctr[0] < 0:
            th = np.arccos(ctr[1] / np.sqrt(ctr[0] ** 2 + ctr[1] ** 2))
            rot = np.array([[np.cos(2 * th), -np.sin(2 * th), 0], [np.sin(2 * th), np.cos(2 * th), 0], [0, 0, 1]], dtype=np.float32)
            obj_pc[:, :3] = obj_pc[:, :3].dot(rot)
        new_pc.append(obj_pc[:, :3])
        new_lab.append(np.asarray(obj_label_ls[i]))
    n0 = pts.shape[0]
    obj = torch.from_numpy(np.concatenate(new_pc, 0).astype(np.float64)).to(dev)
    cat_pc = torch.cat([pts[:, :3].double(), obj], 0).contiguous()
    cat_label = torch.cat([lab, torch.from_numpy(np.concatenate(new_lab, 0)).to(dev).to(lab.dtype)], 0)
    mask = torch.zeros(cat_pc.shape[0], dtype=torch.bool, device=dev)
    mask[n0:] = True
    mask.n_obj = cat_pc.shape[0] - n0
    return cat_pc, cat_label, mask, mask.clone()


def range_keep(cat_pc: torch.Tensor, n_scan: int, fov_up=0.05235, fov_down=-0.43633, proj_W=1024, proj_H=64) -> torch.Tensor:
    """``range_projection(..., obj_mask)['pres_idx']`` (``augmentation_3d.py:161-290``) for a cloud whose inserted-object
    points are the rows from ``n_scan`` on: bool (N,) on the device."""
    pts = cat_pc.contiguous()
    if pts.dtype != torch.float64 or pts.device.type != "cuda":
        raise RuntimeError("range_keep needs a float64 (N,3) tensor on the GPU")
    n = pts.shape[0]
    keep = torch.empty(n, dtype=torch.uint8, device=pts.device)
    ws = workspace.get(query("mopa_vgi_range_keep_workspace_bytes", n, int(proj_W), int(proj_H)), pts.device)
    call("mopa_vgi_range_keep", ptr(pts), n, int(n_scan), float(fov_up), float(fov_down), int(proj_W), int(proj_H), ptr(keep), ptr(ws),
         ws.numel(), stream())
    return keep.bool()


def _rot_matrix(noisy_rot=0.0, flip_x=0.0, flip_y=0.0, rot_z=0.0):
    """The reference's draws, in its order (``augmentation_3d.py:26-46``)."""
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


def post_process(cat_pc_ls, cat_pslabel_ls, obj_mask_ls, scale, full_scale, augment_3d, proj_W=1024, proj_H=64, fov_up=0.05235,
                 fov_down=-0.43633, scan_pth_ls=None, use_proj=True, backbone="SCN"):
    """``post_process`` of the reference (``mixmatch_ss.py:458-559``) on device tensors: occlusion culling, rotation / flip /
    scale / translation (draws from numpy's global RNG in the reference's order), int cast, field filter, collate.
    -> ``[{'x': [locs (sumN',4) int64, feats (sumN',1)]}, cat_ps_label, obj_mask, None]`` on the device."""
    if "SCN" not in backbone:
        raise IndexError("The specified backbone is not supported: {}".format(backbone))
    # Host round trips: the reference works on numpy arrays, here every `.item()` / boolean-mask index waits for the whole
    # stream.  Per scan nothing is read back (the object count rides on the mask tensor, `point_mixmatch` knows it; the NaN test
    # stays a device flag); ONE `nonzero` over the concatenated keep mask compacts the batch, the flags are read behind it.
    coords_ls, keep_ls, labels, masks, nan_flags = [], [], [], [], []
    for i, (pc, lab, om) in enumerate(zip(cat_pc_ls, cat_pslabel_ls, obj_mask_ls)):
        pc = pc.contiguous()
        nan_flags.append(torch.isnan(pc).any())
        n = pc.shape[0]
        n_obj = getattr(om, "n_obj", None)
        if n_obj is None:
            n_obj = int(om.sum().item())
        valid = range_keep(pc, n - n_obj, fov_up, fov_down, proj_W, proj_H) if (use_proj and n_obj > 0) else None
        rot = _rot_matrix(augment_3d["noisy_rot"], augment_3d.get("flip_x", 0.0), augment_3d.get("flip_y", 0.0), augment_3d["rot_z"])
        u = np.random.rand(3) if augment_3d["transl"] else None
        coords = torch.empty(n, 4, dtype=torch.int64, device=pc.device)
        keep = torch.empty(n, dtype=torch.uint8, device=pc.device)
        ws = workspace.get(query("mopa_voxelize_f64_workspace_bytes"), pc.device)
        rot_c = None if rot is None else np.ascontiguousarray(rot.astype(np.float64))
        u_c = None if u is None else np.ascontiguousarray(u, np.float64)
        vk = None if valid is None else valid.to(torch.uint8)
        call("mopa_voxelize_f64", ptr(pc), n, ptr(vk), None if rot_c is None else rot_c.ctypes.data, float(scale), int(full_scale),
             None if u_c is None else u_c.ctypes.data, int(u is not None), i, ptr(coords), ptr(keep), ptr(ws), ws.numel(), stream())
        coords_ls.append(coords)
        keep_ls.append(keep)
        labels.append(torch.as_tensor(lab).to(pc.device))
        masks.append(om)
    sel = torch.nonzero(torch.cat(keep_ls, 0)).squeeze(1)       # the one synchronisation of the batch
    bad = torch.stack(nan_flags).cpu()
    if bool(bad.any()):
        i = int(torch.nonzero(bad)[0])
        raise AssertionError("Found Nan object points: {}".format(scan_pth_ls[i] if scan_pth_ls else i))
    locs = torch.cat(coords_ls, 0).index_select(0, sel)
    return [{"x": [locs, torch.ones(locs.shape[0], 1, device=locs.device)]}, torch.cat(labels, 0).index_select(0, sel),
            torch.cat(masks, 0).index_select(0, sel), None]
