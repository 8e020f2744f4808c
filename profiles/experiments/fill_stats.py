#!/usr/bin/env python3
"""How full are the MFMA row groups of the sparse convolutions under different tilings (CPU, numpy; synthetic nuScenes-shape scans)?
Per UNet level: rules per 16-row MFMA group for the shipped grouping (16-rule groups per 64-row tile and filter offset), for 128-row
tiles, for un-compacted 16- / 32-row slices with register accumulators (the output-stationary alternative: 2x the MFMAs), and the
same after sorting the rows along a Morton curve (barely moves: lidar rows are already spatially coherent).  Round 4, DESIGN section 3.
Usage: python profiles/experiments/fill_stats.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle.scn3d import Geometry, unpack_keys
from mopa_amd import synth
b = synth.make_batch(2, H=16, W=16)
coords = b["x"][0].numpy() if hasattr(b["x"][0], "numpy") else np.asarray(b["x"][0])
g = Geometry(coords, 7, 4096)
def stats(nbr, rows_per, perm=None):
    K, A = nbr.shape
    v = (nbr >= 0)
    if perm is not None: v = v[:, perm]
    pad = (-A) % rows_per
    v = np.pad(v, ((0,0),(0,pad)))
    n = v.reshape(K, -1, rows_per).sum(2)        # rules per (offset, slice)
    rules = n.sum()
    g16 = ((n + 15)//16).sum()                   # 16-rule compacted groups
    act = (n > 0).sum()                          # active (offset, slice)
    return rules, g16, act
def morton(xyzb):
    def part(v):
        v = v.astype(np.uint64) & 0xfff
        r = np.zeros_like(v)
        for i in range(12): r |= ((v >> np.uint64(i)) & np.uint64(1)) << np.uint64(3*i)
        return r
    return (xyzb[:,3].astype(np.uint64) << np.uint64(40)) | part(xyzb[:,0]) << np.uint64(2) | part(xyzb[:,1]) << np.uint64(1) | part(xyzb[:,2])
for l in range(7):
    nbr = g.nbr27[l]
    A = nbr.shape[1]
    r, g64, _ = stats(nbr, 64)
    _, g128, _ = stats(nbr, 128)
    _, _, a16 = stats(nbr, 16)
    _, _, a32 = stats(nbr, 32)
    xyzb = unpack_keys(g.row_keys[l])
    pm = np.argsort(morton(xyzb), kind="stable")
    _, g64m, _ = stats(nbr, 64, pm)
    _, _, a16m = stats(nbr, 16, pm)
    _, _, a32m = stats(nbr, 32, pm)
    print(f"L{l} rows {A} rules {r} nb/row {r/A:.2f} | MFMA row-groups(16): grouped64 {g64} (fill {r/g64/16:.2f}) grouped128 {g128} ({r/g128/16:.2f}) slice16 {a16} ({r/a16/16:.2f}) slice32 {a32*2} ({r/a32/32:.2f}) | morton: grouped64 {g64m} ({r/g64m/16:.2f}) slice16 {a16m} ({r/a16m/16:.2f}) slice32 {a32m*2} ({r/a32m/32:.2f})")
    if l < 6:
        for name, tab in (("down", g.ch[l]), ("up", g.up[l])):
            r, g64, _ = stats(tab, 64); _, _, a16 = stats(tab, 16)
            print(f"   {name} rows {tab.shape[1]} rules {r} grouped64 {g64} ({r/g64/16:.2f}) slice16 {a16} ({r/a16/16:.2f})")
