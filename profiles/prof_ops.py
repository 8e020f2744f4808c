#!/usr/bin/env python3
"""torch.profiler view of one bench workload: aten ops by call count / device time (finds stray elementwise launches).
Usage: python profiles/prof_ops.py [3d|joint]"""
import runpy
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
wl = sys.argv[1] if len(sys.argv) > 1 else "3d"
sys.argv = ["bench.py", "--workload", wl, "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") or "Function" in e.key or "Backward" in e.key]
rows.sort(key=lambda e: -e.count)
print(f"{'op':50s} {'calls':>7s} {'device ms':>10s}")
for e in rows[:40]:
    print(f"{e.key[:50]:50s} {e.count:7d} {e.device_time_total / 1e3:10.3f}")
