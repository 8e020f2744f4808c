#!/usr/bin/env python3
"""Prints the fields of a bench.py JSON line that DESIGN.md quotes (usage: python profiles/show_line.py FILE...)."""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(path)
    print("  ", {k: d.get(k) for k in ("value", "ms_per_step", "value_with_host_inputs", "value_two_calls_per_domain", "peak_device_memory_GB")})
    sp = d.get("roofline_sparse_conv") or {}
    print("   sparse:", {k: sp.get(k) for k in ("frac", "achieved", "avg_launch_us", "frac_rocprof", "avg_launch_us_rocprof",
                                                 "frac_event_bracket_with_queue_wait", "launches_per_step", "algorithmic_bytes_per_launch",
                                                 "traffic", "mixed_roofline", "rocprof_note")})
    r = d.get("roofline") or {}
    print("   roofline:", {k: r.get(k) for k in ("bound", "frac", "frac_rocprof", "achieved", "avg_launch_us", "launches_per_step", "traffic",
                                                   "algorithmic_bytes_per_launch")})
    print("   cpu:", (d.get("cpu_baseline") or {}).get("value"), "| allreduces/step", d["config"].get("allreduces_per_step"))
