#!/usr/bin/env python3
"""Markdown table of the texture-path counters collected by profiles/pmc_spconv3.sh (per kernel instantiation and grid).
Usage: python profiles/pmc_tcp_summary.py gpurun_out/pmc3_sp_2.txt gpurun_out/pmc3_sp_3.txt [name filter]"""
import re
import sys

NTCP = 256   # one vector L1 (TCP) per CU
rows = {}
for path in sys.argv[1:3]:
    for blk in re.split(r"\n(?=\()", open(path).read()):
        lines = blk.strip().split("\n")
        if not lines or not lines[0].startswith("("):
            continue
        m = re.match(r"\('(.*)', (\d+)\)", lines[0])
        key = (m.group(1), int(m.group(2)))
        d = rows.setdefault(key, {})
        for l in lines[1:]:
            p = l.split()
            d[p[0]] = float(p[1])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
print("| kernel <template>, blocks | duration cycles | TCP busy | cache-line accesses / TCP / cycle | L2 read requests | mean L2 read latency (cycles) | TCP stalled on a pending miss |")
print("|---|---|---|---|---|---|---|")
for (name, blocks), d in rows.items():
    if flt not in name or "GRBM_GUI_ACTIVE" not in d:
        continue
    dur = d["GRBM_GUI_ACTIVE"] / 8.0          # rocprofv3 sums the 8 XCDs
    busy = d.get("TCP_GATE_EN1_sum", 0) / NTCP / dur
    acc = d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / NTCP / dur
    req = d.get("TCP_TCC_READ_REQ_sum", 0)
    lat = d.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / req if req else 0
    pend = d.get("TCP_PENDING_STALL_CYCLES_sum", 0) / NTCP / dur
    print(f"| `{name}`, {blocks} | {dur:.0f} | {100 * busy:.0f} % | {acc:.2f} | {req / 1e6:.2f} M | {lat:.0f} | {100 * pend:.0f} % |")
