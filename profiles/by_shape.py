#!/usr/bin/env python3
"""Per-(kernel instantiation, grid) durations from a rocprofv3 kernel_trace.csv.  Usage: by_shape.py trace.csv [filter]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
    if flt not in name:
        continue
    key = (name[:34], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
           "v" + r["VGPR_Count"], "lds" + r["LDS_Block_Size"])
    agg[key][0] += 1
    agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(k, "calls", v[0], "avg us %.1f" % (v[1] / v[0] / 1e3), "total ms %.1f" % (v[1] / 1e6))
