#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel instantiation.  Usage: pmc_summary.py counter_collection.csv [name filter]"""
import collections, csv, re, sys
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))[:40]
    if flt not in name:
        continue
    key = (name, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(key, r["Counter_Name"])] += 1
for key, cs in agg.items():
    print(key)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {v / n[(key, c)]:16.0f} per launch ({n[(key, c)]} launches)")
