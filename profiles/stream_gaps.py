#!/usr/bin/env python3
"""Per-queue timeline of a rocprofv3 --kernel-trace CSV: busy time, idle gaps and the kernels in front of the largest gaps, for the
steady-state steps of a bench run.  Usage: python profiles/stream_gaps.py <run_kernel_trace.csv> [skip_first_fraction=0.5]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
key_q = "Queue_Id" if "Queue_Id" in rows[0] else "Queue_ID"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t0 + (t1 - t0) * skip
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
print(f"analysed span {span:.1f} ms, {len(rows)} kernels")
byq = collections.defaultdict(list)
for r in rows:
    byq[r[key_q]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
    gaps = collections.Counter()
    gapt = collections.Counter()
    small = 0
    tot_gap = 0.0
    for a, b in zip(rs, rs[1:]):
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        if g > 0:
            tot_gap += g
            name = re.sub(r"<.*", "", a["Kernel_Name"]).replace("void ", "")[:40]
            nxt = re.sub(r"<.*", "", b["Kernel_Name"]).replace("void ", "")[:40]
            gaps[(name, nxt)] += 1
            gapt[(name, nxt)] += g
            if g < 5:
                small += 1
    print(f"\nqueue {q}: {len(rs)} kernels, busy {busy:.1f} ms, gaps {tot_gap / 1e3:.1f} ms ({small} gaps < 5 us)")
    for (a, b), t in gapt.most_common(14):
        print(f"   {t / 1e3:7.2f} ms in {gaps[(a, b)]:5d} gaps (avg {t / gaps[(a, b)]:6.1f} us)  {a} -> {b}")
