#!/usr/bin/env python3
"""Aggregate a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv by kernel family
(template instantiations merged) into a small markdown table.  Usage: summarize.py <kernel_stats.csv> [steps]"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        base = re.sub(r"\(.*", "", re.sub(r"<.*", "", r["Name"]).replace("void ", ""))
        agg[base][0] += int(r["Calls"])
        agg[base][1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print("| kernel family | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"| `{k[:70]}` | {v[0]} | {v[1] / 1e6:.3f} | {v[1] / v[0] / 1e3:.2f} | {100 * v[1] / tot:.1f} |")
    print(f"\ntotal kernel time {tot / 1e6:.3f} ms" + (f" = {tot / 1e6 / steps:.3f} ms/step over {steps} steps" if steps else ""))


if __name__ == "__main__":
    main()
