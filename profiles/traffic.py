#!/usr/bin/env python3
"""HBM traffic per launch of each kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB (x1024); on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, i.e. reads exactly 1/2 of a wide coalesced stream -> doubled here
(an upper estimate for narrow/gather access).  Usage: traffic.py fetch_counter_collection.csv write_counter_collection.csv [networks per step = 2]
"""
import collections, csv, json, re, sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        base = re.sub(r"\(.*", "", re.sub(r"<.*", "", r["Kernel_Name"]).replace("void ", ""))
        agg[base][0] += 1
        agg[base][1] += float(r["Counter_Value"])
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] * 2 + w.get(k, [0, 0])[1])):
        nf, vf = f.get(k, [0, 0.0])
        nw, vw = w.get(k, [0, 0.0])
        n = max(nf, nw, 1)
        out[k] = {"launches": n, "read_MB_per_launch": round(2 * vf * 1024 / n / 1e6, 3),
                  "write_MB_per_launch": round(vw * 1024 / n / 1e6, 3),
                  "hbm_bytes_per_launch": int((2 * vf + vw) * 1024 / n)}
    # bytes per STEP: the optimizer kernel runs once per network and step (argv[3] = networks per step: 1 for `--workload 3d`, else 2)
    nets = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    adam = out.get("k_adam_flat", {}).get("launches", 0)
    if adam:
        steps = adam / nets
        total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
        out["_steps"] = steps
        out["_hbm_GB_per_step"] = round(total / steps / 1e9, 2)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
