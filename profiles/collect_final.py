#!/usr/bin/env python3
"""Copy the artifacts of profiles/make_final.sh (gpurun_out/final/) into the tracked profiles/ files of this round, stamped with the
commit they were taken at, and write profiles/<round>_rocprof_family.json: the rocprofv3 average launch duration of the kernel
families bench.py's `roofline` objects are about (bench.py prints `frac_rocprof` from it beside its own HIP-event figure).
Usage (repo root, after `gpurun -- bash profiles/make_final.sh`): python profiles/collect_final.py [round=r3] [lines]
(`lines`: after refresh_lines.sh -- re-embed the bench lines, keep the commit stamp of the traces)"""
import collections, csv, json, os, re, shutil, subprocess, sys
R = os.path.dirname(os.path.abspath(__file__))
F = os.path.join(R, "..", "gpurun_out", "final")
RND = sys.argv[1] if len(sys.argv) > 1 else "r6"
TITLE = {"3d": "Net3DSeg-only training step (bs 8, 1x MI355X) -- BASELINE configs[1]",
         "joint": "joint 2D+3D xMUDA step (bs 8+8, 1x MI355X) -- BASELINE configs[2]",
         "kitti": "A2D2->SemanticKITTI-shape joint step (bs 2+2, 120,000-pt scans, 10 classes, 1x MI355X) -- BASELINE configs[4] per GPU",
         "mopa": "MoPA iteration (bs 4+4, VGI + third 3D pass + pseudo-label CE + SAM-mask loss, 1x MI355X) -- BASELINE configs[3] per GPU"}
CMD = {"3d": "python bench.py --workload 3d --steps 50 --warmup 5", "joint": "python bench.py --steps 20 --warmup 5",
       "kitti": "python bench.py --workload kitti --steps 10 --warmup 3",
       "mopa": "python bench.py --workload mopa --steps 10 --warmup 3 --no-cpu-baseline"}
STEPS = {"3d": 57, "joint": 37, "kitti": 25, "mopa": 15}   # steps traced by make_final.sh (setup + warm-up + timed + host-input steps)
FAMILY = {"sparse_conv": ("k_spconv_t4", "k_spconv_pipe", "k_spconv_fwd", "k_spconv_blk", "k_spconv_run", "k_spconv_stem"),
          "dense_mfma": ("k_conv2d_igemm_mfma", "k_wino4_gemm_out", "k_wino4_conv", "k_wino4_conv32", "k_wino4_conv9"),
          # one launch of these = one weight gradient of the 2D network (bench.py: roofline_wgrad)
          "wgrad_mfma": ("k_conv2d_wgrad_mfma", "k_wino4_wgrad", "k_stem_wgrad_mfma", "k_wgemm_tn")}
# kernels whose TIME belongs to a family's launches without being launches of their own: the ordered per-row sum behind every
# offset-major convolution with a slab (csrc/sprun.hip: one k_spconv_run + one k_run_reduce = one convolution)
FAMILY_EXTRA = {"sparse_conv": ("k_run_reduce",), "wgrad_mfma": ("k_reduce_slabs2", "k_wino4_dw", "k_wino_dw")}


def git(*a):
    return subprocess.run(("git",) + a, cwd=os.path.join(R, ".."), capture_output=True, text=True).stdout.strip()


commit = git("rev-parse", "--short", "HEAD") + ("+dirty" if git("status", "--porcelain", "--", "mopa_amd", "bench.py") else "")
if len(sys.argv) > 2 and sys.argv[2] == "lines":   # only the bench lines were re-taken (refresh_lines.sh): the traces keep their stamp
    prev = os.path.join(R, f"{RND}_rocprof_family.json")
    if os.path.exists(prev):
        commit = json.load(open(prev)).get("commit", commit)
fam = {"commit": commit, "note": "rocprofv3 --kernel-trace --stats of the bench command (profiles/make_final.sh); avg_us = total duration / calls over the family"}
for w in ("3d", "joint", "kitti", "mopa"):
    if not os.path.exists(os.path.join(F, f"bench_{w}.json")):
        continue
    line = open(os.path.join(F, f"bench_{w}.json")).read().strip().splitlines()[-1]
    json.loads(line)
    stats = open(os.path.join(F, f"stats_{w}.md")).read()
    with open(os.path.join(R, f"{RND}_final_{w}_kernel_stats.md"), "w") as f:
        f.write(f"# Round {RND[1:]} final: {TITLE[w]}\n\nTaken at commit `{commit}`.  Commands (profiles/make_final.sh): `{CMD[w]}` (bench line) and "
                f"`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ... <same --steps / --warmup> --no-cpu-baseline` "
                f"(every step traced, warm-up included"
                + ("; the 3D branch runs on a second stream, so kernel times overlap and their sum exceeds wall time" if w != "3d" else "")
                + f").\n\n```\n{line}\n```\n\n{stats}")
    shutil.copy(os.path.join(F, f"stats_{w}.csv"), os.path.join(R, f"{RND}_final_{w}_kernel_stats.csv"))
    t = json.load(open(os.path.join(F, f"traffic_{w}.json")))
    t["_commit"] = commit
    json.dump(t, open(os.path.join(R, f"{RND}_{w}_hbm_traffic.json"), "w"), indent=1)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(F, f"stats_{w}.csv"))):
        base = re.sub(r"\(.*", "", re.sub(r"<.*", "", r["Name"]).replace("void ", ""))
        agg[base][0] += int(r["Calls"])
        agg[base][1] += float(r["TotalDurationNs"])
    fam[w] = {}
    for name, kernels in FAMILY.items():
        calls = sum(agg[k][0] for k in kernels if k in agg)
        ns = sum(agg[k][1] for k in kernels if k in agg) + sum(agg[k][1] for k in FAMILY_EXTRA.get(name, ()) if k in agg)
        if calls:
            fam[w][name] = {"calls": calls, "avg_us": round(ns / calls / 1e3, 3), "calls_per_step": round(calls / STEPS[w], 2)}
json.dump(fam, open(os.path.join(R, f"{RND}_rocprof_family.json"), "w"), indent=1)
print(json.dumps(fam))
if os.path.exists(os.path.join(F, "bench_mopa.json")):
    print(open(os.path.join(F, "bench_mopa.json")).read().strip().splitlines()[-1][:200])
