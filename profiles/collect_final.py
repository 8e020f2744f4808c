#!/usr/bin/env python3
"""Copy the artifacts of profiles/make_final.sh (gpurun_out/final/) into the tracked profiles/ files of this round."""
import json, os, shutil
R = os.path.dirname(os.path.abspath(__file__))
F = os.path.join(R, "..", "gpurun_out", "final")
TITLE = {"3d": "Net3DSeg-only training step (bs 8, 1x MI355X) -- BASELINE configs[1]",
         "joint": "joint 2D+3D xMUDA step (bs 8+8, 1x MI355X) -- BASELINE configs[2]"}
CMD = {"3d": "python bench.py --workload 3d --steps 50 --warmup 5", "joint": "python bench.py --steps 20 --warmup 5"}
for w in ("3d", "joint"):
    line = open(os.path.join(F, f"bench_{w}.json")).read().strip().splitlines()[-1]
    json.loads(line)
    stats = open(os.path.join(F, f"stats_{w}.md")).read()
    with open(os.path.join(R, f"r2_final_{w}_kernel_stats.md"), "w") as f:
        f.write(f"# Round 2 final: {TITLE[w]}\n\nCommands (profiles/make_final.sh): `{CMD[w]}` (bench line) and `rocprofv3 --kernel-trace "
                f"--stats --output-format csv -- python3 bench.py ... <same --steps / --warmup> --no-cpu-baseline` (every step traced, warm-up included"
                + ("; the 3D branch runs on a second stream, so kernel times overlap and their sum exceeds wall time" if w == "joint" else "")
                + f").\n\n```\n{line}\n```\n\n{stats}")
    shutil.copy(os.path.join(F, f"stats_{w}.csv"), os.path.join(R, f"r2_final_{w}_kernel_stats.csv"))
    shutil.copy(os.path.join(F, f"traffic_{w}.json"), os.path.join(R, f"r2_{w}_hbm_traffic.json"))
print(open(os.path.join(F, "bench_mopa.json")).read().strip().splitlines()[-1][:200])
