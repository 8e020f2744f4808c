#!/bin/bash
# Re-runs only the four bench lines of profiles/make_final.sh (same commands), after profiles/collect_final.py has written this
# code's rocprofv3 family durations and PMC traffic into profiles/: the lines then quote figures taken at the same commit.
# Then `python profiles/collect_final.py r3` again re-embeds them in profiles/r3_final_*_kernel_stats.md.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_joint.json 2> $O/bench_joint.err
timeout 600 python3 bench.py --workload 3d --steps 50 --warmup 5 > $O/bench_3d.json 2> $O/bench_3d.err
timeout 600 python3 bench.py --workload mopa --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_mopa.json 2> $O/bench_mopa.err
timeout 600 python3 bench.py --workload kitti --steps 10 --warmup 3 > $O/bench_kitti.json 2> $O/bench_kitti.err
tail -c 400 $O/bench_joint.json
