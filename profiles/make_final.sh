#!/bin/bash
# Regenerates the round's judged artifacts on the GPU box (run from the repo root through gpurun):
#   bench lines (3d / joint / mopa), rocprofv3 --kernel-trace --stats summaries, PMC HBM-traffic passes.
# Everything lands in gpurun_out/final/; copy the summaries into profiles/ afterwards (profiles/collect_final.py).
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_joint.json 2> $O/bench_joint.err   # the driver's command
# (after the joint run: the host-bound 3D-only step is the one that suffers from a cold box -- first process, page cache)
timeout 600 python3 bench.py --workload 3d --steps 50 --warmup 5 > $O/bench_3d.json 2> $O/bench_3d.err
timeout 600 python3 bench.py --workload mopa --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_mopa.json 2> $O/bench_mopa.err
timeout 600 python3 bench.py --workload kitti --steps 10 --warmup 3 > $O/bench_kitti.json 2> $O/bench_kitti.err   # BASELINE configs[4] per GPU
# The trace / counter passes run the SAME timed command; the two extra measurements bench.py makes after its timed region (the
# reference's two-calls-per-domain loop order, the merged 3D pass alone with every sparse-conv launch bracketed) are switched off for
# them: they launch the sparse-conv family on other batch sizes and would mix into the per-family averages.
export MOPA_BENCH_TWO_CALLS=0 MOPA_BENCH_SPARSE_ALONE=0
for W in 3d joint kitti mopa; do
  A=""; [ $W = 3d ] && A="--workload 3d"; [ $W = kitti ] && A="--workload kitti"; [ $W = mopa ] && A="--workload mopa"
  # kernel statistics of the SAME command the bench line comes from (joint: --steps 20 --warmup 5 = the driver's; 25 steps traced)
  S="--steps 20 --warmup 5"; [ $W = 3d ] && S="--steps 50 --warmup 5"; [ $W = kitti ] && S="--steps 10 --warmup 3"; [ $W = mopa ] && S="--steps 10 --warmup 3"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$W -o run -- python3 bench.py $A $S --no-cpu-baseline > $O/trace_$W.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$W -o run -- python3 bench.py $A --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_$W.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$W -o run -- python3 bench.py $A --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_$W.log 2>&1
  N=37; [ $W = 3d ] && N=57; [ $W = kitti ] && N=25; [ $W = mopa ] && N=15   # joint: 2 setup + 5 warm-up + 20 timed + 2 + 8 steps of the host-inputs measurement (kitti: 2 + 3 + 10 + 2 + 8)
  python3 profiles/summarize.py $O/trace_$W/run_kernel_stats.csv $N > $O/stats_$W.md
  NETS=2; [ $W = 3d ] && NETS=1
  python3 profiles/traffic.py $O/pmc_fetch_$W/run_counter_collection.csv $O/pmc_write_$W/run_counter_collection.csv $NETS > $O/traffic_$W.json
  cp $O/trace_$W/run_kernel_stats.csv $O/stats_$W.csv
done
tail -c 600 $O/bench_3d.json; echo; tail -c 300 $O/bench_joint.json; echo; head -12 $O/stats_3d.md
