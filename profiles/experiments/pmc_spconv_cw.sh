#!/bin/bash
# Round 3: SQ + TCP counters of the column-wave experiment (k_spconv_cw) next to k_spconv_t4, per layer shape.
# Usage (GPU box, repo root): bash profiles/pmc_spconv_cw.sh [levels=4]
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=${1:-4}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
P2="GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/pmc_cw_$i -o run -- python3 profiles/bench_spconv_cs.py $L 2 > gpurun_out/pmc_cw_$i.log 2>&1
  python profiles/pmc_summary.py gpurun_out/pmc_cw_$i/run_counter_collection.csv spconv > gpurun_out/pmc_cw_$i.txt
done
