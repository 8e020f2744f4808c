#!/bin/bash
# Round 3: texture-path counters (TA / TCP = the vector L1 and its address unit) of the sparse-conv kernels, per layer shape:
# is the L1 path the co-bottleneck of k_spconv_t4 that its operand byte count per MFMA suggests?
# Usage (GPU box, repo root): bash profiles/pmc_spconv3.sh [levels=5] [name filter=spconv]
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=${1:-5}; F=${2:-spconv}
P1="GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
P2="GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
P3="GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_GATE_EN1_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/pmc3_sp_$i -o run -- python3 profiles/bench_spconv.py $L 3 > gpurun_out/pmc3_sp_$i.log 2>&1
  python profiles/pmc_summary.py gpurun_out/pmc3_sp_$i/run_counter_collection.csv $F > gpurun_out/pmc3_sp_$i.txt
done
