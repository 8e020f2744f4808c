#!/bin/bash
# SQ / TCP / TCC counter passes over profiles/bench_spconv.py (one rocprofv3 --pmc run per counter group).
# Usage (on the GPU box, from the repo root): bash profiles/pmc_spconv.sh [levels=4] [kernel-name filter=spconv_t4]
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=${1:-4}; F=${2:-spconv_t4}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
P2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
P4="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_sp_$i -o run -- python3 profiles/bench_spconv.py $L 5 > gpurun_out/pmc_sp_$i.log 2>&1
  python profiles/pmc_summary.py gpurun_out/pmc_sp_$i/run_counter_collection.csv $F
done
