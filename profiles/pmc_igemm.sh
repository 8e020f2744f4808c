#!/bin/bash
# SQ / TCP / TCC counter passes over profiles/bench_igemm.py (one rocprofv3 --pmc run per counter group).
# Usage (on the GPU box, from the repo root): bash profiles/pmc_igemm.sh [kernel-name filter=igemm_mfma]
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
F=${1:-igemm_mfma}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
P4="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
i=0
for P in "$P1" "$P4" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_ig_$i -o run -- python3 profiles/bench_igemm.py > gpurun_out/pmc_ig_$i.log 2>&1
  python3 profiles/pmc_summary.py gpurun_out/pmc_ig_$i/run_counter_collection.csv $F | head -60
done
