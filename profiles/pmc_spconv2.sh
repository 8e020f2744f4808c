#!/bin/bash
# Round 2: SQ counters (MFMA busy, waits, instruction mix) + effective clock of the sparse-conv kernels, per layer shape.
# Usage (GPU box, repo root): bash profiles/pmc_spconv2.sh [levels=5] [name filter=spconv]
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=${1:-5}; F=${2:-spconv}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/pmc2_sp_$i -o run -- python3 profiles/bench_spconv.py $L 3 > gpurun_out/pmc2_sp_$i.log 2>&1
  python profiles/pmc_summary.py gpurun_out/pmc2_sp_$i/run_counter_collection.csv $F > gpurun_out/pmc2_sp_$i.txt
done
