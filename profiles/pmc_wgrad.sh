#!/bin/bash
# Round 5: SQ counters of the sparse weight-gradient kernels (k_spconv_wgrad2 and k_wgrad_run), per template instantiation and grid.
# Usage (GPU box, repo root): bash profiles/pmc_wgrad.sh [levels=5] [name filter=wgrad]
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=${1:-5}; F=${2:-wgrad}
export MOPA_SPCONV_WGRAD_RUN=2
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/pmc_wg_$i -o run -- python3 profiles/bench_wgrad.py $L 3 > gpurun_out/pmc_wg_$i.log 2>&1
  python profiles/pmc_summary.py gpurun_out/pmc_wg_$i/run_counter_collection.csv $F > gpurun_out/pmc_wg_$i.txt
done
