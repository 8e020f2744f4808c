cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
P4="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
i=0
for P in "$P1" "$P4"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_wg_$i -o run -- python3 bench.py --workload 3d --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_wg_$i.log 2>&1
  python profiles/pmc_summary.py gpurun_out/pmc_wg_$i/run_counter_collection.csv "wgrad2<4, 4"
done
