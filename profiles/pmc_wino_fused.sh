#!/bin/bash
# SQ counters of k_wino4_gemm_out (64 -> 64 at 152x240): separate --pmc passes, kernel-trace only.
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_w4g; mkdir -p $O
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/p$i -o run -- python3 profiles/one_wino_fused.py > $O/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmc_w4g/p*/run_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_wino4_gemm_out" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg): print(f"{k:32s} {agg[k][1] / agg[k][0]:16.0f} per launch ({agg[k][0]} launches)")
PY
