#!/bin/bash
# PMC counters of the fused hop kernel (run on the GPU box): scripts/pmc_fused.sh <outdir> [DBG mask list]
set -e
OUT=${1:-gpurun_out/pmc_fused}; DBG=${2:-0}
mkdir -p $OUT; export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  DBG=$DBG rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python scripts/bench_fused_debug.py > $OUT/$tag.log 2>&1 || true
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_linear_split3" in n or "k_gat_mp_tiled" in n:
            acc[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
