#!/bin/bash
# rocprofv3 kernel stats of scripts/bench_tn.py (the weight-gradient product, direct and packed forms)
O=gpurun_out/r05; mkdir -p $O; R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tnprof -o ks -- python $R/scripts/bench_tn.py > $R/$O/tn_under_rocprof.jsonl 2>/dev/null )
cp $(find $O/tnprof -name "*kernel_stats.csv" | head -1) $O/tn_kernel_stats.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/tn_kernel_stats.csv")))[:14]:
    print("%-100s %5s avg %8.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3))
PY
