#!/bin/bash
# rocprofv3 kernel stats of one stage of scripts/bench_pipeline.py (STAGE=encoder|gat_seq|pooling|classifier), 23 calls
O=gpurun_out/r05; mkdir -p $O; R=$GRAFT_REPO_ROOT; S=${1:-encoder}
( cd /tmp && export TMPDIR=/tmp && STAGE=$S rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/sprof_$S -o ks -- python $R/scripts/bench_pipeline.py > $R/$O/stage_$S.json 2>/dev/null )
cp $(find $O/sprof_$S -name "*kernel_stats.csv" | head -1) $O/stage_${S}_kernel_stats.csv; cat $O/stage_$S.json
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stage_${S}_kernel_stats.csv")))
for r in rows[:22]:
    print("%-105s %5s %8.1f us/call" % (r["Name"][:105], r["Calls"], float(r["TotalDurationNs"])/23e3))
PY
rm -rf $O/sprof_$S
