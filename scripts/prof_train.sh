#!/bin/bash
# Training step of gat_seq (scripts/bench_train.py): wall time + rocprofv3 kernel stats of the step alone
O=gpurun_out/r05; mkdir -p $O; R=$GRAFT_REPO_ROOT
python scripts/bench_train.py 2>/dev/null | tail -1 > $O/train_cfg3.json; cat $O/train_cfg3.json
CONFIG=2 TRAIN_ONLY=1 python scripts/bench_train.py 2>/dev/null | tail -1 > $O/train_cfg2.json; cat $O/train_cfg2.json
( cd /tmp && export TMPDIR=/tmp && TRAIN_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tprof -o ks -- python $R/scripts/bench_train.py > /dev/null 2>&1 )
cp $(find $O/tprof -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step", tot/13e6)
for r in rows[:40]:
    print("%-96s %5s %8.1f us/step" % (r["Name"][:96], r["Calls"], float(r["TotalDurationNs"])/13e3))
PY
