#!/bin/bash
# per-kernel time of a script under rocprofv3:  scripts/prof_kernels.sh <script.py> [rows]     (run on the GPU box)
S=$1; ROWS=${2:-40}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
( cd /tmp && export TMPDIR=/tmp && env $PROF_ENV rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o ks -- python $GRAFT_REPO_ROOT/$S > /tmp/pk.out 2> /tmp/pk.err )
python - <<PY
import csv, glob
fs = glob.glob("/tmp/pk/**/*kernel_stats.csv", recursive=True)
if not fs:
    print(open("/tmp/pk.err").read()[-2000:]); raise SystemExit(1)
for r in list(csv.DictReader(open(fs[0])))[:$ROWS]:
    print("%-120s %6s %12s %10s %6s" % (r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
