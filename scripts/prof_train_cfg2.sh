O=gpurun_out/r05; mkdir -p $O; R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && CONFIG=2 TRAIN_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/t2prof -o ks -- python $R/scripts/bench_train.py > $R/$O/t2.json 2>/dev/null )
cat $O/t2.json
python - <<PY
import csv,glob
f=glob.glob("$O/t2prof/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("kernel ms per step", tot/13e6, "launches per step", calls/13)
for r in rows[:14]:
    print("%-100s %5s %8.1f us/step" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/13e3))
PY
rm -rf $O/t2prof
