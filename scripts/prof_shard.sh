O=gpurun_out/r05; mkdir -p $O; R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/shprof -o ks -- python $R/bench.py --emulate-world 8 --no-cpu-baseline --no-pmc --no-extras > $R/$O/shard8_under_rocprof.json 2>/dev/null )
python - <<PY
import csv,glob,json
d=json.loads(open("$O/shard8_under_rocprof.json").read().strip().splitlines()[-1]); print(d.get("ms_per_step"), d.get("steps"), d.get("warmup"))
f=glob.glob("$O/shprof/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print("%-100s %5s avg %7.1f us total %8.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
rm -rf $O/shprof
