#!/bin/bash
# Round 6, first GPU call: this box's baseline + the questions that steer the round (config-2 aggregate-first ablation, shard under
# graph replay) + the evidence VERDICT r05 found missing (kernel traces of the stand-alone MP kernel, configs 2/4/5, the 8-way shard).
O=gpurun_out/r06; mkdir -p $O; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
python bench.py --no-cpu-baseline --no-extras > $O/first_bench.json 2> $O/first_bench.err
for f in 3 4 5; do CONFIG=2 FUSION=$f python scripts/bench_hopagg.py 2>/dev/null | tail -1; done > $O/cfg2_hop_forms.jsonl
( export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so; for d in 0 1 2 4 5 16 32 33 37 53; do CONFIG=2 FUSION=4 GVQA_HOPAGG_DEBUG=$d python scripts/bench_hopagg.py 2>/dev/null | tail -1; done ) > $O/cfg2_hopagg_loop_parts.jsonl
python scripts/bench_hipgraph.py > $O/hipgraph.jsonl 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/mpprof -o ks -- python $R/scripts/bench_mp_only.py > $R/$O/mp_only_under_rocprof.txt 2>/dev/null )
cp $(find $O/mpprof -name "*kernel_stats.csv" | head -1) $O/mp_standalone_kernel_stats.csv 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/cfprof -o ks -- python $R/scripts/bench_configs.py > $R/$O/configs_under_rocprof.json 2>/dev/null )
cp $(find $O/cfprof -name "*kernel_stats.csv" | head -1) $O/configs245_kernel_stats.csv 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/shprof -o ks -- python $R/bench.py --emulate-world 8 --no-cpu-baseline --no-pmc --no-extras > $R/$O/shard8_under_rocprof.json 2>/dev/null )
cp $(find $O/shprof -name "*kernel_stats.csv" | head -1) $O/shard8_kernel_stats.csv 2>/dev/null
rm -rf $O/mpprof $O/cfprof $O/shprof
ls -la $O
