#!/bin/bash
# Evidence run on the GPU box (everything lands under gpurun_out/r/): scripts/collect_profiles.sh
# bench line (live PMC traffic, comparison legs, CPU baseline), rocprofv3 kernel stats of the default path, BASELINE configs,
# emulated strong-scaling shards, the distributed step with one rank, stand-alone split GEMM variants, SQ counters of the fused hop.
O=gpurun_out/r; mkdir -p $O; export TMPDIR=/tmp
python bench.py > $O/bench_cfg3_n1.json 2> $O/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-extras > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err )
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_cfg3_kernel_stats.csv 2>/dev/null
python scripts/bench_configs.py > $O/configs.json 2> $O/configs.err
for n in 2 4 8; do python bench.py --emulate-world $n --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1; done > $O/emulated_shards.jsonl
GVQA_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 > $O/bench_cfg3_rccl_1rank.json
python scripts/bench_split3.py $O/split_variants.json > /dev/null 2>&1
for v in 113 13; do VAR=$v python scripts/bench_split3_loop.py 2>/dev/null | grep variant | tail -6; done > $O/split_loop_parts.jsonl
python scripts/bench_fused_debug.py 2>/dev/null | grep debug > $O/fused_epilogue_parts.jsonl
python scripts/bench_pack.py 2>/dev/null | grep copy_us > $O/pack.jsonl
bash scripts/pmc_fused.sh $O/pmc_fused 0 > $O/pmc_fused.txt 2>&1
rm -rf $O/prof $O/pmc_fused
ls -la $O
