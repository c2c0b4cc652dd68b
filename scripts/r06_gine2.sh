#!/bin/bash
O=gpurun_out/r06; mkdir -p $O; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for f in 1 0; do GVQA_GINE_FUSED=$f python scripts/bench_gine.py 2>/dev/null | tail -1; done > $O/gine_ab.jsonl
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/gprof -o ks -- python $R/scripts/bench_gine.py > /dev/null 2>&1 )
cp $(find $O/gprof -name "*kernel_stats.csv" | head -1) $O/cfg4_kernel_stats.csv 2>/dev/null; rm -rf $O/gprof
