#!/bin/bash
# training-path evidence only (the tail of scripts/collect_profiles.sh): step times, kernel stats, A/B of every round-5 change, aten ops left, config 2
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
python scripts/bench_train.py 2>/dev/null | tail -1 > $O/train_cfg3.json
CONFIG=2 TRAIN_ONLY=1 python scripts/bench_train.py 2>/dev/null | tail -1 > $O/train_cfg2.json
python scripts/bench_tn.py 2>/dev/null | grep "^{" > $O/train_tn_direct.jsonl
bash scripts/ab_train_parts.sh 2>/dev/null > $O/train_parts_ab.txt
python scripts/prof_train_ops.py 2>/dev/null | grep -v Warning | tail -34 > $O/train_aten_ops.txt
( cd /tmp && TRAIN_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/tprof -o ks -- python $GRAFT_REPO_ROOT/scripts/bench_train.py > /dev/null 2>&1 )
cp $(find $O/tprof -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv 2>/dev/null; rm -rf $O/tprof
python scripts/bench_pipeline.py 2>/dev/null | tail -1 > $O/pipeline_cfg3_batch.json
python scripts/bench_pipeline_train.py 2>/dev/null | tail -1 > $O/pipeline_train_cfg3_batch.json
cat $O/train_cfg3.json $O/train_cfg2.json $O/train_parts_ab.txt $O/pipeline_cfg3_batch.json $O/pipeline_train_cfg3_batch.json
