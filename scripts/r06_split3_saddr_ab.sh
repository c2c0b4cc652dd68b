#!/bin/bash
# same-box A/B: k_linear_split3's ring DMAs with per-lane 64-bit pointers and M0 saved / restored (lib/s3_old, the previous commit's split3.hip) against
# scalar base + constant lane offset (product): parity of everything that runs on it, LCGN fp32 forward, GINE layers, training step, the config-3 encoder
O=gpurun_out/r06d; mkdir -p $O; export TMPDIR=/tmp
OLD=$PWD/graphvqa_amd/lib/s3_old/libgvqa_hip.so
timeout 1200 python -m pytest tests/test_gpu_split3.py tests/test_gpu_gat.py -x -q -k "split or lcgn or gine or encoder or chain or linear or config2 or shard" 2>&1 | tail -3 > $O/split3_saddr_tests.txt
for v in "" $OLD "" $OLD; do
  if [ -z "$v" ]; then unset GVQA_LIB; else export GVQA_LIB=$v; fi
  L=$( [ -z "$v" ] && echo product || echo s3_old )
  a=$(python scripts/bench_lcgn.py 2>/dev/null | tail -1)
  b=$(python scripts/bench_gine.py 2>/dev/null | tail -1)
  c=$(python bench.py --emulate-world 8 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "{\"lib\": \"$L\", \"lcgn\": $a, \"gine\": $b, \"shard8_ms\": $c}"
done > $O/split3_saddr_ab.jsonl
unset GVQA_LIB
