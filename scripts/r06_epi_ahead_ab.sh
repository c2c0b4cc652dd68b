#!/bin/bash
# same-box A/B: batches of epilogue loads in flight ahead of the arithmetic in the one-launch hop kernel (1 = round 5's; 2 = product; 3)
O=gpurun_out/r06g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gat.py -x -q -k "config3 or hopagg or one_launch or aggregate or packed or randomized_fused" 2>&1 | tail -2 > $O/epi_ahead_tests.txt
for v in ea1 "" ea3 ea1 "" ea3 ""; do
  if [ -z "$v" ]; then unset GVQA_LIB; else export GVQA_LIB=$PWD/graphvqa_amd/lib/$v/libgvqa_hip.so; fi
  python bench.py --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'lib': '${v:-product_ea2}', 'ms_per_step': round(d['ms_per_step'],4), 'hop_us': round(d['roofline']['avg_launch_us'],1)}))"
done > $O/epi_ahead_ab.jsonl
