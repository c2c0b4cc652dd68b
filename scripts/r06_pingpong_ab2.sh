#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
for v in "" pp pp0 pp1 "" pp pp0 pp1; do
  if [ -z "$v" ]; then unset GVQA_LIB; else export GVQA_LIB=$PWD/graphvqa_amd/lib/$v/libgvqa_hip.so; fi
  python bench.py --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'lib': '${v:-product}', 'ms_per_step': round(d['ms_per_step'],4), 'hop_us': round(d['roofline']['avg_launch_us'],1)}))"
done > $O/pp_ab2.jsonl
