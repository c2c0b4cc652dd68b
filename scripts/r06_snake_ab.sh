#!/bin/bash
# same-box A/B: the hop kernel's products in snake order (every two consecutive MFMAs share an operand; the matrix-rate probe says +4 % for a bare MFMA stream)
O=gpurun_out/r06g; mkdir -p $O
GVQA_LIB=$PWD/graphvqa_amd/lib/snake/libgvqa_hip.so timeout 600 python -m pytest tests/test_gpu_gat.py -x -q -k "config3_full_batch or hopagg or one_launch" 2>&1 | tail -2 > $O/snake_tests.txt
for v in "" snake "" snake "" snake; do
  if [ -z "$v" ]; then unset GVQA_LIB; else export GVQA_LIB=$PWD/graphvqa_amd/lib/$v/libgvqa_hip.so; fi
  python bench.py --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'lib': '${v:-product}', 'ms_per_step': round(d['ms_per_step'],4), 'hop_us': round(d['roofline']['avg_launch_us'],1)}))"
done > $O/snake_ab.jsonl
