#!/bin/bash
# the one-launch hop kernel with the ping-pong K step (measurement build lib/pp: GVQA_HA_PP=1) against the shipped step: parity, then same-box A/B
O=gpurun_out/r06g; mkdir -p $O
PPL=$PWD/graphvqa_amd/lib/pp/libgvqa_hip.so
GVQA_LIB=$PPL timeout 900 python -m pytest tests/test_gpu_gat.py -x -q -k "config3 or hopagg or one_launch or aggregate or packed or randomized_fused or config2" 2>&1 | tail -3 > $O/pp_tests.txt
for v in "" $PPL "" $PPL "" $PPL; do
  if [ -z "$v" ]; then unset GVQA_LIB; else export GVQA_LIB=$v; fi
  python bench.py --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'lib': 'pp' if '$v' else 'product', 'ms_per_step': round(d['ms_per_step'],4), 'hop_us': round(d['roofline']['avg_launch_us'],1), 'issued_tflops': round(d['roofline']['issued_tflops'],1)}))"
done > $O/pp_ab.jsonl
