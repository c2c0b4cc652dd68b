#!/bin/bash
# same-box A/B: the one-launch hop kernel with its DMA step offsets in two (spilled) VGPRs -- lib/ha_old, built from the previous commit's hopagg.hip --
# against scalar offsets (product)
O=gpurun_out/r06d; mkdir -p $O
for v in "" $PWD/graphvqa_amd/lib/ha_old/libgvqa_hip.so "" $PWD/graphvqa_amd/lib/ha_old/libgvqa_hip.so ""; do
  if [ -z "$v" ]; then unset GVQA_LIB; else export GVQA_LIB=$v; fi
  python bench.py --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'lib': '${v:-product}'.split('/')[-2] if '/' in '${v:-product}' else 'product', 'ms_per_step': round(d['ms_per_step'],4), 'hop_us': round(d['roofline']['avg_launch_us'],1), 'issued_tflops': round(d['roofline']['issued_tflops'],1)}))"
done > $O/hopagg_spill_ab.jsonl
