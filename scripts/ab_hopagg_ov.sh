#!/bin/bash
# the aggregate-first hop kernel with the K step specialised for waves without overflow edges (product build) against the round-4 body
# (variant build: python -m graphvqa_amd.build --variant ovall GVQA_HA_OV_ALWAYS=1), alternated on one box: scripts/ab_hopagg_ov.sh
for r in 1 2 3; do
  for lib in graphvqa_amd/lib/libgvqa_hip.so ${AB_LIBS:-graphvqa_amd/lib/ovall/libgvqa_hip.so}; do
    GVQA_LIB=$lib python bench.py --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$lib', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), d['stage_ms_per_step'].get('proj'))"
  done
done
