#!/bin/bash
O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gat.py -x -q -k "gine" 2>&1 | tail -25 > $O/gine_tests.txt
for f in 1 0; do GVQA_GINE_FUSED=$f python scripts/bench_configs.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config4_gine_convs']; print(json.dumps({'fused': $f, 'ms_per_5_convs': c['ms_per_5_convs_plus_module'], 'stage_ms': c['stage_ms'], 'agg_us': c['aggregate_us_per_layer']}))"; done > $O/gine_fused_ab.jsonl
