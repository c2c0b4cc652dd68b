#!/bin/bash
# config 2 (d = 300, 1000 graphs) fused forward with / without in-kernel coefficients: scripts/ab_cfg2_coeffs.sh
for c in 1 0 1 0; do GVQA_HOP_COEFFS=$c python scripts/bench_configs.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); f=d['config2_gat_d300']['fused']; g=d['config2_gat_d300_EoverN4']['fused']; print('hop_coeffs $c cfg2', round(f['ms_per_forward'],4), {k: round(v,4) for k,v in f['stage_ms'].items()}, 'E/N=4', round(g['ms_per_forward'],4))"; done
