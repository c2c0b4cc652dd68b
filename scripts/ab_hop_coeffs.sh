#!/bin/bash
# A/B of the chained 8-wave hop with in-kernel coefficients (GVQA_OPT_HOP_COEFFS) on a 256-graph shard, with the measurement build's
# ablation bits (GVQA_FUSED_DEBUG: 2 no aggregation, 16 no next-hop logits, 32 no coefficient phase): scripts/ab_hop_coeffs.sh
export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so
one() { python bench.py --emulate-world 8 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | grep emulated_world | python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],4), d['gpu_stage_ms_per_step'])"; }
for d in 0 16 32 48 2 50; do GVQA_FUSED_DEBUG=$d one "coeffs1_dbg$d"; done
GVQA_HOP_COEFFS=0 one coeffs0
GVQA_HOP_COEFFS=0 GVQA_FUSED_DEBUG=2 one coeffs0_dbg2
