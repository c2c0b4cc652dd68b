#!/bin/bash
# aggregate-first hop at config 2 (d = 300, the 4 x 2-wave layout with five column tiles per wave): hop-kernel us with parts of the K step
# switched off (measurement build; GVQA_HOPAGG_DEBUG: 1 no producer, 2 no weight DMA, 4 no MFMAs, 16 no waits / barriers, 32 no epilogue)
export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so
for d in 0 1 4 5 7 21 23 32 55; do CONFIG=2 FUSION=4 GVQA_HOPAGG_DEBUG=$d python scripts/bench_hopagg.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('cfg2 dbg $d hop us', d['hop_kernel_us'], 'wall', d['forward_wall_ms'])"; done
CONFIG=2 FUSION=1 python scripts/bench_hopagg.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('cfg2 8-wave hop us', d['hop_kernel_us'], 'alpha', d['alpha_us'], 'wall', d['forward_wall_ms'])"
