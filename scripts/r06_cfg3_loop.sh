#!/bin/bash
# config 3 (d = 512, 512 row groups: two rounds): main-loop time of the one-launch aggregate-first kernel with parts of the K step switched off
# (GVQA_HOPAGG_DEBUG bits: 1 no producer, 2 no weight DMAs, 4 no MFMAs, 16 no counted wait + barrier at the end of the step; results are wrong by construction)
O=gpurun_out/r06d; mkdir -p $O
export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so GVQA_HOPAGG_STAMPS=/tmp/ha_stamps.bin
for g in 2048 1024; do
for d in 0 1 2 4 5 8 16 21 0; do GVQA_HOPAGG_DEBUG=$d D=512 GRAPHS=$g python scripts/probe_hopagg_seq.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['hop1 [mean, max] us']; print(json.dumps({'graphs': $g, 'debug': $d, 'main_loop_us': h['main loop'], 'hop_total_us': h['hop total'], 'span_us': d['span_us']}))"; done; done > $O/cfg3_loop_parts_stamps.jsonl
