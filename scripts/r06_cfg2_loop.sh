#!/bin/bash
# config 2 (d = 300, packed row groups: one round): main-loop time of the one-launch aggregate-first kernel with parts of the K step switched off
O=gpurun_out/r06; mkdir -p $O
export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so GVQA_HOPAGG_STAMPS=/tmp/ha_stamps.bin
for d in 0 1 2 4 5 8 16 21; do GVQA_HOPAGG_DEBUG=$d D=300 python scripts/probe_hopagg_seq.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['hop1 [mean, max] us']; print(json.dumps({'debug': $d, 'main_loop_us': h['main loop'], 'hop_total_us': h['hop total'], 'span_us': d['span_us']}))"; done > $O/cfg2_loop_parts_stamps.jsonl
