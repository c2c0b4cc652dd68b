#!/bin/bash
# config 3, the one-launch hop kernel's end-of-step synchronisation taken apart (measurement build; results wrong by construction):
# 16 = neither the counted wait nor the barrier, 64 = barrier without the DMA wait, 128 = waits without the barrier
O=gpurun_out/r06g; mkdir -p $O
export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so GVQA_HOPAGG_STAMPS=/tmp/ha_stamps.bin
for d in 0 16 64 128 0; do GVQA_HOPAGG_DEBUG=$d D=512 GRAPHS=2048 timeout 120 python scripts/probe_hopagg_seq.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['hop1 [mean, max] us']; print(json.dumps({'debug': $d, 'main_loop_us': h['main loop'], 'hop_total_us': h['hop total'], 'span_us': d['span_us']}))"; done > $O/cfg3_sync_parts_stamps.jsonl
