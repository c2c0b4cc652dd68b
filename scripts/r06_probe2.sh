#!/bin/bash
O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
( export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so GVQA_HOPAGG_STAMPS=/tmp/ha_stamps.bin; D=300 python scripts/probe_hopagg_seq.py 2>/dev/null | tail -1 ) > $O/cfg2_seq_stamps.json
for ov in 0 1; do GVQA_OVERLAP=$ov CONFIG=2 FUSION=3 python scripts/bench_hopagg.py 2>/dev/null | tail -1 | sed -e "s/^{/{\"overlap\": $ov, /"; done > $O/cfg2_overlap_ab.jsonl
for ov in 0 1 0 1; do GVQA_OVERLAP=$ov python bench.py --emulate-world 8 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 | sed -e "s/^{/{\"overlap\": $ov, /"; done > $O/shard8_overlap_ab.jsonl
