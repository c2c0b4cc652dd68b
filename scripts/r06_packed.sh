#!/bin/bash
O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gat.py -x -q -k "packed or config2 or aggregate_first or default_rule" 2>&1 | tail -15 > $O/packed_tests.txt
for pk in 1 0; do for f in 5 4 3; do GVQA_PACKED_GROUPS=$pk CONFIG=2 FUSION=$f python scripts/bench_hopagg.py 2>/dev/null | tail -1 | sed -e "s/^{/{\"packed\": $pk, /"; done; done > $O/cfg2_packed_ab.jsonl
python scripts/bench_configs.py 2>/dev/null | tail -1 > $O/configs_packed.json
( export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so GVQA_HOPAGG_STAMPS=/tmp/ha_stamps.bin; D=300 python scripts/probe_hopagg_seq.py 2>/dev/null | tail -1 ) > $O/cfg2_seq_stamps_packed.json
