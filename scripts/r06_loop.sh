#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gat.py -x -q -k "aggregate_first or packed or default_rule or config3_full_batch" 2>&1 | tail -3 > $O/loop_tests.txt
for i in 1 2; do for c in 3 2; do CONFIG=$c FUSION=3 python scripts/bench_hopagg.py 2>/dev/null | tail -1; done; done > $O/loop_ab.jsonl
