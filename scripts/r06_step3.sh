#!/bin/bash
O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bench_faults.py -x -q 2>&1 | tail -12 > $O/bench_fault_tests.txt
for v in "" graphvqa_amd/lib/gm_shallow/libgvqa_hip.so "" graphvqa_amd/lib/gm_shallow/libgvqa_hip.so; do GVQA_LIB=$v python scripts/bench_gine.py 2>/dev/null | tail -1 | sed -e "s|^{|{\"lib\": \"${v:-product}\", |"; done > $O/gine_deep_ab.jsonl
python bench.py --no-cpu-baseline --no-pmc > $O/bench_quick.json 2> $O/bench_quick.err
timeout 300 python -m pytest tests/test_gpu_gat.py -x -q -k "gine or bn_relu" 2>&1 | tail -2 >> $O/bench_fault_tests.txt
