#!/bin/bash
# the 256 x 256 bf16 kernel with four loader waves (measurement build lib/big_ldw4) against the shipped form: parity, stand-alone products, the LCGN step
O=gpurun_out/r06d; mkdir -p $O; export TMPDIR=/tmp
L=$PWD/graphvqa_amd/lib/big_ldw4/libgvqa_hip.so
GVQA_LIB=$L timeout 600 python -m pytest tests/test_gpu_gat.py -x -q -k "linear_bf16" 2>&1 | tail -3 > $O/bf16_tests_ldw4.txt
for shape in "29785 1536 1024" "29785 512 512" "29785 512 1024" "29785 1536 512"; do
  for v in "" $L "" $L; do
    if [ -z "$v" ]; then unset GVQA_LIB; else export GVQA_LIB=$v; fi
    python scripts/bench_gemm_bf16_one.py $shape 2>/dev/null | tail -1
  done
done > $O/big_loaders_ab.jsonl
unset GVQA_LIB
for v in "" $L "" $L; do
  if [ -z "$v" ]; then unset GVQA_LIB; else export GVQA_LIB=$v; fi
  python scripts/bench_lcgn_step.py 2>/dev/null | tail -1 | sed -e "s|^{|{\"lib\": \"${v:-product}\", |"
done > $O/big_loaders_lcgn_ab.jsonl
