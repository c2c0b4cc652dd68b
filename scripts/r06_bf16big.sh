#!/bin/bash
# A/B of the 256 x 256 bf16 kernel (default) against the 256 x 128 one (GVQA_BF16_GEMM=wide): parity tests, stand-alone GEMM rates, the LCGN step
O=gpurun_out/r06d; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gat.py -x -q -k "linear_bf16 or lcgn" 2>&1 | tail -5 > $O/bf16_tests.txt
for v in big wide big wide; do
  if [ $v = wide ]; then export GVQA_BF16_GEMM=wide; else unset GVQA_BF16_GEMM; fi
  python scripts/bench_gemm_bf16.py 2>/dev/null | grep '"pieces": 2' | sed -e "s|^{|{\"kernel\": \"$v\", |"
done > $O/bf16_gemm_ab.jsonl
for v in big wide big wide; do
  if [ $v = wide ]; then export GVQA_BF16_GEMM=wide; else unset GVQA_BF16_GEMM; fi
  python scripts/bench_lcgn_step.py 2>/dev/null | tail -1 | sed -e "s|^{|{\"kernel\": \"$v\", |"
done > $O/bf16_lcgn_step_ab.jsonl
unset GVQA_BF16_GEMM
