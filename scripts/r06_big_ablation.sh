#!/bin/bash
# what the 256 x 256 bf16 kernel's K step spends outside its MFMAs: measurement builds (scripts/build_bf16_variants.sh; results of the abl builds are wrong by
# construction -- timing only).  old = per-lane pointers; skew = the two waves of a SIMD issue their DMAs in different halves of the step.
# (The GVQA_BIG_SADDR / GVQA_BIG_SKEW switches behind `old` / `skew` / `oldskew` were removed from gemm_bf16.hip once measured -- the commit
#  "bf16 256 x 256 kernel: scalar-base DMA addressing (+4 %) ..." has them; the abl* and ldw4 builds still exist: scripts/build_bf16_variants.sh)
O=gpurun_out/r06d; mkdir -p $O; export TMPDIR=/tmp
L=graphvqa_amd/lib
timeout 600 python -m pytest tests/test_gpu_gat.py -x -q -k "linear_bf16" 2>&1 | tail -3 > $O/bf16_tests2.txt
for shape in "29785 1536 1024" "29785 512 512" "29785 512 1024"; do
  for v in "" old skew oldskew abl1 abl15 "" old skew; do
    if [ -z "$v" ]; then unset GVQA_LIB; else export GVQA_LIB=$PWD/$L/big_$v/libgvqa_hip.so; fi
    python scripts/bench_gemm_bf16_one.py $shape 2>/dev/null | tail -1
  done
  unset GVQA_LIB
  python scripts/bench_gemm_bf16_one.py $shape zeros 2>/dev/null | tail -1
  GVQA_BF16_GEMM=wide python scripts/bench_gemm_bf16_one.py $shape 2>/dev/null | tail -1
done > $O/big_ablation2.jsonl
