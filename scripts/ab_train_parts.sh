#!/bin/bash
# training step at config 3 with the round-5 fusions switched off one at a time (GVQA_TRAIN_AB bits, graphvqa_amd/gat_skip.py; GVQA_TN_DIRECT=0: both
# gradient products of the projection from packed operands; bit 32 = dropout masks as torch tensors; 64 = node logits as their own pass; 111 = all off = round 4; bit 16 = the built-but-slower
# single-epilogue dh ON), alternated on one box
for r in 1 2; do
  for ab in 0 1 2 4 8 32 64 111 16; do echo "ab=$ab $(GVQA_TRAIN_AB=$ab TRAIN_ONLY=1 python scripts/bench_train.py 2>/dev/null | tail -1 | cut -c28-65)"; done
  echo "ab=0 direct=0 $(GVQA_TN_DIRECT=0 TRAIN_ONLY=1 python scripts/bench_train.py 2>/dev/null | tail -1 | cut -c28-65)"
  echo "ab=111 direct=0 $(GVQA_TRAIN_AB=111 GVQA_TN_DIRECT=0 TRAIN_ONLY=1 python scripts/bench_train.py 2>/dev/null | tail -1 | cut -c28-65)"
done
