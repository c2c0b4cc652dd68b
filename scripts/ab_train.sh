#!/bin/bash
# training step (scripts/bench_train.py, config 3 and CONFIG=2) with the weight-gradient product reading the operands directly (GVQA_TN_DIRECT=1)
# and with transposed packs first (0), alternated on one box
for r in 1 2; do for d in 1 0; do
  echo "direct=$d cfg3 $(GVQA_TN_DIRECT=$d TRAIN_ONLY=1 python scripts/bench_train.py 2>/dev/null | tail -1 | cut -c1-90)"
  echo "direct=$d cfg2 $(GVQA_TN_DIRECT=$d TRAIN_ONLY=1 CONFIG=2 python scripts/bench_train.py 2>/dev/null | tail -1 | cut -c1-90)"
done; done
