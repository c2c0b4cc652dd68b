#!/bin/bash
# direct weight-gradient product: product build against the ablation builds (python -m graphvqa_amd.build --variant tnd<bits> GVQA_TND_DBG=<bits>:
# 1 no loads in the loop, 2 no split / image writes, 4 no MFMAs) on one box; results of the ablated builds are wrong by construction
for lib in graphvqa_amd/lib/libgvqa_hip.so ${AB_LIBS}; do
  echo "== $lib"; GVQA_LIB=$lib python scripts/bench_tn.py 2>/dev/null | grep '"direct": 1' | cut -c1-120
done
