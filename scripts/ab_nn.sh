#!/bin/bash
# direct input-gradient product: product build against ablation builds (python -m graphvqa_amd.build --variant nnd<bits> GVQA_NND_DBG=<bits>:
# 1 no A loads in the loop, 2 no split / A writes, 4 no MFMAs, 8 no B loads / writes) on one box; ablated results are wrong by construction
for lib in graphvqa_amd/lib/libgvqa_hip.so ${AB_LIBS}; do
  echo "== $lib"; GVQA_LIB=$lib python scripts/bench_tn.py 2>/dev/null | grep '"direct": 1' | cut -c1-100
done
