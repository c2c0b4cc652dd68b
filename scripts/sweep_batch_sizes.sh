#!/bin/bash
# forward wall time at config-3 widths over batch sizes around the default rule's thresholds: one launch for the K hops (FUSION=5) vs the
# chained 8-wave kernel (FUSION=1) vs the default rule (FUSION=3)
for g in ${GRAPHS_LIST:-1700 1760 1840 1920 2000 2048 2200 2900 3072}; do
  for f in 5 1 3; do
    GRAPHS=$g FUSION=$f python scripts/bench_hopagg.py 2>/dev/null | tail -1 | python -c 'import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({"graphs": int(sys.argv[1]), "fusion": d["fusion"], "forward_wall_ms": d["forward_wall_ms"]}))' $g
  done
done
