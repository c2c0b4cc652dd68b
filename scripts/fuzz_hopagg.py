#!/usr/bin/env python3
"""Randomised parity sweep of the aggregate-first hop kernel's forms (GVQA_OPT_HOP_FUSION 4: one launch per hop, 5: one launch for the K
hops) against the oracle: every batch regime of tests/fuzz.py (tiny groups, sparse, hubs with in-degrees of tens to hundreds -- the
register / LDS-slice overflow paths --, partial k blocks, many small graphs, big graphs) x K = 1..5 x widths 32..512, H = 4.

    python scripts/fuzz_hopagg.py [cases_per_cell=4] [seed=1]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.fuzz import stratified_case, run, STRATA
per = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
bad, n, t0, worst = [], 0, time.time(), 0.0
for fusion in (4, 5):
    for stratum in STRATA:
        for K in range(1, 6):
            for _ in range(per):
                c = stratified_case(rng, fusion, stratum, K)
                c["H"] = 4
                c["C"] = int(rng.choice([32, 64, 100, 128, 256, 300, 304, 320, 384, 512]))
                if c["C"] % 4:
                    c["C"] += 4 - c["C"] % 4
                ok, errs, sz = run(c, dev)
                n += 1
                worst = max(worst, errs["out"], errs["out2"])
                if not ok:
                    bad.append({"case": {k: (v if not isinstance(v, (np.integer, np.floating)) else v.item()) for k, v in c.items()}, "errs": errs, "size": sz})
print(json.dumps({"cases": n, "failed": len(bad), "worst_max_abs": worst, "seconds": round(time.time() - t0, 1), "first_failures": bad[:3]}))
sys.exit(1 if bad else 0)
