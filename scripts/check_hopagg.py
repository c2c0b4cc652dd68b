#!/usr/bin/env python3
"""Aggregate-first hop kernel (GVQA_OPT_HOP_FUSION = 4) against the oracle on random H = 4 cases + timing at config 3."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from tests.fuzz import stratified_case, run, STRATA
from graphvqa_amd import _lib

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
bad = n = 0
for st in STRATA:
    for K in (1, 2, 3, 5):
        for _ in range(int(os.environ.get("REPS", "2"))):
            c = stratified_case(rng, 4, st, K)
            c["H"] = 4
            if c["C"] > 512 or c["C"] % 4: c["C"] = 64
            _lib.prof_enable(True); _lib.prof_collect()
            try:
                ok, errs, sz = run(c, dev)
            except Exception as e:
                ok, errs, sz = False, {"exception": str(e)[:300]}, None
            pr = _lib.prof_collect(); _lib.prof_enable(False)
            n += 1
            took = pr["proj"][1]
            if not ok:
                bad += 1
                print("FAIL", json.dumps(c), errs, sz, flush=True)
            elif os.environ.get("VERBOSE"):
                print("ok", st, K, c["C"], errs, sz, "proj launches", took, "mp", pr["mp"][1], flush=True)
print(json.dumps({"cases": n, "failed": bad}))
