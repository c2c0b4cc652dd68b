#!/usr/bin/env python3
"""Randomised parity sweep of gat_seq's eval forward against the oracle (tests/fuzz.py): prints one line per failing case and a
summary line.  SEED=<int> CASES=<n> python scripts/fuzz_gat_seq.py   (round 3: seeds 1-4, 7-9, 4150 cases, 0 failures)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from tests.fuzz import case, run, STATS


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
    n, bad = int(os.environ.get("CASES", "200")), 0
    for i in range(n):
        c = case(rng)
        try:
            ok, errs, sz = run(c, dev)
        except Exception as e:      # an error return is a finding too
            ok, errs, sz = False, {"exception": str(e)[:200]}, None
        if not ok:
            bad += 1
            print("FAIL", json.dumps(c), errs, sz, flush=True)
    print(json.dumps({"cases": n, "failed": bad, "seed": int(os.environ.get("SEED", "1")), "bound_census": STATS}))


if __name__ == "__main__":
    main()
