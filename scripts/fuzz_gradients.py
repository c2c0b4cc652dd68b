#!/usr/bin/env python3
"""Randomised gradient sweep of gat_seq's differentiable path against the oracle's fp64 autograd (tests/test_gpu_backward.py's
check): random widths (incl. ones the library products do not take), head / hop counts, batch shapes, eval / batch-statistics
BatchNorm, library products forced or not.  SEED=<int> CASES=<n> python scripts/fuzz_gradients.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphvqa_amd import _lib
from tests.test_gpu_backward import _grads_vs_oracle

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
n, bad = int(os.environ.get("CASES", "60")), 0
for i in range(n):
    H = int(rng.choice([1, 2, 4, 8]))
    dn = int(rng.choice([4, 8, 30, 32, 36, 64, 100, 128, 300]))
    de, di = int(rng.choice([4, 18, 20, 32])), int(rng.choice([0, 8, 22, 48]))
    K = int(rng.integers(1, 5))
    shape = str(rng.choice(["tiny", "ragged", "big", "single", "sparse", "dense"]))
    graphs, nodes, rel = {"tiny": (int(rng.integers(2, 4)), (2, 12), 1.3), "ragged": (int(rng.integers(5, 30)), (1, 40), float(rng.uniform(0.5, 2.5))),
                          "big": (int(rng.integers(2, 4)), (90, 128), 1.0), "single": (int(rng.integers(3, 12)), (1, 2), 1.0),
                          "sparse": (int(rng.integers(10, 60)), (5, 30), float(rng.uniform(0.0, 0.5))),
                          "dense": (int(rng.integers(3, 10)), (10, 30), float(rng.uniform(3.0, 5.0)))}[shape]
    c = dict(H=H, dn=dn, de=de, di=di, K=K, shape=shape, graphs=graphs, nodes=nodes, rel=rel, train=bool(rng.integers(0, 2)),
             force=bool(rng.integers(0, 2)), seed=int(rng.integers(1, 10000)))
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0) if c["force"] else None
    try:
        _grads_vs_oracle(dev, train=c["train"], dims=(dn, de, di, K, H), seed=c["seed"], graphs=graphs, nodes=nodes, rel=rel)
    except Exception as e:
        bad += 1
        print("FAIL", json.dumps(c), type(e).__name__, str(e)[:300].replace("\n", " "), flush=True)
    finally:
        if old is not None:
            _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
print(json.dumps({"cases": n, "failed": bad}))
