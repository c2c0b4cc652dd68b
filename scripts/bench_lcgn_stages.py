#!/usr/bin/env python3
"""LCGN forward at the config-5 shape (config-2 batch, O = 512, 4 iterations): wall ms and in-library stage ms, fp32 and bf16 node features."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.graph import SceneGraphBatch
from graphvqa_amd.lcgn import lcgn_seq
dev = torch.device("cuda:0")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
gb = synth.config2_batch()
N, E, B, O = gb.num_nodes, gb.num_edges, gb.num_graphs, 512
ei, batch = tt(gb.edge_index), tt(gb.batch)
x, q, lstm, xc = tt(synth.normal((N, 300), 1)), tt(synth.normal((B, O), 5)), tt(synth.normal((10, B, O), 6)), tt(synth.normal((N, O), 7))
g = SceneGraphBatch(ei, batch, N, B)
for key, kw in (("fp32", {}), ("bf16", {"node_feature_dtype": torch.bfloat16})):
    m = lcgn_seq(300, O, 300, 5, **kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.lcgn_seq_params(300, O, seed=808).items()})
    m = m.to(dev).eval()
    run = lambda: m(x, ei, batch, q, lstm, graph=g, x_ctx_init=xc)
    for _ in range(12): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    _lib.prof_enable(True); _lib.prof_collect()
    for _ in range(10): run()
    torch.cuda.synchronize(); pr = _lib.prof_collect(); _lib.prof_enable(False)
    print(json.dumps({"mode": key, "ms": round(dt * 1e3, 4), "stage_ms": {k: round(v[0] / 10, 4) for k, v in pr.items() if v[1]},
                      "launches": {k: v[1] // 10 for k, v in pr.items() if v[1]}}))
