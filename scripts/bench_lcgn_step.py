#!/usr/bin/env python3
"""LCGN bf16 forward INCLUDING the per-step CSR build (loader-side layout), as a serving loop would run it: A/B of the side-stream command chain
(GVQA_LCGN_OVERLAP=0|1) under the condition that broke the GAT path's side stream (profiles/r06_overlap_modes_ab.jsonl)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth
from graphvqa_amd.graph import SceneGraphBatch, HostLayout
from graphvqa_amd.lcgn import lcgn_seq
dev = torch.device("cuda:0")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
gb = synth.config2_batch()
N, E, B, O = gb.num_nodes, gb.num_edges, gb.num_graphs, 512
ei, batch = tt(gb.edge_index), tt(gb.batch)
x, q, lstm, xc = tt(synth.normal((N, 300), 1)), tt(synth.normal((B, O), 5)), tt(synth.normal((10, B, O), 6)), tt(synth.normal((N, O), 7))
hl = HostLayout.from_numpy(gb.edge_index, gb.batch, B)
m = lcgn_seq(300, O, 300, 5, node_feature_dtype=torch.bfloat16)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.lcgn_seq_params(300, O, seed=808).items()})
m = m.to(dev).eval()
def step():
    g = SceneGraphBatch(ei, batch, N, B, host_layout=hl)
    return m(x, ei, batch, q, lstm, graph=g, x_ctx_init=xc)
for _ in range(12): step()
best = None
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    best = dt if best is None else min(best, dt)
print(json.dumps({"lcgn_overlap": os.environ.get("GVQA_LCGN_OVERLAP", "1"), "ms_per_step_with_csr_build": best * 1e3}))
