#!/usr/bin/env python3
"""BASELINE config 4 alone: the five GINEConv layers of gine_seq on the config-2 batch (return_convs=True), wall ms per forward and the
in-library stage split.  GVQA_GINE_FUSED=0: the unfused Lin -> ReLU -> Lin of round 5.  Under rocprofv3 --kernel-trace --stats: per-kernel times."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.graph import SceneGraphBatch
from graphvqa_amd.baseline_models import gine_seq
dev = torch.device("cuda:0")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
gb = synth.config2_batch(); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
x, ea, ins = tt(synth.normal((N, 300), 1)).to(dev), tt(synth.normal((E, 300), 2)).to(dev), tt(synth.normal((5, B, 512), 3)).to(dev)
m = gine_seq(300, 300, 512); m.load_state_dict({k: tt(v) for k, v in synth.gine_seq_params(300, 300, 512, 404).items()}); m = m.to(dev).eval()
g = SceneGraphBatch(ei, batch, N, B)
run = lambda: m(x, ei, ea, ins, batch, graph=g, return_convs=True)
for _ in range(15): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
_lib.prof_enable(True); _lib.prof_collect()
for _ in range(10): run()
torch.cuda.synchronize(); pr = _lib.prof_collect(); _lib.prof_enable(False)
fl = 5 * (2.0 * N * 812 * 300 + 2.0 * N * 300 * 300)
print(json.dumps({"fused": os.environ.get("GVQA_GINE_FUSED", "1"), "ms_per_5_convs": dt * 1e3, "stage_ms": {k: v[0] / 10 for k, v in pr.items() if v[1]},
                  "mlp_tflops_algorithmic": fl / ((pr["proj"][0] / 10) * 1e-3) / 1e12}))
