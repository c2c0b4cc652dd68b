import json, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.graph import SceneGraphBatch
from graphvqa_amd.baseline_models import gcn_seq
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)); dev = torch.device("cuda:0")
gb = synth.config2_batch(); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
x = tt(synth.normal((N, 300), 1)).to(dev); ins = tt(synth.normal((5, B, 512), 3)).to(dev)
m = gcn_seq(300, 300, 512); m.load_state_dict({k: tt(v) for k, v in synth.gcn_seq_params(300, 300, 512, 505).items()}); m = m.to(dev).eval()
g = SceneGraphBatch(ei, batch, N, B)
for _ in range(3): m(x, ei, ins, batch, graph=g, return_convs=True)
torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_collect()
for _ in range(10): m(x, ei, ins, batch, graph=g, return_convs=True)
torch.cuda.synchronize(); pr = _lib.prof_collect(); _lib.prof_enable(False)
us = pr["mp"][0] / pr["mp"][1] * 1e3
print(json.dumps({"gcn_aggregate_us": round(us, 2), "GBps": round((8 * N * 300 + 4 * E) / (us * 1e-6) / 1e9, 1)}))
