"""BASELINE config 5 (LCGN on the config-2 batch), fp32 mode: forward time; run under rocprofv3 --kernel-trace --stats for the kernel split."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
torch.set_grad_enabled(False)
from graphvqa_amd import synth
from graphvqa_amd.graph import SceneGraphBatch
from graphvqa_amd.lcgn import lcgn_seq
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)); dev = torch.device("cuda:0")
gb = synth.config2_batch(); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
x = tt(synth.normal((N, 300), 1)).to(dev)
m = lcgn_seq(300, 512, 300, 5); m.load_state_dict({k: tt(v) for k, v in synth.lcgn_seq_params(300, 512, seed=808).items()}); m = m.to(dev).eval()
q, lstm, xc = tt(synth.normal((B, 512), 5)).to(dev), tt(synth.normal((10, B, 512), 6)).to(dev), tt(synth.normal((N, 512), 7)).to(dev)
g = SceneGraphBatch(ei, batch, N, B)
for _ in range(12): m(x, ei, batch, q, lstm, graph=g, x_ctx_init=xc)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): m(x, ei, batch, q, lstm, graph=g, x_ctx_init=xc)
torch.cuda.synchronize()
print(json.dumps({"lcgn_fp32_ms": (time.perf_counter() - t0) / 10 * 1e3}))
