#!/usr/bin/env python3
"""torch.profiler view of one config-3 training step of gat_seq: the aten ops (with shapes) that still run between the library calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from graphvqa_amd import synth
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch
dev = torch.device("cuda:0"); tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
H, K, D, DI = 4, 5, 512, 512
gb = synth.config3_batch(); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
m = gat_seq(D, D, D, DI, K, dropout=0.1, gat_heads=H)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.gat_seq_params(D, D, D, DI, K, H, seed=777).items()}); m = m.to(dev).train()
x, ea, ins = tt(synth.normal((N, D), 1)), tt(synth.normal((E, D), 2)), tt(synth.normal((K, B, DI), 3))
ei, batch = tt(gb.edge_index), tt(gb.batch)
g = SceneGraphBatch(ei, batch, N, B); g.transposed()
opt = torch.optim.SGD(m.parameters(), lr=1e-3)
def step():
    opt.zero_grad(set_to_none=True)
    m(x, ei, ea, ins, batch, graph=g).square().mean().backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and (getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)) > 0]
t = lambda e: getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
rows.sort(key=lambda e: -t(e))
tot = sum(t(e) for e in rows)
print("aten self device time per step: %.0f us" % tot)
for e in rows[:40]:
    print("%8.1f us  x%-3d %-28s %s" % (t(e), e.count, e.key, str(e.input_shapes)[:110]))
