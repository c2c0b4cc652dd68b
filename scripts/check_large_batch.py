#!/usr/bin/env python3
"""Index-width check on a batch far beyond the benchmark's: GRAPHS (default 32768) x 32 nodes x 128 edges at d = 512, H = 4, K = 5
(1 M nodes, 4.2 M edges: N H C = 2^31 elements) through the default eval path; first / middle / last 64-graph windows against the
oracle (graphs are independent in eval mode, so the oracle runs on the window alone).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch, HostLayout
from oracle import ref_torch as R

G = int(os.environ.get("GRAPHS", "32768")); D, H, K = 512, 4, 5
dev = torch.device("cuda:0")
gb = synth.make_graph_batch(G, seed=0x5EED0003, fixed_nodes=32, fixed_rel=96)
N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
p = synth.gat_seq_params(D, D, D, D, K, H, seed=777)
m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H); m.load_state_dict({k: tt(v) for k, v in p.items()}); m = m.to(dev).eval()
gen = torch.Generator(device=dev); gen.manual_seed(5)
x = torch.randn((N, D), device=dev, generator=gen); ea = torch.randn((E, D), device=dev, generator=gen); ins = torch.randn((K, B, D), device=dev, generator=gen)
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
res = {"graphs": G, "N": N, "E": E, "N*H*C": N * H * D}
for name, graph in (("read_back", None), ("host_layout", "hl")):
    if graph == "hl":
        graph = SceneGraphBatch(ei, batch, N, B, host_layout=HostLayout.from_numpy(gb.edge_index, gb.batch, B))
    out = m(x, ei, ea, ins, batch, graph=graph)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m(x, ei, ea, ins, batch, graph=graph)
    torch.cuda.synchronize(); res[name + "_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    worst = 0.0
    for g0 in (0, B // 2 - 32, B - 64):
        n0, n1 = g0 * 32, (g0 + 64) * 32
        em = (gb.edge_index[0] >= n0) & (gb.edge_index[0] < n1)
        eiw = tt(gb.edge_index[:, em] - n0)
        ref = R.gat_seq(x[n0:n1].cpu(), eiw, ea[torch.from_numpy(em).to(dev)].cpu(), ins[:, g0:g0 + 64].cpu(), tt(gb.batch[n0:n1] - g0), {k: tt(v) for k, v in p.items()}, heads=H)
        worst = max(worst, float((out[n0:n1].cpu() - ref).abs().max()))
    res[name + "_max_abs_err"] = worst
print(json.dumps(res))
