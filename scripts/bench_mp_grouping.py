#!/usr/bin/env python3
"""Message-passing op of the differentiable path (forward + backward kernels) with the batch partitioned into coarser units than
graphs: blocks of 1, 2 or 4 consecutive graphs (edges never leave a graph, so any union of whole graphs is a valid unit for the
one-block-per-unit kernels).  Config-3 and config-2 sizes; one JSON line per case."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphvqa_amd import synth
from graphvqa_amd.gat_skip import gat_message_passing
from graphvqa_amd.graph import SceneGraphBatch
dev = torch.device("cuda:0"); tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, gb, D, H in (("config3", synth.config3_batch(), 512, 4), ("config2", synth.config2_batch(), 300, 4)):
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    ei = tt(gb.edge_index)
    for merge in (1, 2, 4):
        batch = tt(gb.batch // merge); Bm = (B + merge - 1) // merge
        g = SceneGraphBatch(ei, batch, N, Bm); g.transposed()
        xp = torch.randn(N, H * D, device=dev, requires_grad=True); an = torch.randn(N, 2 * H, device=dev, requires_grad=True)
        ae = torch.randn(E, H, device=dev, requires_grad=True); w = torch.randn(N, D, device=dev)
        out, _ = gat_message_passing(xp, an, ae, g, H, D)
        loss = (out * w).sum()
        def bwd():
            xp.grad = an.grad = ae.grad = None
            loss.backward(retain_graph=True)
        f = timed(lambda: gat_message_passing(xp, an, ae, g, H, D)); b = timed(bwd)
        print(json.dumps({"case": name, "graphs_per_block": merge, "blocks": Bm, "mp_forward_us": round(f, 1), "mp_backward_us": round(b, 1)}))
