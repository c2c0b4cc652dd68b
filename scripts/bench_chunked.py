#!/usr/bin/env python3
"""Does processing the batch in cache-sized chunks of graphs (all K hops per chunk) beat one pass over the whole batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)      # inference measurements: fused path
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
dev = torch.device("cuda:0"); tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
D, H, K = 512, 4, 5
gb = synth.config3_batch(); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.gat_seq_params(D, D, D, D, K, H, seed=777).items()}); m = m.to(dev).eval()
x, ea, ins = tt(synth.normal((N, D), 1)), tt(synth.normal((E, D), 2)), tt(synth.normal((K, B, D), 3))
ei, batch = tt(gb.edge_index), tt(gb.batch)
def timed(fn, w=3, s=10):
    for _ in range(w): fn()
    torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_collect(); t0 = time.perf_counter()
    for _ in range(s): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / s * 1e3
    pr = _lib.prof_collect(); _lib.prof_enable(False)
    timed.stages = {k: round(v[0] / s, 3) for k, v in pr.items() if v[1]}
    return dt
full = m(x, ei, ea, ins, batch)
print(f"whole batch: {timed(lambda: m(x, ei, ea, ins, batch)):.3f} ms", timed.stages)
for nchunk in (2, 4, 8, 16):
    gpc = B // nchunk; npc, epc = gpc * 32, gpc * 128
    parts = [(x[c*npc:(c+1)*npc], (ei[:, c*epc:(c+1)*epc] - c*npc).contiguous(), ea[c*epc:(c+1)*epc],
              ins[:, c*gpc:(c+1)*gpc].contiguous(), (batch[c*npc:(c+1)*npc] - c*gpc).contiguous()) for c in range(nchunk)]
    def run():
        return [m(*p) for p in parts]
    outs = torch.cat(run())
    print(f"{nchunk:2d} chunks of {gpc} graphs ({npc*H*D*4/1e6:.0f} MB xp each): {timed(run):.3f} ms   max|diff| {float((outs-full).abs().max()):.2e}", timed.stages)
