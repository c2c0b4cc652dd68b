#!/usr/bin/env python3
"""Small-batch (serving) latency of gat_seq: eager launches vs the whole forward captured in a HIP graph
(torch.cuda.CUDAGraph == hipGraph on ROCm; the library only enqueues on the caller's stream, so it is capturable once
the batch handle -- whose construction reads statistics back -- has been built outside the capture)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch
dev = torch.device("cuda:0"); tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
D, K, H = 512, 5, 4
m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.gat_seq_params(D, D, D, D, K, H, seed=7).items()}); m = m.to(dev).eval()
res = []
for B in (1, 8, 64, 256, 2048):
    gb = synth.make_graph_batch(B, seed=5 + B, fixed_nodes=32, fixed_rel=96)
    N, E = gb.num_nodes, gb.num_edges
    x, ea, ins = tt(synth.normal((N, D), 1)), tt(synth.normal((E, D), 2)), tt(synth.normal((K, B, D), 3))
    ei, b = tt(gb.edge_index), tt(gb.batch)
    g = SceneGraphBatch(ei, b, N, B)
    def timed(fn, iters=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
    eager_full = timed(lambda: m(x, ei, ea, ins, b))                      # CSR build inside
    eager = timed(lambda: m(x, ei, ea, ins, b, graph=g))
    ref = m(x, ei, ea, ins, b, graph=g)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): m(x, ei, ea, ins, b, graph=g)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, stream=s):
        out = m(x, ei, ea, ins, b, graph=g)
    graphed = timed(cg.replay)
    cg.replay(); torch.cuda.synchronize()
    res.append({"B": B, "N": N, "E": E, "eager_with_csr_build_ms": round(eager_full, 4), "eager_prebuilt_graph_ms": round(eager, 4),
                "hip_graph_replay_ms": round(graphed, 4), "max_abs_graph_vs_eager": float((out - ref).abs().max())})
    print(json.dumps(res[-1]), flush=True)
