#!/usr/bin/env python3
"""Config-3 steps (CSR build + gat_seq eval forward, as bench.py times them) issued on ONE stream against the same steps alternated over
TWO streams (two module instances: own workspace / weight cache each; batches are independent, nothing joins the streams): does the next
batch's front end (edge logits, layout pass, instruction terms: HBM-bound) fill the tail of the previous batch's hop launch?

    python scripts/bench_two_streams.py [steps=40] [rounds=3]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphvqa_amd import synth
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch
from graphvqa_amd.parallel import BatchShard

D, H, K = 512, 4, 5
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
params = synth.gat_seq_params(D, D, D, D, K, H, seed=777)
models = []
for _ in range(2):
    m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H)
    m.load_state_dict({k: tt(v) for k, v in params.items()})
    models.append(m.to(dev).eval())
gb = synth.make_graph_batch(2048, seed=0x5EED0003, fixed_nodes=32, fixed_rel=96)
x, ea, ins = synth.normal((gb.num_nodes, D), 1), synth.normal((gb.num_edges, D), 2), synth.normal((K, gb.num_graphs, D), 3)
s = BatchShard(gb.edge_index, gb.batch, gb.num_graphs, x, ea, ins, 0, 1, dev)
hl = s.host_layout()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def step(i, nstreams):
    j = i % nstreams
    with torch.cuda.stream(streams[j]):
        g = SceneGraphBatch(s.edge_index, s.batch, s.num_nodes, s.num_graphs, host_layout=hl)
        return models[j](s.x, s.edge_index, s.edge_attr, s.instr, s.batch, graph=g)


def timed(nstreams):
    for i in range(6):
        step(i, nstreams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i, nstreams)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


ref = None
res = {"steps": steps, "one_stream_ms": [], "two_streams_ms": []}
for r in range(rounds):
    for n, key in ((1, "one_stream_ms"), (2, "two_streams_ms")):
        ms, out = timed(n)
        res[key].append(round(ms, 4))
        if ref is None:
            ref = out.clone()
        res["max_abs_dev_between_forms"] = max(res.get("max_abs_dev_between_forms", 0.0), float((out - ref).abs().max()))
res["edges_per_s_one"] = gb.num_edges / (min(res["one_stream_ms"]) * 1e-3)
res["edges_per_s_two"] = gb.num_edges / (min(res["two_streams_ms"]) * 1e-3)
print(json.dumps(res))
