#!/usr/bin/env python3
"""Hop-kernel time of the aggregate-first path at config 3 (stage timers), optionally with parts of a K step switched off in the
measurement build: GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so GVQA_HOPAGG_DEBUG=<mask> python scripts/bench_hopagg.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch, HostLayout
CONFIG = int(os.environ.get("CONFIG", "3"))
D, H, K = int(os.environ.get("D", "512" if CONFIG == 3 else "300")), 4, 5
DI = 512
dev = torch.device("cuda:0")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
gb = synth.config3_batch(int(os.environ.get("GRAPHS", "2048"))) if CONFIG == 3 else synth.config2_batch()
N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
m = gat_seq(D, D, D, DI, K, dropout=0.1, gat_heads=H)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.gat_seq_params(D, D, D, DI, K, H, seed=777).items()})
m = m.to(dev).eval()
m.hop_fusion = int(os.environ.get("FUSION", "4"))
x, ea, ins = tt(synth.normal((N, D), 1)), tt(synth.normal((E, D), 2)), tt(synth.normal((K, B, DI), 3))
ei, bt = tt(gb.edge_index), tt(gb.batch)
g = SceneGraphBatch(ei, bt, N, B, host_layout=HostLayout.from_numpy(gb.edge_index, gb.batch, B))
for _ in range(3): m(x, ei, ea, ins, bt, graph=g)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): m(x, ei, ea, ins, bt, graph=g)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
_lib.prof_enable(True); _lib.prof_collect()
n = 5
for _ in range(n): m(x, ei, ea, ins, bt, graph=g)
torch.cuda.synchronize()
pr = _lib.prof_collect(); _lib.prof_enable(False)
print(json.dumps({"config": CONFIG, "forward_wall_ms": round(wall, 4), "stages_us": {k: round(v[0] / n * 1e3, 1) for k, v in pr.items() if v[1]}, "debug": os.environ.get("GVQA_HOPAGG_DEBUG", "0"), "fusion": m.hop_fusion, "hop_kernel_us": round(pr["proj"][0] / pr["proj"][1] * 1e3, 1),
                  "alpha_us": round(pr["alpha"][0] / max(pr["alpha"][1], 1) * 1e3, 1), "pack_us": round(pr["pack"][0] / max(pr["pack"][1], 1) * 1e3, 1)}))
