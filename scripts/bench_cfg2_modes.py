#!/usr/bin/env python3
"""BASELINE config 2 (GAT, d = 300, 1000 graphs, K = 5) forward under each hop kernel (gat_seq.hop_fusion = 1: 8-wave fused kernel,
2: persistent chained kernel, 0: unfused), with the per-stage times.  One JSON line per mode."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.graph import SceneGraphBatch, HostLayout
from graphvqa_amd.gat_skip import gat_seq
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)); dev = torch.device("cuda:0")
gb = synth.config3_batch() if os.environ.get("BATCH") == "3" else synth.config2_batch()      # BATCH=3: config-3 batch at config-2 widths (the pipeline bench)
N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
x, ea = tt(synth.normal((N, 300), 1)).to(dev), tt(synth.normal((E, 300), 2)).to(dev)
ins = tt(synth.normal((5, B, 512), 3)).to(dev)
m = gat_seq(300, 300, 300, 512, 5, dropout=0.1, gat_heads=4)
m.load_state_dict({k: tt(v) for k, v in synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=303).items()}); m = m.to(dev).eval()
hl = HostLayout.from_numpy(gb.edge_index, gb.batch, B)
run = lambda: m(x, ei, ea, ins, batch, graph=SceneGraphBatch(ei, batch, N, B, host_layout=hl))
for mode in [int(v) for v in os.environ.get("MODES", "1,2,0").split(",")]:
    m.hop_fusion = mode % 10
    _lib.set_option(_lib.OPT_MP_PARTS, mode // 10)      # MODES=20: unfused with two blocks per graph in the message-passing kernel
    for _ in range(5): run()
    torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_collect()
    steps = 20; t0 = time.perf_counter()
    for _ in range(steps): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    prof = _lib.prof_collect(); _lib.prof_enable(False)
    print(json.dumps({"hop_fusion": mode, "ms_per_forward": round(dt * 1e3, 4),
                      "stage_us": {k: [round(v[0] / steps * 1e3, 1), v[1] // steps] for k, v in prof.items() if v[1]}}))
