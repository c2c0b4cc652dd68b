import os, sys, time, json
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "/root/repo/bench.py"); 
# reuse bench internals is awkward; do a direct measurement instead
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch, HostLayout
from graphvqa_amd.parallel import BatchShard
dev = torch.device("cuda:0"); D, H, K = 512, 4, 5
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
for nb in (2048, 256):
    gb = synth.config3_batch(nb); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H); m.load_state_dict({k: tt(v) for k, v in synth.gat_seq_params(D, D, D, D, K, H, seed=777).items()}); m = m.to(dev).eval()
    x, ea, ins = tt(synth.normal((N, D), 1)).to(dev), tt(synth.normal((E, D), 2)).to(dev), tt(synth.normal((K, B, D), 3)).to(dev)
    ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
    hl = HostLayout.from_numpy(gb.edge_index, gb.batch, B)
    def step():
        g = SceneGraphBatch(ei, batch, N, B, host_layout=hl)
        return m(x, ei, ea, ins, batch, graph=g)
    for _ in range(5): step()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"graphs": nb, "cpu_issue_ms_per_step": (t1 - t0) / n * 1e3, "wall_ms_per_step": (t2 - t0) / n * 1e3}))
import cProfile, pstats, io
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(200): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
