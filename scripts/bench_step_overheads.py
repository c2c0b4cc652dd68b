"""Where the config-3 step's wall time goes beyond its kernels: CSR build + finalize (host sync) vs a prebuilt batch handle."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch
torch.set_grad_enabled(False)
dev = torch.device("cuda:0"); D, H, K = 512, 4, 5
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
nb = int(os.environ.get("NB", "2048"))
gb = synth.config3_batch(nb); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H); m.load_state_dict({k: tt(v) for k, v in synth.gat_seq_params(D, D, D, D, K, H, seed=777).items()}); m = m.to(dev).eval()
x, ea, ins = tt(synth.normal((N, D), 1)).to(dev), tt(synth.normal((E, D), 2)).to(dev), tt(synth.normal((K, B, D), 3)).to(dev)
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
def timeit(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
g = SceneGraphBatch(ei, batch, N, B)
res = {"graphs": nb,
       "step_with_build_ms": timeit(lambda: m(x, ei, ea, ins, batch, graph=SceneGraphBatch(ei, batch, N, B))),
       "step_prebuilt_ms": timeit(lambda: m(x, ei, ea, ins, batch, graph=g)),
       "build_only_ms": timeit(lambda: SceneGraphBatch(ei, batch, N, B))}
_lib.prof_enable(True); _lib.prof_collect()
for _ in range(10): m(x, ei, ea, ins, batch, graph=g)
torch.cuda.synchronize(); p = _lib.prof_collect(); _lib.prof_enable(False)
res["kernel_sum_ms"] = sum(v[0] for v in p.values()) / 10; res["stages_us_per_step"] = {k: round(v[0] * 100, 1) for k, v in p.items() if v[1]}
print(json.dumps(res))
