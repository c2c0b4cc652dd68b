"""BASELINE config 5 shape: lcgn_seq (in=300, out=512, 4 iterations) on the config-2 batch; fp32 and bf16-node-feature modes.
Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import json, sys, time
import numpy as np, torch
torch.set_grad_enabled(False)      # inference measurements: fused path
sys.path.insert(0, ".")
from graphvqa_amd import synth
from graphvqa_amd.graph import SceneGraphBatch
from graphvqa_amd.lcgn import lcgn_seq

dev = torch.device("cuda:0")
tt = torch.from_numpy
gb = synth.config2_batch()
N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
x = tt(synth.normal((N, 300), 1)).to(dev)
q, lstm = tt(synth.normal((B, 512), 5)).to(dev), tt(synth.normal((10, B, 512), 6)).to(dev)
xc = tt(synth.normal((N, 512), 7)).to(dev)
g = SceneGraphBatch(ei, batch, N, B)

def load(m, p):
    m.load_state_dict({k: tt(v) for k, v in p.items()})
    return m.to(dev).eval()

def timed(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

res = {"N": N, "E": E, "B": B}
flops = 2.0 * N * 512 * (300 + 512 + 3 * 512 + 4 * (512 + 3 * 1024 + 1024) + 2 * 512)
for name, dt_ in (("fp32", torch.float32), ("bf16_node_features", torch.bfloat16)):
    m = load(lcgn_seq(300, 512, 300, 5, node_feature_dtype=dt_), synth.lcgn_seq_params(300, 512, seed=808))
    dt = timed(lambda: m(x, ei, batch, q, lstm, graph=g, x_ctx_init=xc))
    res[name] = {"ms_per_forward": dt * 1e3, "edges_per_s": E / dt, "node_gemm_tflops": flops / dt / 1e12}
print(json.dumps(res))
