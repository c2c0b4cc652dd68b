import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch, HostLayout
dev = torch.device("cuda:0"); D, H, K = 512, 4, 5
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
for nb in (2048, 256):
    gb = synth.config3_batch(nb); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H); m.load_state_dict({k: tt(v) for k, v in synth.gat_seq_params(D, D, D, D, K, H, seed=777).items()}); m = m.to(dev).eval()
    x, ea, ins = tt(synth.normal((N, D), 1)).to(dev), tt(synth.normal((E, D), 2)).to(dev), tt(synth.normal((K, B, D), 3)).to(dev)
    ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
    g = SceneGraphBatch(ei, batch, N, B, host_layout=HostLayout.from_numpy(gb.edge_index, gb.batch, B))
    for _ in range(3): ref = m(x, ei, ea, ins, batch, graph=g)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): m(x, ei, ea, ins, batch, graph=g)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(cg):
        out = m(x, ei, ea, ins, batch, graph=g)
    cg.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(out, ref))
    def timeit(f, n=30):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"graphs": nb, "replay_equals_eager": same, "eager_forward_ms": timeit(lambda: m(x, ei, ea, ins, batch, graph=g)), "graph_replay_ms": timeit(cg.replay)}))
