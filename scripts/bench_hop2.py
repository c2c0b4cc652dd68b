"""A/B of the hop kernels on the config-3 batch inside one process, interleaved rounds: GVQA_OPT_HOP_FUSION 1 (8-wave fused
kernel, one workgroup per CU) vs 2 (hop2.hip: persistent 4-wave workgroups, two per CU).  Prints the hop kernel's
average launch duration (library stage timers, HIP events on the stream) and the wall time of a whole forward, and the
max-abs difference of the two outputs."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
D = int(os.environ.get("D", "512")); H = int(os.environ.get("H", "4")); K = 5
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
nb = int(os.environ.get("NB", "2048"))
gb = synth.config3_batch(nb) if D == 512 else synth.config2_batch()
N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
Di = 512
m = gat_seq(D, D, D, Di, K, dropout=0.1, gat_heads=H)
m.load_state_dict({k: tt(v) for k, v in synth.gat_seq_params(D, D, D, Di, K, H, seed=777).items()}); m = m.to(dev).eval()
x, ea, ins = tt(synth.normal((N, D), 1)).to(dev), tt(synth.normal((E, D), 2)).to(dev), tt(synth.normal((K, B, Di), 3)).to(dev)
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
if os.environ.get("ZERO"):        # DVFS probe: all-zero operands (same instruction stream, far less switching energy)
    x.zero_()
    for p_ in m.parameters(): p_.data.zero_()
g = SceneGraphBatch(ei, batch, N, B)
print(json.dumps({"hop2_blocks_per_cu": _lib.load().gvqa_hop2_blocks_per_cu(H), "N": N, "E": E, "row_groups": g.c.num_row_groups,
                  "max_row_group_edges": g.c.max_row_group_edges}), flush=True)
outs = {}
modes = [int(v) for v in os.environ.get("MODES", "1,2").split(",")]
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for mode in modes:
        _lib.set_option(_lib.OPT_HOP_FUSION, mode)
        for _ in range(3): out = m(x, ei, ea, ins, batch, graph=g)
        torch.cuda.synchronize()
        outs[mode] = out.clone()
        t0 = time.perf_counter()
        for _ in range(10): m(x, ei, ea, ins, batch, graph=g)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
        _lib.prof_enable(True); _lib.prof_collect()
        for _ in range(10): m(x, ei, ea, ins, batch, graph=g)
        torch.cuda.synchronize(); p = _lib.prof_collect(); _lib.prof_enable(False)
        print(json.dumps({"round": rnd, "hop_fusion": mode, "hop_kernel_us": round(p["proj"][0] / p["proj"][1] * 1e3, 1),
                          "forward_wall_ms": round(wall, 3),
                          "stages_us_per_step": {k: round(v[0] * 100, 1) for k, v in p.items() if v[1]}}), flush=True)
if len(outs) == 2:
    a, b = [outs[k] for k in sorted(outs)]
    print(json.dumps({"max_abs_diff_between_modes": float((a - b).abs().max()), "out_absmax": float(a.abs().max())}))
