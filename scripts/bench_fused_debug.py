"""Ablation of the fused hop kernel's epilogue (GVQA_FUSED_DEBUG bit mask, read per launch) on the config-3 batch."""
import os as _os
_os.environ.setdefault("GVQA_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "graphvqa_amd", "lib", "probes", "libgvqa_hip.so"))   # the measurement build (python -m graphvqa_amd.build --probes)
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch
torch.set_grad_enabled(False)
dev = torch.device("cuda:0"); D, H, K = 512, 4, 5
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
gb = synth.config3_batch(); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H); m.load_state_dict({k: tt(v) for k, v in synth.gat_seq_params(D, D, D, D, K, H, seed=777).items()}); m = m.to(dev).eval()
x, ea, ins = tt(synth.normal((N, D), 1)).to(dev), tt(synth.normal((E, D), 2)).to(dev), tt(synth.normal((K, B, D), 3)).to(dev)
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
g = SceneGraphBatch(ei, batch, N, B)
for dbg in os.environ.get("DBG", "0,8,1,2,4,3,7").split(","):
    os.environ["GVQA_FUSED_DEBUG"] = dbg
    for _ in range(3): m(x, ei, ea, ins, batch, graph=g)
    _lib.prof_enable(True); _lib.prof_collect()
    for _ in range(10): m(x, ei, ea, ins, batch, graph=g)
    torch.cuda.synchronize(); p = _lib.prof_collect(); _lib.prof_enable(False)
    print(json.dumps({"debug": int(dbg), "fused_kernel_us": round(p["proj"][0] / p["proj"][1] * 1e3, 1)}), flush=True)
