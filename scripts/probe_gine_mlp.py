#!/usr/bin/env python3
"""Phase stamps of the fused GINE MLP kernel (measurement variant: python -m graphvqa_amd.build --variant gm_stamps GVQA_GM_STAMPS=1;
GVQA_LIB=graphvqa_amd/lib/gm_stamps/libgvqa_hip.so python scripts/probe_gine_mlp.py).  The variant overwrites 12 floats of every workgroup's first
output row with six 100 MHz stamps of wave 0: begin | rings primed | layer 1 done | between-layers done | layer 2 done | stores issued."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth
from graphvqa_amd.graph import SceneGraphBatch
from graphvqa_amd.baseline_models import gine_seq
dev = torch.device("cuda:0")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
gb = synth.config2_batch(); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
x, ea, ins = tt(synth.normal((N, 300), 1)).to(dev), tt(synth.normal((E, 300), 2)).to(dev), tt(synth.normal((5, B, 512), 3)).to(dev)
m = gine_seq(300, 300, 512); m.load_state_dict({k: tt(v) for k, v in synth.gine_seq_params(300, 300, 512, 404).items()}); m = m.to(dev).eval()
g = SceneGraphBatch(ei, batch, N, B)
conv = m.convs[0]
for _ in range(5): out = conv(x, ei, ea, graph=g, ins=ins[0].contiguous())
torch.cuda.synchronize()
G = (N + 127) // 128
raw = np.ascontiguousarray(out.cpu().numpy()).reshape(-1).view(np.uint8)
st = np.stack([raw[128 * w * 300 * 4: 128 * w * 300 * 4 + 48].copy().view(np.uint64) for w in range(G)]).astype(np.float64) / 100.0
names = ["prologue (prime rings + rows)", "layer 1 (19 steps)", "between layers", "layer 2 (19 steps)", "drain", ]
d = np.diff(st, axis=1)
print(json.dumps({"workgroups": G, "span_us": float(st.max() - st.min()), "start_spread_us": float(st[:, 0].max() - st[:, 0].min()),
                  **{names[k]: [round(float(d[:, k].mean()), 2), round(float(d[:, k].max()), 2)] for k in range(5)}, "total [mean, max]": [round(float((st[:, 5] - st[:, 0]).mean()), 2), round(float((st[:, 5] - st[:, 0]).max()), 2)]}))
