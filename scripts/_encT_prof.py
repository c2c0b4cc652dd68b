import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphvqa_amd import synth
from graphvqa_amd.graph import SceneGraphBatch
from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
dev = torch.device("cuda:0"); tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
gb = synth.config3_batch(); N, E, B, V, D = gb.num_nodes, gb.num_edges, gb.num_graphs, 3000, 300
enc = GroundTruth_SceneGraph_Encoder(V, 0, D); enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_params(V, D, seed=1).items()}); enc = enc.to(dev).train()
data = types.SimpleNamespace(x=tt(synth.randint(N * 12, 71, 0, V, stream=9).reshape(N, 12)), edge_attr=tt(synth.randint(E, 72, 1, V, stream=9).reshape(E, 1)),
                             edge_index=tt(gb.edge_index), batch=tt(gb.batch), added_sym_edge=tt(np.arange(0, E, 7, dtype=np.int64)))
g = SceneGraphBatch(data.edge_index, data.batch, N, B); g.transposed()
for _ in range(6):
    for p in enc.parameters(): p.grad = None
    xe, ee, _ = enc(data, graph=g)
    (xe.square().mean() + ee.square().mean()).backward()
torch.cuda.synchronize()
