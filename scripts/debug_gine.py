import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth
from graphvqa_amd.baseline_models import GINEConv
from graphvqa_amd.graph import SceneGraphBatch
from oracle import ref_torch as R
from torch.nn import Linear, ReLU, Sequential
dev = torch.device("cuda:0")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
gb = synth.config2_batch(); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
dn, C, di = 300, 300, int(os.environ.get("DI", "512"))
rng = np.random.default_rng(1)
x, ea, ins = synth.normal((N, dn), 1), synth.normal((E, dn), 2), synth.normal((B, max(di,1)), 3)[:, :di]
w = lambda *s: (rng.standard_normal(s) / np.sqrt(s[-1])).astype(np.float32)
p = {"nn.0.weight": w(C, dn + di), "nn.0.bias": w(C), "nn.2.weight": w(C, C), "nn.2.bias": w(C)}
conv = GINEConv(Sequential(Linear(dn + di, C), ReLU(), Linear(C, C))).to(dev).eval()
conv.load_state_dict({**{k: tt(v) for k, v in p.items()}, "eps": torch.tensor([0.0])})
g = SceneGraphBatch(tt(gb.edge_index).to(dev), tt(gb.batch).to(dev), N, B)
for rep in range(3):
    out = conv(tt(x).to(dev), tt(gb.edge_index).to(dev), tt(ea).to(dev), graph=g, ins=tt(ins).to(dev) if di else None).cpu()
    xc = np.concatenate([x, ins[gb.batch]], 1) if di else x
    ec = np.concatenate([ea, ins[gb.batch[gb.edge_index[0]]]], 1) if di else ea
    ref = R.gine_conv(tt(xc).double(), tt(gb.edge_index), tt(ec).double(), {k: tt(v).double() for k, v in p.items()}).float()
    err = (out - ref).abs()
    bad = (err > 1e-4).nonzero()
    print("rep", rep, "max", float(err.max()), "bad elems", len(bad), "of", err.numel())
    if len(bad):
        r, c = bad[:, 0].numpy(), bad[:, 1].numpy()
        print(" bad rows", len(np.unique(r)), "row%128 hist(>>5)", np.bincount((r % 128) >> 5, minlength=4), "rows min/max", r.min(), r.max())
        print(" bad cols", len(np.unique(c)), "col>>5 hist", np.bincount(c >> 5, minlength=10), " (c%32)>>2 hist", np.bincount((c % 32) >> 2, minlength=8))
        print(" row blocks (row//128) first", np.unique(r // 128)[:20], "count", len(np.unique(r // 128)))
        rr = r[0]; print(" example row", rr, "bad cols", c[r == rr][:20], "err", err[rr, c[r == rr][:5]].numpy(), "ref", ref[rr, c[r == rr][:5]].numpy())
from graphvqa_amd.baseline_models import gine_seq
p = synth.gine_seq_params(300, 300, 512, 404)
m = gine_seq(300, 300, 512); m.load_state_dict({k: tt(v) for k, v in p.items()}); m = m.to(dev).eval()
ins5 = synth.normal((5, B, 512), 3)
args = [tt(a).to(dev) for a in (x, gb.edge_index, ea, ins5, gb.batch)]
out, convs = m(*args, return_convs=True)
ref_out, ref_convs = R.gine_seq(tt(x), tt(gb.edge_index), tt(ea), tt(ins5), tt(gb.batch), {k: tt(v) for k, v in p.items()}, return_convs=True)
for i, (a, b) in enumerate(zip(convs, ref_convs)):
    err = (a.cpu() - b).abs(); bad = (err > 1e-4).nonzero()
    print("layer", i, "max", float(err.max()), "bad", len(bad), "ref max", float(b.abs().max()))
    if len(bad):
        r, c = bad[:, 0].numpy(), bad[:, 1].numpy()
        print("  bad rows", len(np.unique(r)), "row%128>>5", np.bincount((r % 128) >> 5, minlength=4), "cols", len(np.unique(c)), "col>>5", np.bincount(c >> 5, minlength=10))
        rr = r[0]; print("  example row", rr, "n bad cols", (r == rr).sum(), "err", err[rr, c[r == rr][:4]].numpy(), "ref", b[rr, c[r == rr][:4]].numpy())
