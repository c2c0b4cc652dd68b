#!/usr/bin/env python3
"""Training step of gat_seq at BASELINE config 3 (64k nodes / 256k edges, d=512, K=5; CONFIG=2: config 2, d=300): forward + backward through the
differentiable path (HIP message passing + HIP backward, torch GEMMs / BatchNorm / dropout), and the two backward
kernels alone.  Prints one JSON object."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq, gat_message_passing
from graphvqa_amd.graph import SceneGraphBatch

dev = torch.device("cuda:0"); tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
H, K = 4, 5
if os.environ.get("CONFIG") == "2":        # BASELINE config 2: d = 300, instruction vectors 512, 1000 graphs of ~30 nodes
    D, DI = 300, 512
    gb = synth.config2_batch()
else:
    D, DI = 512, 512
    gb = synth.config3_batch()
N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
m = gat_seq(D, D, D, DI, K, dropout=0.1, gat_heads=H)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.gat_seq_params(D, D, D, DI, K, H, seed=777).items()})
m = m.to(dev).train()
x, ea, ins = tt(synth.normal((N, D), 1)), tt(synth.normal((E, D), 2)), tt(synth.normal((K, B, DI), 3))
ei, batch = tt(gb.edge_index), tt(gb.batch)
g = SceneGraphBatch(ei, batch, N, B); g.transposed()
opt = torch.optim.SGD(m.parameters(), lr=1e-3)

def step():
    opt.zero_grad(set_to_none=True)
    out = m(x, ei, ea, ins, batch, graph=g)
    out.square().mean().backward()
    opt.step()

def timed(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters

res = {"N": N, "E": E, "train_step_ms": timed(step) * 1e3}
res["train_edges_per_s"] = E / (res["train_step_ms"] * 1e-3)
if os.environ.get("TRAIN_ONLY"):          # profiling runs: the training step alone
    print(json.dumps(res)); sys.exit(0)
def fwd_only():
    with torch.no_grad():
        m(x, ei, ea, ins, batch, graph=g)
res["train_mode_forward_only_ms"] = timed(fwd_only) * 1e3

# the message-passing op alone: forward, backward
xp = torch.randn(N, H * D, device=dev, requires_grad=True); an = torch.randn(N, 2 * H, device=dev, requires_grad=True)
ae = torch.randn(E, H, device=dev, requires_grad=True); w = torch.randn(N, D, device=dev)
out, _ = gat_message_passing(xp, an, ae, g, H, D)
loss = (out * w).sum()
def bwd():
    xp.grad = an.grad = ae.grad = None
    loss.backward(retain_graph=True)
res["mp_forward_us"] = timed(lambda: gat_message_passing(xp, an, ae, g, H, D), 20) * 1e6
res["mp_backward_us"] = timed(bwd, 20) * 1e6
# algorithmic bytes of the backward: read xp, dout (twice: by destination and by source), alpha, logits; write dxp, da_*
alg = 4 * (N * H * D + 2 * N * D + N * H * D + 3 * E * H + 2 * N * 2 * H + 2 * (E + N + 1))
res["mp_backward_alg_bytes"] = alg
res["mp_backward_GBps"] = alg / (res["mp_backward_us"] * 1e-6) / 1e9
print(json.dumps(res))
