#!/usr/bin/env python3
"""Secondary measurements for the BASELINE configs that are not the headline bench line:
config 2 (GAT, d=300, 1k graphs), config 4 (GINEConv x5 on the config-2 batch), config 5 shape (LCGN, fp32).
Prints one JSON object; run on the GPU box:  python scripts/bench_configs.py > gpurun_out/configs.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.set_grad_enabled(False)      # inference measurements: fused path

from graphvqa_amd import synth, _lib
from graphvqa_amd.graph import SceneGraphBatch, HostLayout

tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
dev = torch.device("cuda:0")


def timed(fn, warmup=12, steps=20):
    """Wall time per call WITHOUT the in-library stage timers (their HIP events cost a launch-bound forward 5-10 %), then the stage
    breakdown from a second pass with them on."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    _lib.prof_enable(True); _lib.prof_collect()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    prof = _lib.prof_collect(); _lib.prof_enable(False)
    return dt, {k: (v[0] / steps, v[1] // steps) for k, v in prof.items() if v[1]}


def load(m, p):
    m.load_state_dict({k: tt(v) for k, v in p.items()})
    return m.to(dev).eval()


res = {}
gb = synth.config2_batch()
N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
x, ea = tt(synth.normal((N, 300), 1)).to(dev), tt(synth.normal((E, 300), 2)).to(dev)
ins = tt(synth.normal((5, B, 512), 3)).to(dev)

from graphvqa_amd.gat_skip import gat_seq
m = load(gat_seq(300, 300, 300, 512, 5, dropout=0.1, gat_heads=4), synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=303))
def gat_modes(run, Nn, Ee, C=300, H=4):
    """Default path (fused hops) and the unfused one, whose stand-alone message-passing kernel is priced against the HBM roofline
    with SURVEY 8(d)'s algorithmic bytes (+ 4 N C: the skip rows read by the fused epilogue)."""
    out = {}
    dt, prof = timed(run)
    out["fused"] = {"ms_per_forward": dt * 1e3, "edges_per_s": Ee / dt, "stage_ms": {k: v[0] for k, v in prof.items()}}
    old = _lib.set_option(_lib.OPT_HOP_FUSION, 0)
    try:
        dt, prof = timed(run)
    finally:
        _lib.set_option(_lib.OPT_HOP_FUSION, old)
    mp_ms, mp_n = prof["mp"]
    alg = 4 * (Nn * H * C + 2 * Nn * H + Ee * H + Ee + (Nn + 1) + Nn * C) + 4 * Nn * C
    out["unfused"] = {"ms_per_forward": dt * 1e3, "edges_per_s": Ee / dt, "mp_us_per_hop": mp_ms / mp_n * 1e3, "mp_alg_bytes": alg,
                      "mp_GBps": alg / (mp_ms / mp_n * 1e-3) / 1e9, "mp_frac_of_8TBps": alg / (mp_ms / mp_n * 1e-3) / 8e12,
                      "stage_ms": {k: v[0] for k, v in prof.items()}}
    return out


# a forward = device CSR build from COO + gat_seq, with the loader-side per-graph layout (no device read-back), as in bench.py
hl2 = HostLayout.from_numpy(gb.edge_index, gb.batch, B)
res["config2_gat_d300"] = dict(N=N, E=E, B=B, **gat_modes(lambda: m(x, ei, ea, ins, batch, graph=SceneGraphBatch(ei, batch, N, B, host_layout=hl2)), N, E))

# "GQA-shaped" alternative of SURVEY 8(d): E/N ~ 4 (one self loop + ~3 relations per node), same dims
gb4 = synth.make_graph_batch(1000, seed=0x5EED0002, nodes_lo=20, nodes_hi=40, rel_per_node=3.0)
N4, E4 = gb4.num_nodes, gb4.num_edges
ei4, batch4 = tt(gb4.edge_index).to(dev), tt(gb4.batch).to(dev)
x4, ea4 = tt(synth.normal((N4, 300), 1)).to(dev), tt(synth.normal((E4, 300), 2)).to(dev)
hl4 = HostLayout.from_numpy(gb4.edge_index, gb4.batch, B)
res["config2_gat_d300_EoverN4"] = dict(N=N4, E=E4, **gat_modes(lambda: m(x4, ei4, ea4, ins, batch4, graph=SceneGraphBatch(ei4, batch4, N4, B, host_layout=hl4)), N4, E4))

# config 3 (the headline batch): the stand-alone message-passing kernel's roofline next to config 2's
gb3 = synth.config3_batch()
N3, E3, B3 = gb3.num_nodes, gb3.num_edges, gb3.num_graphs
m3 = load(gat_seq(512, 512, 512, 512, 5, dropout=0.1, gat_heads=4), synth.gat_seq_params(512, 512, 512, 512, 5, 4, seed=777))
a3 = [tt(v).to(dev) for v in (synth.normal((N3, 512), 1), gb3.edge_index, synth.normal((E3, 512), 2), synth.normal((5, B3, 512), 3), gb3.batch)]
hl3 = HostLayout.from_numpy(gb3.edge_index, gb3.batch, B3)
res["config3_gat_d512"] = dict(N=N3, E=E3, B=B3, **gat_modes(lambda: m3(*a3, graph=SceneGraphBatch(a3[1], a3[4], N3, B3, host_layout=hl3)), N3, E3, C=512))
del m3, a3

from graphvqa_amd.baseline_models import gine_seq, gcn_seq
m = load(gine_seq(300, 300, 512), synth.gine_seq_params(300, 300, 512, 404))
g = SceneGraphBatch(ei, batch, N, B)
dt, prof = timed(lambda: m(x, ei, ea, ins, batch, graph=g, return_convs=True))
agg_ms, agg_n = prof["mp"]
alg = 4 * (N * 300 + E * 300 + E + (N + 1) + N * 300)
res["config4_gine_convs"] = {"ms_per_5_convs_plus_module": dt * 1e3, "edges_per_s": E / dt,
                             "aggregate_us_per_layer": agg_ms / agg_n * 1e3, "aggregate_alg_bytes": alg,
                             "aggregate_GBps": alg / (agg_ms / agg_n * 1e-3) / 1e9,
                             "stage_ms": {k: v[0] for k, v in prof.items()}}
dt, _ = timed(lambda: m(x, ei, ea, ins, batch))
res["config4_gine_module_as_written"] = {"ms": dt * 1e3}
m = load(gcn_seq(300, 300, 512), synth.gcn_seq_params(300, 300, 512, 505))
dt, prof = timed(lambda: m(x, ei, ins, batch, graph=g, return_convs=True))
agg_ms, agg_n = prof["mp"]
alg = 8 * N * 300 + 4 * E
res["gcn_convs"] = {"ms_per_5_convs_plus_module": dt * 1e3, "aggregate_us_per_layer": agg_ms / agg_n * 1e3, "aggregate_alg_bytes": alg,
                    "aggregate_GBps": alg / (agg_ms / agg_n * 1e-3) / 1e9, "stage_ms": {k: v[0] for k, v in prof.items()}}

from graphvqa_amd.lcgn import lcgn_seq
m = load(lcgn_seq(300, 512, 300, 5), synth.lcgn_seq_params(300, 512, seed=808))
q, lstm = tt(synth.normal((B, 512), 5)).to(dev), tt(synth.normal((10, B, 512), 6)).to(dev)
xc = tt(synth.normal((N, 512), 7)).to(dev)
dt, prof = timed(lambda: m(x, ei, batch, q, lstm, graph=g, x_ctx_init=xc))
res["config5_lcgn_fp32"] = {"ms_per_forward": dt * 1e3, "edges_per_s": E / dt, "stage_ms": {k: v[0] for k, v in prof.items()}}
m16 = load(lcgn_seq(300, 512, 300, 5, node_feature_dtype=torch.bfloat16), synth.lcgn_seq_params(300, 512, seed=808))
dt, prof = timed(lambda: m16(x, ei, batch, q, lstm, graph=g, x_ctx_init=xc))
res["config5_lcgn_bf16_node_features"] = {"ms_per_forward": dt * 1e3, "edges_per_s": E / dt,
                                          "max_abs_vs_fp32_mode": float((m16(x, ei, batch, q, lstm, graph=g, x_ctx_init=xc) -
                                                                         m(x, ei, batch, q, lstm, graph=g, x_ctx_init=xc)).abs().max())}
print(json.dumps(res))
