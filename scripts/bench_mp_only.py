#!/usr/bin/env python3
"""Time gvqa_gat_message_passing alone (config-3 shapes) for a given build of the library:
    python scripts/bench_mp_only.py [path/to/libgvqa_hip.so ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphvqa_amd import synth, _lib
dev = torch.device("cuda:0")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
gb = synth.config3_batch()
N, E, B, Cc, H = gb.num_nodes, gb.num_edges, gb.num_graphs, 512, 4
ei, batch = tt(gb.edge_index), tt(gb.batch)
xp, a_node, a_edge = torch.randn(N, H * Cc, device=dev), torch.randn(N, 2 * H, device=dev), torch.randn(E, 5 * H, device=dev)
T, skip = torch.randn(B, Cc + H, device=dev), torch.randn(N, Cc, device=dev)
vec = [torch.rand(Cc, device=dev) + 0.5 for _ in range(5)]
out = torch.empty(N, Cc, device=dev)
ws = torch.empty(E * H * 4 + 8 * Cc + 256, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for path in (sys.argv[1:] or [_lib.LIB_PATH]):
    lib = C.CDLL(path)
    lib.gvqa_graph_workspace_bytes.restype = C.c_size_t
    lib.gvqa_graph_workspace_bytes.argtypes = [C.c_int64] * 3
    g = _lib.Graph()
    gws = torch.empty(lib.gvqa_graph_workspace_bytes(N, E, B), dtype=torch.uint8, device=dev)
    lib.gvqa_graph_build.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(_lib.Graph)]
    lib.gvqa_graph_finalize.argtypes = [C.POINTER(_lib.Graph), C.c_void_p]
    lib.gvqa_gat_message_passing.argtypes = [C.POINTER(_lib.Graph), C.POINTER(_lib.GatMpDesc), C.c_void_p, C.c_size_t, C.c_void_p]
    assert lib.gvqa_graph_build(N, E, B, ei.data_ptr(), batch.data_ptr(), gws.data_ptr(), gws.numel(), st, C.byref(g)) == 0
    assert lib.gvqa_graph_finalize(C.byref(g), st) == 0
    d = _lib.GatMpDesc()
    d.C, d.H, d.negative_slope, d.bn_eps = Cc, H, 0.2, 1e-5
    d.xp, d.a_node, d.a_edge, d.a_edge_stride = xp.data_ptr(), a_node.data_ptr(), a_edge.data_ptr(), 5 * H
    d.graph_term, d.graph_term_ld, d.skip = T.data_ptr(), Cc + H, skip.data_ptr()
    d.bias, d.bn_weight, d.bn_bias, d.bn_mean, d.bn_var = [v.data_ptr() for v in vec]
    d.out = out.data_ptr()
    run = lambda: lib.gvqa_gat_message_passing(C.byref(g), C.byref(d), ws.data_ptr(), ws.numel(), st)
    for _ in range(5): assert run() == 0
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{os.path.basename(path):28s} MP kernel: min {min(ts):6.1f} us  median {sorted(ts)[2]:6.1f} us  (checksum {float(out.sum()):.3f})", flush=True)
