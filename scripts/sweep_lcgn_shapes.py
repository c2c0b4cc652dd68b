#!/usr/bin/env python3
"""The node products of lcgn_seq at the config-5 shape (M = 29785 rows) on every two-piece instantiation of the split GEMM: us per launch
and issued PF/s per (shape, variant), and what the default rule picks.  python scripts/sweep_lcgn_shapes.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphvqa_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream


def timeit(f, n=20, w=5):
    for _ in range(w): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


M = int(os.environ.get("M", "29785"))
for (N, K, name) in [(512, 512, "proj_x_ctx / proj_x_loc / fin"), (1536, 1024, "J"), (512, 1024, "output_layer"), (1536, 512, "XL"), (512, 300, "init")]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev)
    apk = torch.empty(lib.gvqa_split2h_packed_bytes(M, K), dtype=torch.uint8, device=dev)
    wpk = torch.empty(lib.gvqa_split2h_packed_bytes(N, K), dtype=torch.uint8, device=dev)
    _lib.check(lib.gvqa_split2h_pack(M, K, A.data_ptr(), K, apk.data_ptr(), st))
    _lib.check(lib.gvqa_split2h_pack(N, K, W.data_ptr(), K, wpk.data_ptr(), st))
    gemm = lambda: _lib.check(lib.gvqa_linear_split2h(M, N, K, apk.data_ptr(), wpk.data_ptr(), None, None, 0, None, 0, 0, C.data_ptr(), N, st))
    pack = lambda: _lib.check(lib.gvqa_split2h_pack(M, K, A.data_ptr(), K, apk.data_ptr(), st))
    row = {"product": name, "M": M, "N": N, "K": K}
    for var in (0, 112, 118, 114, 134, 124):
        if var == 112 and ((K + 15) // 16) % 2: continue
        _lib.set_option(_lib.OPT_SPLIT3_VARIANT, var)
        try:
            t = timeit(gemm)
            row["auto" if var == 0 else f"v{var}"] = [round(t * 1e6, 1), round(3 * 2.0 * M * N * K / t / 1e15, 3)]
        except Exception as e:
            row[f"v{var}"] = str(e)[:60]
    _lib.set_option(_lib.OPT_SPLIT3_VARIANT, 0)
    row["pack_a_us"] = round(timeit(pack) * 1e6, 1)
    print(json.dumps(row), flush=True)
