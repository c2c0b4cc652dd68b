#!/usr/bin/env python3
"""The weight-gradient product dW = dy^T x alone and the one-call projection backward, at config-3 (and config-2) sizes, with the product
reading the row-major operands itself (GVQA_OPT_TN_DIRECT = 1, tn_direct.hip) and with transposed packs in HBM first (0).  One JSON line each."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphvqa_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream

def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6

for (R, M, K) in ((65536, 2048, 512), (30000, 1200, 300)):
    dy, x, W = torch.randn(R, M, device=dev), torch.randn(R, K, device=dev), torch.randn(M, K, device=dev)
    dW, dx = torch.empty(M, K, device=dev), torch.empty(R, K, device=dev)
    am = dy.abs().max().reshape(1)
    ws = torch.empty(lib.gvqa_linear_backward_workspace_bytes(R, M, K), dtype=torch.uint8, device=dev)
    ref = (dy.double().t() @ x.double())
    for direct in (1, 0):
        old = _lib.set_option(_lib.OPT_TN_DIRECT, direct)
        try:
            tn = lambda: _lib.check(lib.gvqa_linear_tn_split2h(R, M, K, dy.data_ptr(), M, x.data_ptr(), K, am.data_ptr(), 1, None, 0, dW.data_ptr(), K,
                                                               ws.data_ptr(), ws.numel(), st()))
            both = lambda: _lib.check(lib.gvqa_linear_backward_split2h(R, M, K, dy.data_ptr(), M, W.data_ptr(), K, x.data_ptr(), K, am.data_ptr(), 1,
                                                                       dx.data_ptr(), K, 0, dW.data_ptr(), K, ws.data_ptr(), ws.numel(), st()))
            t_tn = timed(tn)
            err = float((dW.double() - ref).abs().max() / ref.abs().max())
            t_both = timed(both)
            print(json.dumps({"R": R, "M": M, "K": K, "direct": direct, "tn_us": round(t_tn, 1), "backward_both_us": round(t_both, 1), "dW_rel_err": err}))
        finally:
            _lib.set_option(_lib.OPT_TN_DIRECT, old)
