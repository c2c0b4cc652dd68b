#!/usr/bin/env python3
"""Tall-skinny logit products of the differentiable path (csrc/train.hip) alone, at config-3 node and edge sizes: launch time and
the HBM rate of the one stream each kernel makes over X (or dX).  Prints one JSON line per case."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphvqa_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
def timed(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for R, D, J in ((65536, 512, 8), (262144, 512, 20), (29785, 300, 8)):
    X = torch.randn(R, D, device=dev); V = torch.randn(D, J, device=dev); G = torch.randn(R, J, device=dev)
    Y = torch.empty(R, J, device=dev); dV = torch.empty(D, J, device=dev); dX = torch.empty(R, D, device=dev)
    ws = torch.empty(lib.gvqa_skinny_backward_weight_workspace_bytes(R, D, J), dtype=torch.uint8, device=dev)
    f = timed(lambda: lib.gvqa_skinny_forward(R, D, J, X.data_ptr(), D, V.data_ptr(), Y.data_ptr(), st))
    w = timed(lambda: lib.gvqa_skinny_backward_weight(R, D, J, X.data_ptr(), D, G.data_ptr(), dV.data_ptr(), ws.data_ptr(), ws.numel(), st))
    i = timed(lambda: lib.gvqa_skinny_backward_input(R, D, J, G.data_ptr(), V.data_ptr(), None, 0, dX.data_ptr(), D, st))
    t_f = timed(lambda: torch.mm(X, V)); t_w = timed(lambda: torch.mm(X.t(), G)); t_i = timed(lambda: torch.mm(G, V.t()))
    b = 4.0 * R * D
    print(json.dumps({"R": R, "D": D, "J": J, "forward_us": round(f, 1), "forward_TBps": round(b / f / 1e6, 2), "dV_us": round(w, 1),
                      "dV_TBps": round(b / w / 1e6, 2), "dX_us": round(i, 1), "dX_TBps": round(b / i / 1e6, 2),
                      "torch_mm_us": [round(t_f, 1), round(t_w, 1), round(t_i, 1)]}))

# the weight-gradient product dW = dy^T x (gvqa_linear_tn_split2h) against torch's fp32 matmul
for R, M, N in ((65536, 2048, 512), (29785, 1200, 300)):
    X = torch.randn(R, M, device=dev); Y = torch.randn(R, N, device=dev); Cc = torch.empty(M, N, device=dev)
    ws = torch.empty(lib.gvqa_linear_tn_workspace_bytes(R, M, N), dtype=torch.uint8, device=dev)
    mx = torch.tensor([float(X.abs().max()), float(Y.abs().max())], device=dev)
    t_all = timed(lambda: lib.gvqa_linear_tn_split2h(R, M, N, X.data_ptr(), M, Y.data_ptr(), N, None, 0, None, 0, Cc.data_ptr(), N, ws.data_ptr(), ws.numel(), st))
    t_known = timed(lambda: lib.gvqa_linear_tn_split2h(R, M, N, X.data_ptr(), M, Y.data_ptr(), N, mx.data_ptr(), 1, mx.data_ptr() + 4, 1, Cc.data_ptr(), N, ws.data_ptr(), ws.numel(), st))
    t_torch = timed(lambda: torch.mm(X.t(), Y))
    ref = X.double().t() @ Y.double()
    print(json.dumps({"tn_product": [R, M, N], "us": round(t_all, 1), "us_with_known_maxima": round(t_known, 1), "torch_mm_us": round(t_torch, 1),
                      "max_err_vs_fp64": float((Cc.double() - ref).abs().max()), "torch_max_err_vs_fp64": float((torch.mm(X.t(), Y).double() - ref).abs().max())}))
