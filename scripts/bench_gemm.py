#!/usr/bin/env python3
"""Micro-benchmark of gvqa_linear_f32 (C = A . B^T, fp32 MFMA) at the path's shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphvqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
shapes = [(65536, 2048, 512, "cfg3 proj"), (29785, 1200, 300, "cfg2 proj"), (4096, 4096, 4096, "4096^3"),
          (262144, 20, 512, "cfg3 edge logits"), (65536, 8, 512, "cfg3 node logits")]
st = torch.cuda.current_stream().cuda_stream
for M, N, K, name in shapes:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    def run():
        _lib.check(lib.gvqa_linear_f32(M, N, K, A.data_ptr(), K, B.data_ptr(), K, None, 0, C.data_ptr(), N, st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref = A[:64] @ B.T
    err = float((C[:64] - ref).abs().max())
    print(f"{name:18s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s  {(M*K+N*K+M*N)*4/ms/1e6:7.0f} GB/s  err {err:.2e}", flush=True)
