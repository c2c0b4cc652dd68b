"""One two-piece bf16 product, timed with HIP events over 50 back-to-back launches: python scripts/bench_gemm_bf16_one.py M N K [zeros]"""
import json, os, sys, torch
sys.path.insert(0, ".")
from graphvqa_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
M, N, K = (int(v) for v in sys.argv[1:4]); zeros = len(sys.argv) > 4
A = (torch.zeros(M, K, device=dev) if zeros else torch.randn(M, K, device=dev)).bfloat16(); W = torch.zeros(N, K, device=dev) if zeros else torch.randn(N, K, device=dev)
Wpk = torch.empty(N, 2 * K, dtype=torch.bfloat16, device=dev)
_lib.check(lib.gvqa_pack_weight_bf16(N, K, 2, W.data_ptr(), K, Wpk.data_ptr(), st))
C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
f = lambda: _lib.check(lib.gvqa_linear_bf16(M, N, K, 2, A.data_ptr(), K, Wpk.data_ptr(), None, None, 0, None, 0, 0, C.data_ptr(), N, 1, st))
for _ in range(20): f()
best = None
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    best = us if best is None else min(best, us)
print(json.dumps({"lib": os.environ.get("GVQA_LIB", "product"), "bf16_gemm": os.environ.get("GVQA_BF16_GEMM", "default"), "M": M, "N": N, "K": K, "zeros": zeros, "us": round(best, 2),
                  "issued_pflops": round(4.0 * M * N * K / best / 1e9, 4)}))
