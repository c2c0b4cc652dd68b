"""k_linear_bf16 on the LCGN config-5 shapes: effective TFLOP/s (2*M*N*K*pieces / time)."""
import json, sys, time, torch
sys.path.insert(0, ".")
from graphvqa_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
res = []
for (M, N, K) in [(29785, 1536, 1024), (29785, 512, 1024), (29785, 512, 512), (29785, 1536, 512), (65536, 2048, 512)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev)
    for P in (1, 2):
        Wpk = torch.empty(N, P * K, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.gvqa_pack_weight_bf16(N, K, P, W.data_ptr(), K, Wpk.data_ptr(), st))
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        f = lambda: _lib.check(lib.gvqa_linear_bf16(M, N, K, P, A.data_ptr(), K, Wpk.data_ptr(), None, None, 0, None, 0, 0, C.data_ptr(), N, 1, st))
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        res.append({"M": M, "N": N, "K": K, "pieces": P, "us": dt * 1e6, "mfma_tflops": 2.0 * M * N * K * P / dt / 1e12,
                    "useful_tflops": 2.0 * M * N * K / dt / 1e12})
for r in res: print(json.dumps(r))
