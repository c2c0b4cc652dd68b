#!/usr/bin/env python3
"""Plain two-piece product from packed operands: k_linear_split3 (GVQA_PK_DIRECT=0) against k_linear_pk_direct (=1; tn_direct.hip) -- run once per setting
(the switch is read once per process): time per launch and max error against fp64 on a row sample."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphvqa_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
for (M, N, K, relu, add) in ((65536, 2048, 512, 0, 0), (65536, 512, 2048, 0, 1), (29785, 512, 512, 1, 0), (29785, 1536, 1024, 0, 0), (29785, 512, 300, 2, 1),
                             (262144, 300, 300, 1, 0), (5000, 132, 100, 0, 1)):
    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn((M, K), generator=g).to(dev), torch.randn((N, K), generator=g).to(dev)
    A[::5] *= 37.0
    bias = torch.randn(N, generator=g).to(dev); addend = torch.randn((M, N), generator=g).to(dev) if add else None
    apk = torch.empty(lib.gvqa_split2h_packed_bytes(M, K), dtype=torch.uint8, device=dev); wpk = torch.empty(lib.gvqa_split2h_packed_bytes(N, K), dtype=torch.uint8, device=dev)
    _lib.check(lib.gvqa_split2h_pack(M, K, A.data_ptr(), K, apk.data_ptr(), st())); _lib.check(lib.gvqa_split2h_pack(N, K, W.data_ptr(), K, wpk.data_ptr(), st()))
    C = torch.empty((M, N), device=dev)
    run = lambda: _lib.check(lib.gvqa_linear_split2h(M, N, K, apk.data_ptr(), wpk.data_ptr(), bias.data_ptr(), None if addend is None else addend.data_ptr(), N, None, 0, relu,
                                                     C.data_ptr(), N, st()))
    us = timed(run)
    rows = torch.arange(0, M, max(M // 512, 1), device=dev)
    ref = A[rows].double() @ W.double().t() + bias.double()
    if addend is not None: ref = ref + addend[rows].double()
    if relu == 1: ref = ref.clamp_min(0)
    elif relu == 2: ref = torch.where(ref > 0, ref, torch.expm1(ref))
    err = float((C[rows].double() - ref).abs().max() / ref.abs().max())
    print(json.dumps({"M": M, "N": N, "K": K, "pk_direct": os.environ.get("GVQA_PK_DIRECT", "0"), "us": round(us, 1), "TF_issued": round(6.0 * M * N * K / us / 1e6, 1), "rel_err": err}))
