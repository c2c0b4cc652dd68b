"""Operand packing passes alone (plain rows, no logits): split3 vs split2h on the config-3 node matrix."""
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
    return e0.elapsed_time(e1) / n * 1e3
for (M, K) in [(65536, 512), (65536, 1024), (29785, 300)]:
    A = torch.relu(torch.randn(M, K, device=dev))
    B = torch.empty_like(A)
    row = {"M": M, "K": K, "copy_us": round(timeit(lambda: B.copy_(A)), 1)}
    for name, nbytes, pack in (("split3", lib.gvqa_split3_packed_bytes, lib.gvqa_split3_pack), ("split2h", lib.gvqa_split2h_packed_bytes, lib.gvqa_split2h_pack)):
        buf = torch.empty(nbytes(M, K), dtype=torch.uint8, device=dev)
        t = timeit(lambda: _lib.check(pack(M, K, A.data_ptr(), K, buf.data_ptr(), st)))
        row[name + "_us"] = round(t, 1); row[name + "_GBps"] = round((A.numel() * 4 + buf.numel()) / t / 1e3, 0)
    print(json.dumps(row), flush=True)
