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

# the chained forms the fp32 lcgn_seq forward actually launches (round 5): packed output (128 x 512 tile), two A segments, epilogue operands
print("--- chained forms (gvqa_linear_split2h_chain) ---", flush=True)
O = 512
def img(rows, K):
    X = torch.randn(rows, K, device=dev)
    buf = torch.empty(lib.gvqa_split2h_packed_bytes(rows, K), dtype=torch.uint8, device=dev)
    _lib.check(lib.gvqa_split2h_pack(rows, K, X.data_ptr(), K, buf.data_ptr(), st))
    return buf
u, v = img(M, O), img(M, O)
w1, w2, w3 = img(O, O), img(3 * O, 2 * O), img(O, 2 * O)
bias = torch.randn(3 * O, device=dev); mulv = torch.randn(M, O, device=dev); add3 = torch.randn(M, 3 * O, device=dev)
C1 = torch.empty(M, O, device=dev); C3 = torch.empty(M, 3 * O, device=dev)
pk = torch.empty(lib.gvqa_split2h_packed_bytes(M, O), dtype=torch.uint8, device=dev)
forms = {
    "prod: [N,512]x[512,512] bias+mul -> packed only": lambda: lib.gvqa_linear_split2h_chain(M, O, O, u.data_ptr(), 0, None, w1.data_ptr(), bias.data_ptr(), None, 0, mulv.data_ptr(), O, 0, None, O, pk.data_ptr(), st),
    "same -> fp32 rows (no packed output)": lambda: lib.gvqa_linear_split2h_chain(M, O, O, u.data_ptr(), 0, None, w1.data_ptr(), bias.data_ptr(), None, 0, mulv.data_ptr(), O, 0, C1.data_ptr(), O, None, st),
    "J: two segments [N,512|512]x[1536,1024] + addend -> fp32": lambda: lib.gvqa_linear_split2h_chain(M, 3 * O, O, u.data_ptr(), O, v.data_ptr(), w2.data_ptr(), None, add3.data_ptr(), 3 * O, None, 0, 0, C3.data_ptr(), 3 * O, None, st),
    "output: two segments [N,512|512]x[512,1024] bias -> packed only": lambda: lib.gvqa_linear_split2h_chain(M, O, O, u.data_ptr(), O, v.data_ptr(), w3.data_ptr(), bias.data_ptr(), None, 0, None, 0, 0, None, O, pk.data_ptr(), st),
    "same -> fp32 rows": lambda: lib.gvqa_linear_split2h_chain(M, O, O, u.data_ptr(), O, v.data_ptr(), w3.data_ptr(), bias.data_ptr(), None, 0, None, 0, 0, C1.data_ptr(), O, None, st),
}
for name, fn in forms.items():
    _lib.check(fn())
    t = timeit(lambda: _lib.check(fn()))
    K = 2 * O if "two segments" in name else O
    N = 3 * O if name.startswith("J") else O
    print(json.dumps({"form": name, "us": round(t * 1e6, 1), "issued_PFs": round(3 * 2.0 * M * N * K / t / 1e15, 3)}), flush=True)
uv = img(M, 2 * O)
more = {
    "J: ONE segment [N,1024]x[1536,1024] + addend -> fp32": lambda: lib.gvqa_linear_split2h_chain(M, 3 * O, 2 * O, uv.data_ptr(), 0, None, w2.data_ptr(), None, add3.data_ptr(), 3 * O, None, 0, 0, C3.data_ptr(), 3 * O, None, st),
    "J: ONE segment, no addend": lambda: lib.gvqa_linear_split2h_chain(M, 3 * O, 2 * O, uv.data_ptr(), 0, None, w2.data_ptr(), None, None, 0, None, 0, 0, C3.data_ptr(), 3 * O, None, st),
    "J: two segments, no addend": lambda: lib.gvqa_linear_split2h_chain(M, 3 * O, O, u.data_ptr(), O, v.data_ptr(), w2.data_ptr(), None, None, 0, None, 0, 0, C3.data_ptr(), 3 * O, None, st),
}
for name, fn in more.items():
    _lib.check(fn())
    t = timeit(lambda: _lib.check(fn()))
    print(json.dumps({"form": name, "us": round(t * 1e6, 1), "issued_PFs": round(3 * 2.0 * M * 3 * O * 2 * O / t / 1e15, 3)}), flush=True)
