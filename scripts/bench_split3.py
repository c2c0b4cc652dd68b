"""Split projections (pack + GEMM; variants < 100: three bf16 pieces, >= 100: two fp16 pieces) vs the f32-MFMA kernel on the
config-3 / config-2 hop projection shapes.
Usage: python scripts/bench_split3.py [out.json]   (GVQA_GEMM_BACKEND=rocblas adds the vendor number to the f32 column)"""
import os as _os
_os.environ.setdefault("GVQA_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "graphvqa_amd", "lib", "probes", "libgvqa_hip.so"))   # the measurement build (python -m graphvqa_amd.build --probes)
import json, os, sys, time, torch
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


res = []
VARIANTS = [int(v) for v in os.environ.get("S3_VARIANTS", "14,13,34,33,118,117,114,113,134,133").split(",")]
SHAPES = os.environ.get("S3_SHAPES", "config3,config2").split(",")
for (M, N, K, name) in [(65536, 2048, 512, "config3"), (29785, 1200, 300, "config2"), (16384, 2048, 512, "quarter")]:
    if name not in SHAPES: continue
    for data in ("randn", "relu"):
        A = torch.randn(M, K, device=dev)
        if data == "relu": A = torch.relu(A)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        C = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev)
        row = {"shape": name, "M": M, "N": N, "K": K, "data": data}
        Cs = {}
        for scheme, nbytes, pack, linear, products in (("split3", lib.gvqa_split3_packed_bytes, lib.gvqa_split3_pack, lib.gvqa_linear_split3, 6),
                                                       ("split2h", lib.gvqa_split2h_packed_bytes, lib.gvqa_split2h_pack, lib.gvqa_linear_split2h, 3)):
            apk = torch.empty(nbytes(M, K), dtype=torch.uint8, device=dev)
            wpk = torch.empty(nbytes(N, K), dtype=torch.uint8, device=dev)
            pack_a = lambda: _lib.check(pack(M, K, A.data_ptr(), K, apk.data_ptr(), st))
            pack_w = lambda: _lib.check(pack(N, K, W.data_ptr(), K, wpk.data_ptr(), st))
            pack_a(); pack_w()
            gemm = lambda: _lib.check(linear(M, N, K, apk.data_ptr(), wpk.data_ptr(), None, None, 0, None, 0, 0, C.data_ptr(), N, st))
            for var in VARIANTS:
                if (var >= 100) != (scheme == "split2h"): continue
                _lib.set_option(_lib.OPT_SPLIT3_VARIANT, var)
                t = timeit(gemm)
                row[f"v{var}_us"] = round(t * 1e6, 1)
                row[f"v{var}_mfma_tflops"] = round(products * 2.0 * M * N * K / t / 1e12, 1)
            _lib.set_option(_lib.OPT_SPLIT3_VARIANT, 0)
            row[f"{scheme}_auto_us"] = round(timeit(gemm) * 1e6, 1)
            row[f"{scheme}_pack_a_us"] = round(timeit(pack_a) * 1e6, 1)
            row[f"{scheme}_pack_w_us"] = round(timeit(pack_w) * 1e6, 1)
            Cs[scheme] = C.clone()
        vend = _lib.set_option(_lib.OPT_VENDOR_GEMM, 0)
        f32 = lambda: _lib.check(lib.gvqa_linear_f32(M, N, K, A.data_ptr(), K, W.data_ptr(), K, None, 0, C2.data_ptr(), N, st))
        t = timeit(f32)
        row["f32_us"] = t * 1e6; row["f32_tflops"] = 2.0 * M * N * K / t / 1e12
        _lib.set_option(_lib.OPT_VENDOR_GEMM, 1)
        t = timeit(f32)
        row["vendor_us"] = t * 1e6; row["vendor_tflops"] = 2.0 * M * N * K / t / 1e12
        _lib.set_option(_lib.OPT_VENDOR_GEMM, vend)
        ref = A[:2048].double() @ W.double().t()
        row["err_split3_vs_fp64"] = float((Cs["split3"][:2048].double() - ref).abs().max())
        row["err_split2h_vs_fp64"] = float((Cs["split2h"][:2048].double() - ref).abs().max())
        row["err_f32_vs_fp64"] = float((C2[:2048].double() - ref).abs().max())
        res.append(row); print(json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
