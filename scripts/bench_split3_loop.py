"""Prices the parts of a K step of k_linear_split3 (VAR=13: three bf16 pieces, 256 x 256 tile, no C store; VAR=113: the same with
two fp16 pieces) by switching them off: GVQA_SPLIT3_LOOP_DEBUG bits 16 (no fragment reads after step 0), 32 (no DMA in the
loop), 64 (no waits / barriers)."""
import os as _os
_os.environ.setdefault("GVQA_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "graphvqa_amd", "lib", "probes", "libgvqa_hip.so"))   # the measurement build (python -m graphvqa_amd.build --probes)
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphvqa_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
M, N, K = 65536, 2048, 512
A = torch.relu(torch.randn(M, K, device=dev)); W = torch.randn(N, K, device=dev) / K ** 0.5
C = torch.empty(M, N, device=dev)
var = int(os.environ.get("VAR", "13"))
nbytes, pack, linear = ((lib.gvqa_split3_packed_bytes, lib.gvqa_split3_pack, lib.gvqa_linear_split3) if var < 100 else
                        (lib.gvqa_split2h_packed_bytes, lib.gvqa_split2h_pack, lib.gvqa_linear_split2h))
apk = torch.empty(nbytes(M, K), dtype=torch.uint8, device=dev)
wpk = torch.empty(nbytes(N, K), dtype=torch.uint8, device=dev)
_lib.check(pack(M, K, A.data_ptr(), K, apk.data_ptr(), st)); _lib.check(pack(N, K, W.data_ptr(), K, wpk.data_ptr(), st))
gemm = lambda: _lib.check(linear(M, N, K, apk.data_ptr(), wpk.data_ptr(), None, None, 0, None, 0, 0, C.data_ptr(), N, st))
def timeit(n=20, w=5):
    for _ in range(w): gemm()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): gemm()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
_lib.set_option(_lib.OPT_SPLIT3_VARIANT, var)
for rnd in range(2):
    for dbg in (0, 16, 32, 48, 64, 112):
        os.environ["GVQA_SPLIT3_LOOP_DEBUG"] = str(dbg)
        us = round(timeit(), 1)
        nb = (M // 256 if var % 100 < 20 else M // 128) * (N // 256)
        clk = C.flatten()[:2 * nb].view(nb, 2).double().cpu()
        print(json.dumps({"variant": var, "loop_debug": dbg, "us": us, "block_cycles": round(float(clk[:, 0].mean())),
                          "block_us": round(float(clk[:, 1].mean()) / 100, 2), "shader_mhz": round(float((clk[:, 0] / clk[:, 1]).mean()) * 100)}), flush=True)
