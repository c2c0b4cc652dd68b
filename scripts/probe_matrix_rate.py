"""gvqa_mfma_stream at several launch lengths and operand kinds: the matrix pipes' rate as a function of what the operands hold and of how long the burst is."""
import ctypes, json, sys, torch
sys.path.insert(0, ".")
from graphvqa_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
sink = torch.empty(1 << 20, device=dev)
kinds = {"normal": torch.empty(1 << 19, dtype=torch.float16, device=dev).normal_(), "zeros": torch.zeros(1 << 19, dtype=torch.float16, device=dev),
         "ones": torch.ones(1 << 19, dtype=torch.float16, device=dev),
         "small_ints": torch.randint(-4, 5, (1 << 19,), device=dev).to(torch.float16)}
for bf in (0, 1, 2, 4, 3):           # bit 0: bf16; bits 1-2: order of a step's products (1 snake, 2 same fragments 8 times in a row)
    for name, ops in kinds.items():
        if bf & 1: ops = ops.float().bfloat16()
        for iters in ((500, 4000, 32000) if bf < 2 else (4000,)):
            fl = ctypes.c_int64(0)
            f = lambda: _lib.check(lib.gvqa_mfma_stream(ops.data_ptr(), ops.numel() * 2, sink.data_ptr(), sink.numel(), iters, bf, ctypes.byref(fl), st))
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            print(json.dumps({"dtype": "bf16" if bf & 1 else "f16", "order": bf >> 1, "operands": name, "iters": iters, "launch_ms": round(ms, 3), "tflops": round(fl.value / ms / 1e9, 1)}))
