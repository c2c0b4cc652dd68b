"""Main loop of the persistent hop kernel alone (measurement build, GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so): the epilogue is
switched off and parts of a K step are removed one at a time, with two workgroups per CU and with one.  Wrong results by design.
Reports the hop kernel's launch time and the mean main-loop time per item from the in-kernel stamps."""
import os as _os
_os.environ.setdefault("GVQA_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "graphvqa_amd", "lib", "probes", "libgvqa_hip.so"))   # the measurement build (python -m graphvqa_amd.build --probes)
import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch
torch.set_grad_enabled(False)
dev = torch.device("cuda:0"); D, H, K = 512, 4, 5
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
gb = synth.config3_batch(); N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H); m.load_state_dict({k: tt(v) for k, v in synth.gat_seq_params(D, D, D, D, K, H, seed=777).items()}); m = m.to(dev).eval()
x, ea, ins = tt(synth.normal((N, D), 1)).to(dev), tt(synth.normal((E, D), 2)).to(dev), tt(synth.normal((K, B, D), 3)).to(dev)
ei, batch = tt(gb.edge_index).to(dev), tt(gb.batch).to(dev)
g = SceneGraphBatch(ei, batch, N, B)
lib = _lib.load()
_lib.set_option(_lib.OPT_HOP_FUSION, 2)
setbuf = lib.gvqa_probe_hop2_buffer; setbuf.argtypes = [ctypes.c_void_p]; setbuf.restype = ctypes.c_int
setdbg = lib.gvqa_probe_hop2_debug; setdbg.argtypes = [ctypes.c_int]; setdbg.restype = ctypes.c_int
buf = torch.zeros((512, 32, 4, 8), dtype=torch.int64, device=dev)
names = {1: "main loop only", 3: "... no DMA", 5: "... no fragment reads", 9: "... no waits/barriers", 15: "MFMAs only"}
for one_wg in (0, 16):
    for bits in (0, 1, 3, 5, 9, 15):
        setdbg(bits | one_wg)
        for _ in range(2): m(x, ei, ea, ins, batch, graph=g)
        _lib.prof_enable(True); _lib.prof_collect()
        for _ in range(5): m(x, ei, ea, ins, batch, graph=g)
        torch.cuda.synchronize(); p = _lib.prof_collect(); _lib.prof_enable(False)
        buf.zero_(); setbuf(buf.data_ptr()); m(x, ei, ea, ins, batch, graph=g); torch.cuda.synchronize(); setbuf(None)
        f = buf.cpu().numpy().astype(np.float64)
        on = f[:, :, 0, 0] > 0
        main = ((f[:, :, 0, 1] - f[:, :, 0, 0]) / 100.0)[on]
        print(json.dumps({"workgroups_per_cu": 1 if one_wg else 2, "what": names.get(bits, "whole kernel"), "hop_kernel_us": round(p["proj"][0] / p["proj"][1] * 1e3, 1),
                          "main_loop_us_per_item": round(float(main.mean()), 2), "items_per_wg": int(on.sum(1).max())}), flush=True)
setdbg(0)
