"""Phase timeline of the persistent hop kernel (csrc/hop2.hip) from its in-kernel stamps (measurement build:
python -m graphvqa_amd.build --probes; run with GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so).  Per workgroup and item:
start / main loop end / end on the 100 MHz clock.  Prints mean phase lengths and how much of a workgroup's epilogue time
its CU partner (workgroup +- grid/2) spends in its main loop."""
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
for _ in range(3): m(x, ei, ea, ins, batch, graph=g)
NWG = 512
buf = torch.zeros((NWG, 32, 4, 8), dtype=torch.int64, device=dev)
fn = lib.gvqa_probe_hop2_buffer; fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
sel = lib.gvqa_probe_hop2_select; sel.argtypes = [ctypes.c_int, ctypes.c_int]; sel.restype = ctypes.c_int
sel(K, int(os.environ.get("HOP", str(K - 1))))      # which hop's launch leaves its stamps
fn(buf.data_ptr())
m(x, ei, ea, ins, batch, graph=g)
torch.cuda.synchronize(); fn(None)
full = buf.cpu().numpy().astype(np.int64)
items = int((full[:, :, 0, 0] > 0).sum(1).max())
def phases(w):
    q = full[:, :items, w, :7].astype(np.float64)
    d = np.diff(q, axis=2) / 100.0
    return [round(float(v), 2) for v in d.mean((0, 1))]
print(json.dumps({"phase_us [main, img0+sync, agg0, sync+img1, agg1, tail]": {"wave0": phases(0), "wave3": phases(3)}}))
t = full[:, :, 0, :][:, :, [0, 1, 6, 6]]
t0 = t[:, :items, 0][t[:, :items, 0] > 0].min()
st, me, en = [(t[:, :items, k] - t0) / 100.0 for k in range(3)]
main, epi = me - st, en - me
print(json.dumps({"items_per_wg": items, "main_us_mean": round(float(main.mean()), 2), "epilogue_us_mean": round(float(epi.mean()), 2),
                  "kernel_span_us": round(float(en.max()), 1), "first_start_spread_us": round(float(st[:, 0].max()), 2),
                  "main_us_by_item": [round(float(v), 1) for v in main.mean(0)], "epi_us_by_item": [round(float(v), 1) for v in epi.mean(0)]}))
# overlap: fraction of each workgroup's epilogue time during which its CU partner is inside a main loop
half = NWG // 2
ov = []
for w in range(NWG):
    p = (w + half) % NWG
    tot = got = 0.0
    for i in range(items):
        a0, a1 = me[w, i], en[w, i]
        tot += a1 - a0
        for j in range(items):
            got += max(0.0, min(a1, me[p, j]) - max(a0, st[p, j]))
    ov.append(got / max(tot, 1e-9))
print(json.dumps({"partner_in_main_during_my_epilogue": round(float(np.mean(ov)), 3)}))
for w in (0, 1, 100):
    print("wg", w, "start/main_end/end:", [(round(float(st[w, i]), 1), round(float(me[w, i]), 1), round(float(en[w, i]), 1)) for i in range(items)])
    p = w + half
    print("wg", p, "start/main_end/end:", [(round(float(st[p, i]), 1), round(float(me[p, i]), 1), round(float(en[p, i]), 1)) for i in range(items)])
