"""Stand-alone message-passing kernel (unfused hop) under the plan tunables GVQA_MP_NBUF / GVQA_MP_CW / GVQA_MP_LDS / GVQA_MP_PARTS:
config 3 and config 2, us per launch and fraction of 8 TB/s with SURVEY 8(d)'s bytes."""
import os as _os
_os.environ.setdefault("GVQA_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "graphvqa_amd", "lib", "probes", "libgvqa_hip.so"))   # the measurement build (python -m graphvqa_amd.build --probes)
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)); dev = torch.device("cuda:0")
_lib.set_option(_lib.OPT_HOP_FUSION, 0)
cases = {}
for name, gb, d in (("config3", synth.config3_batch(), 512), ("config2", synth.config2_batch(), 300)):
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    m = gat_seq(d, d, d, 512, 5, dropout=0.1, gat_heads=4)
    m.load_state_dict({k: tt(v) for k, v in synth.gat_seq_params(d, d, d, 512, 5, 4, seed=777).items()}); m = m.to(dev).eval()
    args = [tt(v).to(dev) for v in (synth.normal((N, d), 1), gb.edge_index, synth.normal((E, d), 2), synth.normal((5, B, 512), 3), gb.batch)]
    cases[name] = (m, args, 4 * (N * 4 * d + 2 * N * 4 + E * 4 + E + (N + 1) + N * d) + 4 * N * d)
PLAN_SWEEP = os.environ.get("MP_SWEEP", "plan") == "plan"
for env in ([{}] + [dict(GVQA_MP_DEBUG=str(d)) for d in (1, 2, 4, 3, 7, 15)] + [{}]) if not PLAN_SWEEP else [{}] + [dict(GVQA_MP_NBUF=str(n)) for n in (3, 4)] + [dict(GVQA_MP_NBUF=str(n), GVQA_MP_LDS=str(l)) for n in (2, 3) for l in (81920, 163840)] + \
           [dict(GVQA_MP_CW=str(c), GVQA_MP_NBUF=str(n)) for c in (256, 64) for n in (2, 3)]:
    for k in ("GVQA_MP_NBUF", "GVQA_MP_CW", "GVQA_MP_LDS", "GVQA_MP_DEBUG"): os.environ.pop(k, None)
    os.environ.update(env)
    row = {"env": env}
    for name, (m, args, alg) in cases.items():
        try:
            for _ in range(2): m(*args)
            _lib.prof_enable(True); _lib.prof_collect()
            for _ in range(4): m(*args)
            torch.cuda.synchronize(); p = _lib.prof_collect(); _lib.prof_enable(False)
            us = p["mp"][0] / p["mp"][1] * 1e3
            row[name] = {"mp_us": round(us, 1), "frac_8TBps": round(alg / (us * 1e-6) / 8e12, 3)}
        except Exception as e:
            row[name] = {"error": str(e)[:80]}
    print(json.dumps(row), flush=True)
