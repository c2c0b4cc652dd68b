#!/usr/bin/env python3
"""Phase stamps of the one-launch aggregate-first kernel (k_hopagg4<..., SEQ>, hop_fusion = 5), measurement build:

    GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so GVQA_HOPAGG_STAMPS=/tmp/ha_stamps.bin python scripts/probe_hopagg_seq.py

Every workgroup's wave 0 stamps the 100 MHz clock at 8 phase boundaries of every hop; printed: the mean / max duration of every
phase over workgroups, per hop, and the launch's span."""
import json, os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
from graphvqa_amd.graph import SceneGraphBatch, HostLayout
path = os.environ.get("GVQA_HOPAGG_STAMPS")
assert path, "set GVQA_HOPAGG_STAMPS (and GVQA_LIB to the measurement build)"
D, H, K, DI = int(os.environ.get("D", "512")), 4, 5, 512
dev = torch.device("cuda:0")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
gb = synth.config3_batch(int(os.environ.get("GRAPHS", "2048"))) if D == 512 else synth.config2_batch()
N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
m = gat_seq(D, D, D, DI, K, dropout=0.1, gat_heads=H)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.gat_seq_params(D, D, D, DI, K, H, seed=777).items()})
m = m.to(dev).eval()
m.hop_fusion = 5
x, ea, ins = tt(synth.normal((N, D), 1)), tt(synth.normal((E, D), 2)), tt(synth.normal((K, B, DI), 3))
ei, bt = tt(gb.edge_index), tt(gb.batch)
g = SceneGraphBatch(ei, bt, N, B, host_layout=HostLayout.from_numpy(gb.edge_index, gb.batch, B))
for _ in range(4):
    m(x, ei, ea, ins, bt, graph=g)
torch.cuda.synchronize()
raw = open(path, "rb").read()
G, KK = struct.unpack("ii", raw[:8])
raw_st = np.frombuffer(raw[8:], dtype=np.uint64).reshape(G, KK, 8)
hwid = raw_st[:, KK - 1, 7]                            # XCC_ID << 32 | HW_ID of wave 0 (gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13])
st = raw_st.astype(np.float64) / 100.0      # us
names = ["prime+produce0", "main loop", "B1+issue DMAs", "epilogue arith", "wait+B2+stores+logit FMAs", "reduce part", "B4 softmax..", "-> next hop"]
t0 = st[:, 0, 0].min()
out = {"groups": G, "K": KK, "span_us": round(float(st.max() - t0), 1), "first_start_spread_us": round(float(st[:, 0, 0].max() - t0), 1)}
for hop in range(KK):
    row = {}
    last = 7 if hop + 1 < KK else 3
    for k in range(last):
        d = st[:, hop, k + 1] - st[:, hop, k]
        row[names[k]] = [round(float(d.mean()), 2), round(float(d.max()), 2)]
    if hop + 1 < KK:
        d = st[:, hop + 1, 0] - st[:, hop, 0]
        row["hop total"] = [round(float(d.mean()), 2), round(float(d.max()), 2)]
    out[f"hop{hop} [mean, max] us"] = row
# where the spread is: per workgroup, the main loop averaged over its hops (a persistent property of the CU / XCD it ran on?) against
# the hop-to-hop variation inside a workgroup; by XCD (workgroup w of the launch order runs on XCD w % 8) and by dispatch round
ml = st[:, :, 2] - st[:, :, 1]                       # [G, K] main-loop us
per_wg = ml.mean(axis=1)
q = lambda a, p: round(float(np.percentile(a, p)), 1)
out["main_loop_per_workgroup_us (mean over hops) [min, p10, median, p90, max]"] = [q(per_wg, 0), q(per_wg, 10), q(per_wg, 50), q(per_wg, 90), q(per_wg, 100)]
out["main_loop_within_workgroup_std_us (median over workgroups)"] = round(float(np.median(ml.std(axis=1))), 2)
out["main_loop_by_xcd_us"] = [round(float(per_wg[x::8].mean()), 1) for x in range(8)]
cus = 256
if G > cus:
    out["main_loop_by_round_us"] = [round(float(per_wg[:cus].mean()), 1), round(float(per_wg[cus:].mean()), 1)]
    out["round2_start_spread_us"] = round(float(st[cus:, 0, 0].max() - st[cus:, 0, 0].min()), 1)
xcc = (hwid >> np.uint64(32)).astype(np.int64) & 0xF
hw = (hwid & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
where = xcc * 1000 + se * 100 + sh * 10 + cu * 0          # (xcc, se, sh)
order = np.argsort(-per_wg)
out["slowest_20 [workgroup, main_loop_us, xcc, se, sh, cu]"] = [[int(i), round(float(per_wg[i]), 1), int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])] for i in order[:20]]
out["fastest_10 [workgroup, main_loop_us, xcc, se, sh, cu]"] = [[int(i), round(float(per_wg[i]), 1), int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])] for i in order[-10:]]
by_cu = {}
for i in range(G):
    by_cu.setdefault((int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])), []).append(float(per_wg[i]))
out["distinct_cus_seen"] = len(by_cu)
out["main_loop_by_cu_id_us (mean over xcc / se / sh)"] = {str(c): round(float(np.mean([np.mean(v) for k, v in by_cu.items() if k[3] == c])), 1) for c in sorted({k[3] for k in by_cu})}
out["main_loop_by_se_us"] = {str(c): round(float(np.mean([np.mean(v) for k, v in by_cu.items() if k[1] == c])), 1) for c in sorted({k[1] for k in by_cu})}
hist, edges = np.histogram(per_wg, bins=12)
out["main_loop_histogram"] = [[round(float(edges[i]), 0), int(hist[i])] for i in range(len(hist))]
wg_total = st[:, KK - 1, 2] - st[:, 0, 0]
out["workgroup_total_us [min, median, max]"] = [q(wg_total, 0), q(wg_total, 50), q(wg_total, 100)]
print(json.dumps(out))
