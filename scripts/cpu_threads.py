import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from graphvqa_amd import synth
from oracle import ref_torch as R
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
D,H,K,nb = 512,4,5,32
gb = synth.config3_batch(nb); N,E = gb.num_nodes, gb.num_edges
p = {k: tt(v) for k,v in synth.gat_seq_params(D,D,D,D,K,H,seed=777).items()}
args = (tt(synth.normal((N,D),11)), tt(gb.edge_index), tt(synth.normal((E,D),12)), tt(synth.normal((K,nb,D),13)), tt(gb.batch), p)
print("cores", os.cpu_count())
for th in (8,16,32,64,128,256):
    torch.set_num_threads(th)
    with torch.no_grad():
        R.gat_seq(*args, heads=H)
        t0=time.perf_counter(); R.gat_seq(*args, heads=H); dt=time.perf_counter()-t0
    print(th, round(dt,3), "s", round(E/dt), "edges/s", flush=True)
