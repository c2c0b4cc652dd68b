#!/usr/bin/env python3
"""Randomised parity sweep of the variants and "next" rows on their fused HIP paths against the oracle: tapped GINE / GCN conv
results, lcgn_seq (fp32; every third case also bf16 node features against a CPU emulation), global attention pooling + classifier, the scene-graph encoder -- random widths up to the reference's
(300 / 512), ragged batches with empty graphs, library products forced (size threshold 0) or left to the default rule.
SEED=<int> CASES=<n> python scripts/fuzz_variants.py   (round 3: seeds 1-4, 630 cases, 0 mismatches)"""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from graphvqa_amd import synth, _lib
from graphvqa_amd.baseline_models import gine_seq, gcn_seq
from graphvqa_amd.lcgn import lcgn_seq
from graphvqa_amd.pipeline_head import MyConditionalGlobalAttention, ShortAnswerClassifier
from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
from oracle import ref_torch as R

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))


def t(a, d=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x.to(d) if d else x


tp = lambda p: {k: t(v) for k, v in p.items()}


def rand_batch():
    kind = str(rng.choice(["small", "ragged", "many", "big"]))
    B = {"small": int(rng.integers(1, 4)), "ragged": int(rng.integers(2, 30)), "many": int(rng.choice([32, 64, 200])), "big": int(rng.integers(1, 5))}[kind]
    hi = {"small": 12, "ragged": 50, "many": 24, "big": 128}[kind]
    sizes = rng.integers(0 if kind == "ragged" else 1, hi + 1, size=B)
    if sizes.sum() == 0: sizes[0] = 2
    if sizes[-1] == 0: sizes[-1] = 1      # (the reference's one_hot(batch) @ Linear(cmd), lcgn.py:150-153, needs the last graph non-empty)
    batch = np.repeat(np.arange(B), sizes).astype(np.int64); offs = np.concatenate([[0], np.cumsum(sizes)])
    src, dst = [np.zeros(0, np.int64)], [np.zeros(0, np.int64)]
    dens = float(rng.choice([0.0, 1.0, 3.0]))
    for g in range(B):
        n = int(sizes[g])
        if n == 0: continue
        e = int(rng.integers(0, int(dens * n) + 1))
        src.append(np.concatenate([np.arange(n), rng.integers(0, n, size=e)]) + offs[g]); dst.append(np.concatenate([np.arange(n), rng.integers(0, n, size=e)]) + offs[g])
    return kind, B, batch, np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)


n, bad = int(os.environ.get("CASES", "60")), 0
for case in range(n):
    kind, B, batch, ei = rand_batch(); N, E = batch.shape[0], ei.shape[1]
    D = int(rng.choice([8, 12, 30, 64, 128, 300])); Di = int(rng.choice([8, 16, 512]))
    force = bool(rng.integers(0, 2))
    info = dict(case=case, kind=kind, B=B, N=N, E=E, D=D, Di=Di, force=force)
    worst = {}
    def upd(name, a, b):
        e = float((a.detach().cpu().double() - b.detach().double()).abs().max()) if a.numel() else 0.0
        s = max(1.0, float(b.abs().max())) if b.numel() else 1.0
        worst[name] = max(worst.get(name, 0.0), e / s if np.isfinite(e) else float("inf"))
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0) if force else None
    try:
        x, ea, ins = rng.standard_normal((N, D)).astype(np.float32), rng.standard_normal((E, D)).astype(np.float32), rng.standard_normal((5, B, Di)).astype(np.float32)
        p = synth.gine_seq_params(D, D, Di, seed=case); m = gine_seq(D, D, Di); m.load_state_dict(tp(p)); m = m.to(dev).eval()
        out, convs = m(t(x, dev), t(ei, dev), t(ea, dev), t(ins, dev), t(batch, dev), return_convs=True)
        ro, rc = R.gine_seq(t(x), t(ei), t(ea), t(ins), t(batch), tp(p), return_convs=True)
        upd("gine.out", out, ro); [upd("gine.conv", a, b) for a, b in zip(convs, rc)]
        p = synth.gcn_seq_params(D, D, Di, seed=case); m = gcn_seq(D, D, Di); m.load_state_dict(tp(p)); m = m.to(dev).eval()
        out, convs = m(t(x, dev), t(ei, dev), t(ins, dev), t(batch, dev), return_convs=True)
        ro, rc = R.gcn_seq(t(x), t(ei), t(ins), t(batch), tp(p), return_convs=True)
        upd("gcn.out", out, ro); [upd("gcn.conv", a, b) for a, b in zip(convs, rc)]
        O = int(rng.choice([8, 16, 40, 128, 512])); L = int(rng.integers(1, 12))
        info["O"] = O
        p = synth.lcgn_seq_params(D, O, seed=case, cmd_dim=O, question_dim=O); m = lcgn_seq(D, O, D, 5, gat_cmd_dim=O, question_dim=O)
        m.load_state_dict(tp(p), strict=False); m = m.to(dev).eval()
        q, lstm, xc = rng.standard_normal((B, O)).astype(np.float32), rng.standard_normal((L, B, O)).astype(np.float32), rng.standard_normal((N, O)).astype(np.float32)
        out = m(t(x, dev), t(ei, dev), t(batch, dev), t(q, dev), t(lstm, dev), x_ctx_init=t(xc, dev))
        upd("lcgn", out, R.lcgn_seq(t(x), t(ei), t(batch), t(q), t(lstm), tp(p), t(xc)))
        if case % 3 == 0:        # bf16 node-feature storage (BASELINE config 5's build-side mode) against a CPU emulation rounding at the same points
            from tests.test_gpu_gat import _lcgn_bf16_storage_emulation
            pieces = int(rng.choice([1, 2]))
            mb = lcgn_seq(D, O, D, 5, gat_cmd_dim=O, question_dim=O, node_feature_dtype=torch.bfloat16, bf16_weight_pieces=pieces)
            mb.load_state_dict(tp(p), strict=False); mb = mb.to(dev).eval()
            outb = mb(t(x, dev), t(ei, dev), t(batch, dev), t(q, dev), t(lstm, dev), x_ctx_init=t(xc, dev))
            emu = _lcgn_bf16_storage_emulation(t(x), t(ei), t(batch), t(q), t(lstm), tp(p), t(xc), pieces=pieces)
            eb = float((outb.cpu() - emu).abs().max()) if outb.numel() else 0.0
            worst["lcgn.bf16"] = eb / (120.0 * max(1.0, float(emu.abs().max()) if emu.numel() else 1.0))      # bound 1.2e-2 of the output scale = 3 bf16 ulps (one
            # rounding that falls the other way at a storage point is 1 ulp = 3.9e-3; seen: 4.8e-3 once in 50 cases), in 1e-4 units
        Q, A = int(rng.choice([8, 24, 512])), int(rng.choice([5, 33, 1842]))
        pp, pc = synth.attention_pool_params(D, Q, seed=case), synth.classifier_params(Q, 16, A, seed=case)
        pool, clf = MyConditionalGlobalAttention(D, Q), ShortAnswerClassifier(Q, 16, A)
        pool.load_state_dict(tp(pp)); clf.load_state_dict(tp(pc)); pool, clf = pool.to(dev).eval(), clf.to(dev).eval()
        u = rng.standard_normal((B, Q)).astype(np.float32)
        upd("head", clf(pool(t(x, dev), t(u, dev), t(batch, dev)), t(u, dev)), R.short_answer_logits(R.global_attention_pool(t(x), t(u), t(batch), tp(pp), B), t(u), tp(pc)))
        V = int(rng.choice([40, 500])); pe = synth.encoder_params(V, D, seed=case); enc = GroundTruth_SceneGraph_Encoder(V, 0, D); enc.load_state_dict(tp(pe)); enc = enc.to(dev).eval()
        xt, et = rng.integers(0, V, size=(N, 12)), rng.integers(1, V, size=(E, 1)); added = rng.choice(E, size=min(E, 5), replace=False).astype(np.int64) if E else np.zeros(0, np.int64)
        data = types.SimpleNamespace(x=t(xt, dev), edge_attr=t(et, dev), edge_index=t(ei, dev), batch=t(batch, dev), added_sym_edge=t(added, dev))
        xe, ee, _ = enc(data); rxe, ree = R.scene_graph_encoder(t(xt), t(ei), t(et), t(added), t(batch), B, tp(pe))
        upd("enc.x", xe, rxe); upd("enc.e", ee, ree)
    except Exception as e:
        worst["exception"] = float("inf"); info["exc"] = type(e).__name__ + ": " + str(e)[:200].replace("\n", " ")
    finally:
        if old is not None:
            _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    fails = {k: v for k, v in worst.items() if not v < 1e-4}
    if fails:
        bad += 1
        print("FAIL", json.dumps(info), fails, flush=True)
print(json.dumps({"cases": n, "failed": bad}))
