"""The oracle (oracle/ref_torch.py) against the golden vectors recorded from the reference's
own code (tests/golden/make_golden.py).  CPU only.  Tolerances are stated per case: the
oracle repeats the reference's op sequence, so fp32 agreement is at rounding level."""
import numpy as np
import pytest
import torch

from graphvqa_amd import synth
from graphvqa_amd.scene_graph import scene_graph_topology  # noqa: F401
from oracle import ref_torch as R
from tests.util import load_golden, t, tparams, maxabs

TOL = 2e-5   # fp32, same op order as the reference up to BLAS blocking


def test_gat_conv_small():
    meta, g = load_golden("gat_conv_small")
    p = synth.gat_seq_params(20, 8, 16, 4, 1, 4, seed=meta["param_seed"])
    p = tparams({k[len("convs.0."):]: v for k, v in p.items() if k.startswith("convs.0.")})
    out, alpha = R.gat_conv(t(g["x"]), t(g["edge_index"]), t(g["edge_attr"]), p, heads=4,
                            return_attention_weights=True)
    assert maxabs(out, g["out"]) < TOL
    assert maxabs(alpha, g["alpha"]) < TOL
    # softmax over incoming edges: per-destination sums are 1, node 9 has no in-edge
    s = np.zeros((10, 4)); np.add.at(s, g["edge_index"][1], g["alpha"])
    assert np.allclose(s[:9][np.bincount(g["edge_index"][1], minlength=10)[:9] > 0], 1, atol=1e-5)
    assert np.all(s[9] == 0)


def test_gat_conv_concat_and_pair_inputs():
    """concat=True and tuple in_channels of the reference's `gat` (gat_skip.py:78-80,136-143,162-163), recorded from its own class."""
    meta, g = load_golden("gat_conv_concat_pair")
    H = meta["heads"]
    base = {"lin_l.weight": g["p_lin_l_weight"], "lin_e.weight": g["p_lin_e_weight"], "att_l": g["p_att_l"], "att_r": g["p_att_r"],
            "att_e": g["p_att_e"]}
    pa = tparams(dict(base, **{"lin_r.weight": g["p_lin_l_weight"], "bias": g["p_bias_hc"]}))
    out, alpha = R.gat_conv(t(g["x"]), t(g["edge_index"]), t(g["edge_attr"]), pa, heads=H, concat=True, return_attention_weights=True)
    assert maxabs(out, g["out_concat"]) < TOL and maxabs(alpha, g["alpha_concat"]) < TOL
    pb = tparams(dict(base, **{"lin_r.weight": g["p_lin_r_weight"], "bias": g["p_bias_c"]}))
    out, alpha = R.gat_conv((t(g["x"]), t(g["x_r"])), t(g["edge_index"]), t(g["edge_attr"]), pb, heads=H, return_attention_weights=True)
    assert maxabs(out, g["out_pair"]) < TOL and maxabs(alpha, g["alpha_pair"]) < TOL
    pc = tparams(dict(base, **{"lin_r.weight": g["p_lin_r_weight"], "bias": g["p_bias_hc"]}))
    out = R.gat_conv((t(g["x"]), t(g["x_r"])), t(g["edge_index"]), t(g["edge_attr"]), pc, heads=H, concat=True)
    assert maxabs(out, g["out_pair_concat"]) < TOL


def test_gat_seq_small_all_hops():
    meta, g = load_golden("gat_seq_small")
    p = tparams(synth.gat_seq_params(meta["dn"], meta["dn"], meta["de"], meta["di"], meta["K"],
                                     meta["heads"], seed=meta["param_seed"]))
    out, hs, alphas = R.gat_seq(t(g["x"]), t(g["edge_index"]), t(g["edge_attr"]), t(g["instr"]),
                                t(g["batch"]), p, heads=meta["heads"], return_all=True)
    assert maxabs(out, g["out"]) < TOL
    for i in range(meta["K"]):
        assert maxabs(hs[i], g["hs"][i]) < TOL
        assert maxabs(alphas[i], g["alphas"][i]) < TOL


def test_gat_seq_train_bn():
    meta, g0 = load_golden("gat_seq_small")
    _, g = load_golden("gat_seq_small_trainbn")
    p = tparams(synth.gat_seq_params(meta["dn"], meta["dn"], meta["de"], meta["di"], meta["K"],
                                     meta["heads"], seed=meta["param_seed"]))
    out = R.gat_seq(t(g0["x"]), t(g0["edge_index"]), t(g0["edge_attr"]), t(g0["instr"]),
                    t(g0["batch"]), p, heads=meta["heads"], training_bn=True)
    assert maxabs(out, g["out"]) < 5e-5


@pytest.mark.parametrize("name", ["gat_seq_debug2_d300", "gat_seq_debug4_d300"])
def test_gat_seq_real_dims_debug_graphs(name):
    meta, g = load_golden(name)
    s = meta["input_seeds"]
    N, E, B = g["batch"].shape[0], g["edge_index"].shape[1], int(g["batch"].max()) + 1
    x, ea = synth.normal((N, 300), s["x"]), synth.normal((E, 300), s["edge_attr"])
    ins = synth.normal((5, B, 512), s["instr"])
    p = tparams(synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=meta["param_seed"]))
    out, hs, alphas = R.gat_seq(t(x), t(g["edge_index"]), t(ea), t(ins), t(g["batch"]), p,
                                return_all=True)
    assert maxabs(out, g["out"]) < 5e-5
    assert maxabs(np.stack([h.numpy() for h in hs]), g["hs"]) < 5e-5
    assert maxabs(alphas[0], g["alpha0"]) < TOL and maxabs(alphas[4], g["alpha4"]) < TOL


def test_debug_topology_pinned():
    meta, g = load_golden("debug_topology")
    want = {"2375429": (21, 85), "2354786": (12, 40), "2336498": (20, 107), "2315892": (6, 23)}
    for k, (n, e) in want.items():
        assert tuple(meta["n_e"][k]) == (n, e)
        ei = g[f"ei_{k}"]
        assert ei.shape == (2, e) and ei.max() == n - 1
        # one explicit self loop per node, emitted first for its source node
        assert (ei[0] == ei[1]).sum() >= n


def test_gine_seq_small():
    meta, g = load_golden("gine_seq_small")
    p = tparams(synth.gine_seq_params(meta["dn"], meta["dn"], meta["di"], meta["param_seed"]))
    out, convs = R.gine_seq(t(g["x"]), t(g["edge_index"]), t(g["edge_attr"]), t(g["instr"]),
                            t(g["batch"]), p, return_convs=True)
    assert maxabs(out, g["out"]) < TOL
    assert maxabs(torch.stack(convs), g["convs"]) < 5e-5


def test_gcn_seq_small_and_conv():
    meta, g = load_golden("gcn_seq_small")
    p = tparams(synth.gcn_seq_params(meta["dn"], meta["dn"], meta["di"], meta["param_seed"]))
    out, convs = R.gcn_seq(t(g["x"]), t(g["edge_index"]), t(g["instr"]), t(g["batch"]), p,
                           return_convs=True)
    assert maxabs(out, g["out"]) < TOL
    assert maxabs(torch.stack(convs), g["convs"]) < 5e-5
    meta, g = load_golden("gcn_conv_small")
    p = synth.gcn_seq_params(20, 8, 4, meta["param_seed"], num_layers=1)
    o = R.gcn_conv(t(g["x"]), t(g["edge_index"]), tparams({"weight": p["convs.0.weight"],
                                                           "bias": p["convs.0.bias"]}))
    assert maxabs(o, g["out"]) < TOL


@pytest.mark.parametrize("name", ["lcgn_seq_small", "lcgn_seq_debug4_d512"])
def test_lcgn_seq(name):
    meta, g = load_golden(name)
    s = meta["input_seeds"]
    O, in_c, L = meta["out_channels"], meta["in_channels"], meta["L"]
    N, B = g["batch"].shape[0], int(g["batch"].max()) + 1
    p = tparams(synth.lcgn_seq_params(in_c, O, seed=meta["param_seed"], cmd_dim=O, question_dim=O))
    x, q, lstm = synth.normal((N, in_c), s["x"]), synth.normal((B, O), s["q"]), synth.normal((L, B, O), s["lstm"])
    out = R.lcgn_seq(t(x), t(g["edge_index"]), t(g["batch"]), t(q), t(lstm), p, t(g["x_ctx_init"]))
    assert maxabs(out, g["out"]) < 1e-4
    # the stored noise is what torch.randn draws under the recorded seed (lcgn.py:306)
    torch.manual_seed(meta["torch_seed"])
    assert torch.equal(torch.randn(N, O), t(g["x_ctx_init"]))
    # the node-storage hook (BASELINE config 5's oracle-side model): identity storage IS the pinned function, bit for bit; bf16 storage
    # in fp64 moves the result by bf16-sized amounts only (2^-9 relative per stored tensor), never by more than 3 % of the output scale
    args = (t(g["edge_index"]), t(g["batch"]))
    assert torch.equal(R.lcgn_seq(t(x), *args, t(q), t(lstm), p, t(g["x_ctx_init"]), node_store=lambda v: v), out)
    p64 = {k: v.double() for k, v in p.items()}
    o64 = R.lcgn_seq(t(x).double(), *args, t(q).double(), t(lstm).double(), p64, t(g["x_ctx_init"]).double())
    o64s = R.lcgn_seq(t(x).double(), *args, t(q).double(), t(lstm).double(), p64, t(g["x_ctx_init"]).double(), node_store=R.bf16_storage)
    dev_ = float((o64s - o64).abs().max())
    assert 1e-5 < dev_ < 3e-2 * float(o64.abs().max()), dev_


@pytest.mark.parametrize("name", ["pool_head_small", "pool_head_debug4"])
def test_pool_and_classifier(name):
    meta, g = load_golden(name)
    in_c, ch, nans = meta["in_channels"], meta["channels"], meta["num_answers"]
    N, B = g["batch"].shape[0], int(g["batch"].max()) + 1
    pp = tparams(synth.attention_pool_params(in_c, ch, seed=meta["pool_seed"]))
    cp = tparams(synth.classifier_params(ch, ch, nans, seed=meta["fc_seed"]))
    x, u = synth.normal((N, in_c), meta["input_seeds"]["x"]), synth.normal((B, ch), meta["input_seeds"]["u"])
    pooled = R.global_attention_pool(t(x), t(u), t(g["batch"]), pp, B)
    assert maxabs(pooled, g["pooled"]) < TOL
    logits = R.short_answer_logits(pooled, t(u), cp)
    assert maxabs(logits, g["logits"]) < 5e-5


def test_scene_graph_encoder():
    meta, g = load_golden("sg_encoder_debug4")
    p = tparams(synth.encoder_params(meta["vocab"], meta["dim"], seed=meta["param_seed"], pad_idx=meta["pad_idx"]))
    B = int(g["batch"].max()) + 1
    xe, ee = R.scene_graph_encoder(t(g["x_tokens"]), t(g["edge_index"]), t(g["edge_tokens"]), t(g["added_sym_edge"]),
                                   t(g["batch"]), B, p)
    assert maxabs(ee, g["edge_attr_encoded"]) < 2e-5
    assert maxabs(xe, g["x_encoded"]) < 5e-5


def test_config1_pipeline_forward_pinned_to_the_reference():
    """BASELINE config 1: the oracle chain encoder -> gat_seq -> pooling -> logits against tensors captured from the
    REFERENCE'S OWN `PipelineModel.forward` (pipeline_model_gat.py:743-821) on debug graphs 2354786 + 2375429, batch 2
    (tests/golden/pipeline_debug2.npz; instruction vectors / question feature are the recorded outputs of the
    reference's transformer decoder / encoder, which are out of scope)."""
    meta, g = load_golden("pipeline_debug2")
    B = int(g["batch"].max()) + 1
    assert (g["batch"].shape[0], g["edge_index"].shape[1], B) == (33, 125, 2)
    pe = tparams(synth.encoder_params(meta["vocab"], 300, seed=meta["encoder_seed"], pad_idx=meta["pad_idx"]))
    pg = tparams(synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=meta["gat_seq_seed"]))
    pp = tparams(synth.attention_pool_params(300, 512, seed=meta["pool_seed"]))
    pc = tparams(synth.classifier_params(512, 512, 1842, seed=meta["fc_seed"]))
    xe, ee = R.scene_graph_encoder(t(g["x_tokens"]), t(g["edge_index"]), t(g["edge_tokens"]), t(g["added_sym_edge"]),
                                   t(g["batch"]), B, pe)
    assert maxabs(xe, g["x_encoded"]) < 5e-5 and maxabs(ee, g["edge_attr_encoded"]) < 2e-5
    h = R.gat_seq(xe, t(g["edge_index"]), ee, t(g["instr_vectors"]), t(g["batch"]), pg)
    assert maxabs(h, g["x_executed"]) < 1e-4
    q = t(g["question_feature"])
    pooled = R.global_attention_pool(h, q, t(g["batch"]), pp, B)
    assert maxabs(pooled, g["pooled"]) < 1e-4
    logits = R.short_answer_logits(pooled, q, pc)
    assert maxabs(logits, g["short_answer_logits"]) < 1e-4


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_gat_seq_backward_matches_reference_autograd(mode):
    """The oracle's autograd against gradients recorded from the REFERENCE's own gat_seq under autograd
    (tests/golden/gat_seq_small_grads.npz): pins the checker used by the GPU backward tests."""
    meta, g0 = load_golden("gat_seq_small")
    _, g = load_golden("gat_seq_small_grads")
    dn, de, di, K, H = meta["dn"], meta["de"], meta["di"], meta["K"], meta["heads"]
    p = synth.gat_seq_params(dn, dn, de, di, K, H, seed=meta["param_seed"])
    rp = {k: (v.double().requires_grad_("running" not in k) if v.is_floating_point() else v) for k, v in tparams(p).items()}
    xs = [t(g0[k]).double().requires_grad_(True) for k in ("x", "edge_attr", "instr")]
    out = R.gat_seq(xs[0], t(g0["edge_index"]), xs[1], xs[2], t(g0["batch"]), rp, heads=H, training_bn=(mode == "train"))
    (out * t(g["w"]).double()).sum().backward()
    assert maxabs(out, g[f"{mode}.out"]) < 1e-4
    scale = max(float(np.abs(g[k]).max()) for k in g if k.startswith(mode + ".d_"))
    for name, v in zip(("x", "edge_attr", "instr"), xs):
        assert maxabs(v.grad, g[f"{mode}.d_{name}"]) < 2e-4 * max(float(np.abs(g[f"{mode}.d_{name}"]).max()), 1e-3 * scale), name
    for k, v in rp.items():
        key = f"{mode}.d_{k}"
        if key not in g:
            continue
        rg = v.grad if v.grad is not None else torch.zeros_like(v)
        if k.endswith("lin_l.weight"):          # the reference's lin_r IS lin_l: its recorded gradient is the sum
            rr = rp[k.replace("lin_l", "lin_r")].grad
            rg = rg if rr is None else rg + rr
        assert maxabs(rg, g[key]) < 2e-4 * max(float(np.abs(g[key]).max()), 1e-3 * scale), k


def test_scene_graph_builder_pinned_to_the_reference_converter():
    """graphvqa_amd.scene_graph.convert_scene_graph / collate_scene_graphs against the reference's own
    `convert_one_gqa_scene_graph` (gqa_dataset_entry.py:190-372) on the four debug scene graphs and an empty one (inputs
    and outputs in tests/golden/sg_builder_debug4.npz): topology, `added_sym_edge`, per-edge token ids exactly; per-node
    token ids as multisets (the reference iterates a Python set)."""
    from graphvqa_amd.scene_graph import convert_scene_graph, collate_scene_graphs
    meta, g = load_golden("sg_builder_debug4")
    stoi = {w: i for i, w in enumerate(meta["itos"])}
    graphs = [meta["scene_graphs"][k] for k in meta["graphs"]]
    for idx, sg in enumerate(graphs):
        x, ei, et, added = convert_scene_graph(sg, stoi)
        assert [x.shape[0], ei.shape[1]] == meta["sizes"][idx]
        assert np.array_equal(ei, g[f"g{idx}.edge_index"]) and np.array_equal(et, g[f"g{idx}.edge_attr"])
        assert np.array_equal(added, g[f"g{idx}.added_sym_edge"])
        assert np.array_equal(x[:, 0], g[f"g{idx}.x"][:, 0])
        assert np.array_equal(np.sort(x, axis=1), np.sort(g[f"g{idx}.x"], axis=1))
    c = collate_scene_graphs(graphs, stoi)
    sizes = meta["sizes"]
    assert c.num_graphs == len(graphs) and c.num_nodes == sum(s[0] for s in sizes) and c.num_edges == sum(s[1] for s in sizes)
    assert np.array_equal(c.batch, np.repeat(np.arange(len(graphs)), [s[0] for s in sizes]))
    n_off = np.concatenate([[0], np.cumsum([s[0] for s in sizes])])
    e_off = np.concatenate([[0], np.cumsum([s[1] for s in sizes])])
    for idx in range(len(graphs)):
        assert np.array_equal(c.edge_index[:, e_off[idx]:e_off[idx + 1]], g[f"g{idx}.edge_index"] + n_off[idx])
    assert np.array_equal(c.added_sym_edge, np.concatenate([g[f"g{i}.added_sym_edge"] + e_off[i] for i in range(len(graphs))]))
    hl = c.host_layout()
    assert hl.graph_ptr.tolist() == n_off.tolist() and hl.edge_ptr.tolist() == e_off.tolist()


def test_native_collate_pinned_to_the_reference_converter():
    """The library's host-side collate behind the C ABI (gvqa_scene_graph_collate, csrc/collate.hip; no GPU involved) over the
    flattened, pre-tokenised scene graphs: the same tensors as the reference's converter + Batch.from_data_list on the four debug
    graphs and the empty one (tests/golden/sg_builder_debug4.npz), equal to the Python restatement field by field; malformed
    input is reported, not collated."""
    import ctypes as C
    from graphvqa_amd import _lib
    from graphvqa_amd.scene_graph import collate_scene_graphs, flatten_scene_graphs, collate_flat_scene_graphs
    meta, g = load_golden("sg_builder_debug4")
    stoi = {w: i for i, w in enumerate(meta["itos"])}
    graphs = [meta["scene_graphs"][k] for k in meta["graphs"]]
    flat = flatten_scene_graphs(graphs, stoi)
    c, ref = collate_flat_scene_graphs(flat), collate_scene_graphs(graphs, stoi)
    for k in ("x", "edge_index", "edge_attr", "added_sym_edge", "batch", "nodes_per_graph", "edges_per_graph"):
        assert np.array_equal(getattr(c, k), getattr(ref, k)), k
    sizes = meta["sizes"]
    n_off = np.concatenate([[0], np.cumsum([s[0] for s in sizes])])
    e_off = np.concatenate([[0], np.cumsum([s[1] for s in sizes])])
    for idx in range(len(graphs)):
        assert np.array_equal(c.edge_index[:, e_off[idx]:e_off[idx + 1]], g[f"g{idx}.edge_index"] + n_off[idx])
        assert np.array_equal(c.edge_attr[e_off[idx]:e_off[idx + 1]], g[f"g{idx}.edge_attr"])
        assert np.array_equal(np.sort(c.x[n_off[idx]:n_off[idx + 1]], axis=1), np.sort(g[f"g{idx}.x"], axis=1))
    assert np.array_equal(c.added_sym_edge, np.concatenate([g[f"g{i}.added_sym_edge"] + e_off[i] for i in range(len(graphs))]))
    hl, hr = c.host_layout(), ref.host_layout()
    assert hl.graph_ptr.tolist() == n_off.tolist() and hl.edge_ptr.tolist() == e_off.tolist() and hl.max_in_degree == hr.max_in_degree
    # a relation that points outside its graph
    bad = flatten_scene_graphs(graphs[:1], stoi)
    bad.rel_dst[0] = 10 ** 6
    with pytest.raises(_lib.GvqaError):
        collate_flat_scene_graphs(bad)
