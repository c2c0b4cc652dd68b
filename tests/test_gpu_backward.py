"""GPU tests of the differentiable path: HIP message-passing backward against torch autograd through
the oracle's restatement of PyG's gather / segment softmax / scatter-add, and gat_seq parameter / input gradients
against autograd through the oracle's gat_seq.  Tolerances are relative to the gradient's scale."""
import ctypes as C
import numpy as np
import pytest
import torch

from graphvqa_amd import synth
from tests.util import t, tparams, maxabs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rel(a, b, floor=0.0):
    """max-abs error relative to the reference gradient's scale (`floor`: scale below which a gradient counts as zero
    -- e.g. a bias added right before a batch-statistics BatchNorm has an exactly zero gradient)."""
    b = b.detach().cpu().double()
    return maxabs(a, b) / max(float(b.abs().max()), floor, 1e-12)


def _mp_reference(xp, a_node, a_edge, mask, edge_index, N, H, C, slope):
    """out, alpha of the bare message passing, fp64 torch on the CPU (autograd-capable)."""
    import torch.nn.functional as F
    from oracle import ref_torch as R
    src, dst = edge_index[0], edge_index[1]
    z = a_node[:, :H].index_select(0, src) + a_node[:, H:].index_select(0, dst) + a_edge
    alpha = R.segment_softmax(F.leaky_relu(z, slope), dst, N)
    am = alpha if mask is None else alpha * mask
    msg = xp.view(N, H, C).index_select(0, src) * am.unsqueeze(-1)
    return R.scatter_add_rows(msg, dst, N).mean(dim=1), alpha


@pytest.mark.parametrize("C,H,with_mask", [(32, 4, False), (12, 4, True), (30, 1, False), (30, 2, True), (300, 4, True),
                                           (512, 8, False), (516, 2, False), (32, 3, True)])
def test_message_passing_backward_vs_autograd(dev, C, H, with_mask):
    from graphvqa_amd.gat_skip import gat_message_passing
    from graphvqa_amd.graph import SceneGraphBatch
    gb = synth.make_graph_batch(6, seed=0xB00 + C, nodes_lo=5, nodes_hi=40, rel_per_node=2.0)
    ei = gb.edge_index.copy()
    # a hub: > 64 in-edges into node 0 of the first graph (recompute path of the by-destination kernel) and a
    # node without in-edges is impossible here (self loops), so also drop nothing
    n0 = int((gb.batch == 0).sum())
    extra = np.stack([np.arange(70) % n0, np.zeros(70, np.int64)])
    ei = np.concatenate([ei, extra], axis=1)
    N, E = gb.num_nodes, ei.shape[1]
    rng = np.random.default_rng(C * 7 + H)
    xp = rng.standard_normal((N, H * C)).astype(np.float32)
    a_node = rng.standard_normal((N, 2 * H)).astype(np.float32)
    a_edge = rng.standard_normal((E, H)).astype(np.float32)
    mask = ((rng.random((E, H)) > 0.3) / 0.7).astype(np.float32) if with_mask else None
    w = rng.standard_normal((N, C)).astype(np.float32)

    g = SceneGraphBatch(t(ei, device=dev), t(gb.batch, device=dev), N, gb.num_graphs)
    xs = [t(a, device=dev).requires_grad_(True) for a in (xp, a_node, a_edge)]
    out, alpha = gat_message_passing(xs[0], xs[1], xs[2], g, H, C, 0.2, None if mask is None else t(mask, device=dev))
    (out * t(w, device=dev)).sum().backward()

    rs = [t(a).double().requires_grad_(True) for a in (xp, a_node, a_edge)]
    ref_out, ref_alpha = _mp_reference(rs[0], rs[1], rs[2], None if mask is None else t(mask).double(), t(ei), N, H, C, 0.2)
    (ref_out * t(w).double()).sum().backward()
    assert maxabs(out, ref_out) < 1e-5 and maxabs(alpha, ref_alpha) < 1e-6
    for got, ref, name in zip(xs, rs, ("dxp", "da_node", "da_edge")):
        assert _rel(got.grad, ref.grad) < 2e-5, name


def _grads_vs_oracle(dev, train, alpha_masks=None, feature_masks=None, dims=(32, 32, 48, 3, 4), seed=5, graphs=5, nodes=(6, 20), rel=1.5):
    from oracle import ref_torch as R
    from graphvqa_amd.gat_skip import gat_seq
    dn, de, di, K, H = dims
    gb = synth.make_graph_batch(graphs, seed=0xA11CE + seed, nodes_lo=nodes[0], nodes_hi=nodes[1], rel_per_node=rel)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(dn, dn, de, di, K, H, seed=seed)
    rng = np.random.default_rng(seed)
    for k in list(p):                                   # non-trivial BN affine / bias so that their gradients matter
        if k.endswith("bias") or k.endswith("bns.weight"):
            p[k] = (p[k] + 0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
    x, ea, ins = synth.normal((N, dn), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    w = synth.normal((N, dn), 4)

    m = gat_seq(dn, dn, de, di, K, dropout=0.0, gat_heads=H)
    m.load_state_dict({k: t(v) for k, v in p.items()})
    m = m.to(dev)
    m.train(train)
    xs = [t(a, device=dev).requires_grad_(True) for a in (x, ea, ins)]
    kw = {}
    if alpha_masks is not None:
        out = m._forward_autograd(xs[0], t(gb.edge_index, device=dev), xs[1], xs[2], t(gb.batch, device=dev),
                                  __import__("graphvqa_amd.graph", fromlist=["SceneGraphBatch"]).SceneGraphBatch(
                                      t(gb.edge_index, device=dev), t(gb.batch, device=dev), N, B),
                                  alpha_masks=[t(a, device=dev) for a in alpha_masks],
                                  feature_masks=[t(a, device=dev) for a in feature_masks])
    else:
        out = m(xs[0], t(gb.edge_index, device=dev), xs[1], xs[2], t(gb.batch, device=dev), **kw)
    (out * t(w, device=dev)).sum().backward()

    rp = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k and "num_batches" not in k)
          for k, v in tparams(p).items()}
    rs = [t(a).double().requires_grad_(True) for a in (x, ea, ins)]
    ref = R.gat_seq(rs[0], t(gb.edge_index), rs[1], rs[2], t(gb.batch), rp, heads=H, training_bn=train,
                    alpha_masks=None if alpha_masks is None else [t(a).double() for a in alpha_masks],
                    feature_masks=None if feature_masks is None else [t(a).double() for a in feature_masks])
    (ref * t(w).double()).sum().backward()
    assert maxabs(out, ref) < 1e-4
    worst = {}
    floor = 1e-2 * max(float(r.grad.abs().max()) for r in list(rs) + [v for v in rp.values() if v.grad is not None] if r.grad.numel())
    for got, r, name in zip(xs, rs, ("x", "edge_attr", "instr_vectors")):
        if r.numel():                                   # (ins_dim = 0: no instruction vectors)
            worst[name] = _rel(got.grad, r.grad, floor)
    sd = dict(m.named_parameters())
    for k, r in rp.items():
        if not r.requires_grad or k.endswith("lin_r.weight"):
            continue
        rg = r.grad
        if k.endswith("lin_l.weight"):                  # lin_r is lin_l (gat_skip.py:76-77): one parameter, summed gradient
            rr = rp[k.replace("lin_l", "lin_r")].grad
            rg = rg if rr is None else rg + rr
        worst[k] = _rel(sd[k].grad, rg, floor)
    bad = {k: v for k, v in worst.items() if not v < 2e-4}
    assert not bad, bad
    return N, E, K, H


def test_gat_seq_gradients_eval_bn(dev):
    _grads_vs_oracle(dev, train=False)


def test_gat_seq_gradients_train_bn(dev):
    _grads_vs_oracle(dev, train=True)


def test_gat_seq_gradients_with_dropout_masks(dev):
    """Attention dropout (gat_skip.py:205) and feature dropout (:276) with GIVEN masks: forward and every gradient
    match the oracle applying the same masks (the drawn masks themselves are implementation-specific)."""
    dims = (32, 32, 48, 3, 4)
    gb = synth.make_graph_batch(5, seed=0xA11CE + 9, nodes_lo=6, nodes_hi=20, rel_per_node=1.5)
    rng = np.random.default_rng(99)
    am = [((rng.random((gb.num_edges, 4)) > 0.1) / 0.9).astype(np.float32) for _ in range(3)]
    fm = [((rng.random((gb.num_nodes, 32)) > 0.1) / 0.9).astype(np.float32) for _ in range(2)]
    _grads_vs_oracle(dev, train=True, alpha_masks=am, feature_masks=fm, dims=dims, seed=9)


def test_gat_seq_train_step_with_dropout_runs_and_learns(dev):
    """.train() with p > 0 draws its own masks; a few SGD steps on a fixed batch reduce the loss, running statistics
    move, and .eval() afterwards reproduces the fused inference path on the updated weights."""
    from graphvqa_amd.gat_skip import gat_seq
    gb = synth.make_graph_batch(16, seed=0xD0, nodes_lo=10, nodes_hi=30, rel_per_node=1.5)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    torch.manual_seed(0)
    m = gat_seq(64, 64, 64, 96, 3, dropout=0.1, gat_heads=4).to(dev).train()
    x, ea, ins = [t(a, device=dev) for a in (synth.normal((N, 64), 1), synth.normal((E, 64), 2), synth.normal((3, B, 96), 3))]
    ei, b = t(gb.edge_index, device=dev), t(gb.batch, device=dev)
    target = t(synth.normal((N, 64), 4), device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=3e-3)
    losses = []
    for _ in range(60):
        opt.zero_grad()
        loss = ((m(x, ei, ea, ins, b) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses
    assert float(m.bns[0].running_mean.abs().max()) > 0 and int(m.bns[0].num_batches_tracked) == 60
    m.eval()
    with torch.no_grad():
        fused = m(x, ei, ea, ins, b)                                   # fused inference kernels
    diff = m._forward_autograd(x, ei, ea, ins, b, __import__("graphvqa_amd.graph", fromlist=["x"]).SceneGraphBatch(ei, b, N, B))
    assert maxabs(fused, diff) < 1e-4 * (1.0 + float(fused.abs().max()))


@pytest.mark.parametrize("N,C", [(1000, 32), (5000, 300), (257, 7)])
def test_bn_relu_train_forward_backward_vs_torch(dev, N, C):
    """gvqa_bn_relu_train_* against torch's BatchNorm1d(train) + ReLU in fp64, incl. the running-statistics update."""
    from graphvqa_amd.gat_skip import _bn_relu_train
    rng = np.random.default_rng(N + C)
    x = rng.standard_normal((N, C)).astype(np.float32) * 2 + 0.5
    w, b = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32), (0.2 * rng.standard_normal(C)).astype(np.float32)
    g = rng.standard_normal((N, C)).astype(np.float32)
    bn = torch.nn.BatchNorm1d(C).to(dev).train()
    ref = torch.nn.BatchNorm1d(C).double().train()
    with torch.no_grad():
        bn.weight.copy_(t(w)); bn.bias.copy_(t(b)); ref.weight.copy_(t(w)); ref.bias.copy_(t(b))
    xg = t(x, device=dev).requires_grad_(True)
    y = _bn_relu_train(bn, xg)
    (y * t(g, device=dev)).sum().backward()
    xr = t(x).double().requires_grad_(True)
    yr = torch.relu(ref(xr))
    (yr * t(g).double()).sum().backward()
    assert maxabs(y, yr) < 1e-5
    assert _rel(xg.grad, xr.grad) < 1e-5 and _rel(bn.weight.grad, ref.weight.grad) < 1e-5 and _rel(bn.bias.grad, ref.bias.grad) < 1e-5
    assert maxabs(bn.running_mean, ref.running_mean) < 1e-6 and maxabs(bn.running_var, ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1


def test_gat_layer_gradients_vs_oracle(dev):
    """The single `gat` layer (gat_skip.py:111-177) is differentiable too."""
    from oracle import ref_torch as R
    from graphvqa_amd.gat_skip import gat
    gb = synth.make_graph_batch(4, seed=0x6A7, nodes_lo=6, nodes_hi=20, rel_per_node=1.5)
    N, E = gb.num_nodes, gb.num_edges
    conv = gat(24, 8, 20, heads=4, concat=False, negative_slope=0.2, dropout=0.0, bias=True).to(dev)
    x, ea, w = synth.normal((N, 24), 1), synth.normal((E, 20), 2), synth.normal((N, 8), 3)
    xs = [t(a, device=dev).requires_grad_(True) for a in (x, ea)]
    out = conv(xs[0], t(gb.edge_index, device=dev), xs[1])
    (out * t(w, device=dev)).sum().backward()
    rp = {k: v.detach().cpu().double().requires_grad_(True) for k, v in conv.state_dict().items() if k != "lin_r.weight"}
    rp["lin_r.weight"] = rp["lin_l.weight"]
    rs = [t(a).double().requires_grad_(True) for a in (x, ea)]
    ref = R.gat_conv(rs[0], t(gb.edge_index), rs[1], rp, heads=4)
    (ref * t(w).double()).sum().backward()
    assert maxabs(out, ref) < 1e-5
    assert _rel(xs[0].grad, rs[0].grad) < 1e-4 and _rel(xs[1].grad, rs[1].grad) < 1e-4
    for k, v in conv.named_parameters():
        assert _rel(v.grad, rp[k].grad) < 1e-4, k


def test_gine_gcn_tapped_convs_gradients_vs_oracle(dev):
    """The conv results the reference computes (pipeline_model_gine.py:665, pipeline_model_gcn.py:660) are differentiable: a loss on
    all five tapped results of gine_seq / gcn_seq gives every conv parameter, the BatchNorm affines on the way, x, edge_attr and the
    instruction vectors the gradients autograd gives through the oracle's restatement (fp64); hubs, self loops, duplicate edges
    and isolated nodes included; the same weights under no_grad run the fused inference kernels and agree."""
    from oracle import ref_torch as R
    from graphvqa_amd.baseline_models import gine_seq, gcn_seq
    gb = synth.make_graph_batch(6, seed=0x61E, nodes_lo=1, nodes_hi=17, rel_per_node=1.7)
    N, B, E, D, Di = gb.num_nodes, gb.num_graphs, gb.num_edges, 24, 12
    ei_np = np.array(gb.edge_index)
    ei_np[1, ::7] = ei_np[0, ::7]              # explicit self loops (GCN drops them and adds its own; GINE keeps them)
    ei_np[:, 1] = ei_np[:, 2]                  # a duplicate edge
    loops = ei_np[0] == ei_np[1]
    assert loops.any() and (~loops).any()
    x, ea, ins = synth.normal((N, D), 1), synth.normal((E, D), 2), synth.normal((5, B, Di), 3)
    ws = [synth.normal((N, D), 10 + i) for i in range(5)]
    ei, b = t(ei_np, device=dev), t(gb.batch, device=dev)
    for kind in ("gine", "gcn"):
        p = (synth.gine_seq_params if kind == "gine" else synth.gcn_seq_params)(D, D, Di, seed=77)
        rng = np.random.default_rng(5)
        for k in list(p):
            if k.endswith("bias") and "bns" not in k:
                p[k] = (p[k] + 0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
        m = (gine_seq if kind == "gine" else gcn_seq)(D, D, Di)
        m.load_state_dict({k: t(v) for k, v in p.items()})
        m = m.to(dev).eval()
        xs = [t(a, device=dev).requires_grad_(True) for a in ((x, ea, ins) if kind == "gine" else (x, ins))]
        args = (xs[0], ei, xs[1], xs[2], b) if kind == "gine" else (xs[0], ei, xs[1], b)
        out, convs = m(*args, return_convs=True)
        loss = sum((c * t(w, device=dev)).sum() for c, w in zip(convs, ws)) + out.sum()
        loss.backward()
        rp = {k: (v.double().requires_grad_("running" not in k and not k.endswith("eps")) if v.is_floating_point() else v)
              for k, v in tparams(p).items()}
        rs = [t(a).double().requires_grad_(True) for a in ((x, ea, ins) if kind == "gine" else (x, ins))]
        rargs = (rs[0], t(ei_np), rs[1], rs[2], t(gb.batch)) if kind == "gine" else (rs[0], t(ei_np), rs[1], t(gb.batch))
        rout, rconvs = (R.gine_seq if kind == "gine" else R.gcn_seq)(*rargs, rp, return_convs=True)
        (sum((c * t(w).double()).sum() for c, w in zip(rconvs, ws)) + rout.sum()).backward()
        assert maxabs(out, rout) < 1e-4
        for c, rc in zip(convs, rconvs):
            assert maxabs(c, rc) < 1e-4 * max(1.0, float(rc.detach().abs().max()))
        floor = 1e-3 * max(float(v.grad.abs().max()) for v in list(rp.values()) + rs if getattr(v, "grad", None) is not None)
        bad = {}
        for got, r, name in zip(xs, rs, ("x", "edge_attr", "ins") if kind == "gine" else ("x", "ins")):
            bad[name] = _rel(got.grad, r.grad, floor)
        for k, v in m.named_parameters():
            rg = rp[k].grad if rp[k].grad is not None else torch.zeros_like(rp[k])
            bad[k] = _rel(v.grad if v.grad is not None else torch.zeros_like(v), rg, floor)
        bad = {k: e for k, e in bad.items() if not e < 5e-4}
        assert not bad, (kind, bad)
        with torch.no_grad():
            fout, fconvs = m(*args, return_convs=True)
        assert maxabs(fout, out) < 1e-4
        for c, fc in zip(convs, fconvs):
            assert maxabs(c, fc) < 1e-4 * max(1.0, float(fc.abs().max()))
        # training mode: batch statistics in the chain, running statistics updated, still differentiable
        m.train()
        m.zero_grad()
        out_t, convs_t = m(*args, return_convs=True)
        sum(c.sum() for c in convs_t).backward()
        assert int(m.bns[0].num_batches_tracked) == 1 and out_t.shape == (N, D)
        assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in m.convs.parameters() if q.requires_grad)


def test_lcgn_seq_gradients_vs_oracle(dev):
    """lcgn_seq (lcgn.py:303-323) is differentiable: gradients of every parameter and input against autograd through
    the oracle's lcgn_seq; the same weights under no_grad run the fused inference path and agree."""
    from oracle import ref_torch as R
    from graphvqa_amd.lcgn import lcgn_seq
    gb = synth.make_graph_batch(5, seed=0x1C6, nodes_lo=4, nodes_hi=14, rel_per_node=1.5)
    N, B, O, L = gb.num_nodes, gb.num_graphs, 32, 6
    p = synth.lcgn_seq_params(20, O, seed=11, cmd_dim=O, question_dim=O)
    rng = np.random.default_rng(3)
    for k in list(p):
        if k.endswith("bias"):
            p[k] = (p[k] + 0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
    m = lcgn_seq(20, O, 20, 5, gat_cmd_dim=O, question_dim=O)
    missing, unexpected = m.load_state_dict({k: t(v) for k, v in p.items()}, strict=False)
    assert not unexpected and all(k.startswith("bns.") for k in missing), (missing, unexpected)
    m = m.to(dev).eval()
    x, q, lstm, xc, w = (synth.normal((N, 20), 1), synth.normal((B, O), 2), synth.normal((L, B, O), 3),
                         synth.normal((N, O), 4), synth.normal((N, O), 5))
    xs = [t(a, device=dev).requires_grad_(True) for a in (x, q, lstm)]
    ei, b = t(gb.edge_index, device=dev), t(gb.batch, device=dev)
    out = m(xs[0], ei, b, xs[1], xs[2], x_ctx_init=t(xc, device=dev))
    (out * t(w, device=dev)).sum().backward()
    rp = {k: v.double().requires_grad_(True) for k, v in tparams(p).items() if v.is_floating_point()}
    rs = [t(a).double().requires_grad_(True) for a in (x, q, lstm)]
    ref = R.lcgn_seq(rs[0], t(gb.edge_index), t(gb.batch), rs[1], rs[2], rp, t(xc).double())
    (ref * t(w).double()).sum().backward()
    assert maxabs(out, ref) < 1e-4
    floor = 1e-3 * max(float(v.grad.abs().max()) for v in list(rp.values()) + rs if v.grad is not None)
    bad = {}
    for got, r, name in zip(xs, rs, ("x", "q_encoding", "lstm_outputs")):
        bad[name] = _rel(got.grad, r.grad, floor)
    for k, v in m.named_parameters():
        if k.startswith("bns."):
            continue
        rg = rp[k].grad if rp[k].grad is not None else torch.zeros_like(rp[k])
        bad[k] = _rel(v.grad if v.grad is not None else torch.zeros_like(v), rg, floor)
    bad = {k: e for k, e in bad.items() if not e < 5e-4}
    assert not bad, bad
    with torch.no_grad():
        fused = m(xs[0], ei, b, xs[1], xs[2], x_ctx_init=t(xc, device=dev))
    assert maxabs(fused, out) < 1e-4


def test_lcgn_bf16_node_features_are_trainable(dev):
    """BASELINE config 5's storage mode under autograd: the differentiable path rounds the per-node tensors to bf16 where the
    inference kernels store them (straight-through gradient).  Forward: within bf16-storage distance of the bf16 inference
    kernels on the same weights (<= 3e-3 of the output scale, the bound test_lcgn_bf16_node_features states for its emulation);
    gradients: those of the fp32 model up to the storage rounding (<= 5 % of each gradient's scale against fp64 autograd through
    the oracle), and really different from the fp32 path's (the rounding is in the graph, not skipped)."""
    from oracle import ref_torch as R
    from graphvqa_amd.lcgn import lcgn_seq
    gb = synth.make_graph_batch(12, seed=0xBF16, nodes_lo=4, nodes_hi=20, rel_per_node=1.5)
    N, B, O, L, Din = gb.num_nodes, gb.num_graphs, 64, 6, 40
    p = synth.lcgn_seq_params(Din, O, seed=21, cmd_dim=O, question_dim=O)
    m = lcgn_seq(Din, O, Din, 5, gat_cmd_dim=O, question_dim=O, node_feature_dtype=torch.bfloat16)
    m32 = lcgn_seq(Din, O, Din, 5, gat_cmd_dim=O, question_dim=O)
    for mod in (m, m32):
        missing, unexpected = mod.load_state_dict({k: t(v) for k, v in p.items()}, strict=False)
        assert not unexpected and all(k.startswith("bns.") for k in missing)
    m, m32 = m.to(dev).eval(), m32.to(dev).eval()
    x, q, lstm, xc, w = (synth.normal((N, Din), 1), synth.normal((B, O), 2), synth.normal((L, B, O), 3),
                         synth.normal((N, O), 4), synth.normal((N, O), 5))
    ei, b = t(gb.edge_index, device=dev), t(gb.batch, device=dev)
    grads = {}
    for name, mod in (("bf16", m), ("fp32", m32)):
        xs = [t(a, device=dev).requires_grad_(True) for a in (x, q, lstm)]
        out = mod(xs[0], ei, b, xs[1], xs[2], x_ctx_init=t(xc, device=dev))
        (out * t(w, device=dev)).sum().backward()
        grads[name] = (out.detach(), [v.grad for v in xs], {k: v.grad for k, v in mod.named_parameters() if not k.startswith("bns.")})
    rp = {k: v.double().requires_grad_(True) for k, v in tparams(p).items() if v.is_floating_point()}
    rs = [t(a).double().requires_grad_(True) for a in (x, q, lstm)]
    ref = R.lcgn_seq(rs[0], t(gb.edge_index), t(gb.batch), rs[1], rs[2], rp, t(xc).double())
    (ref * t(w).double()).sum().backward()
    scale = float(ref.abs().max())
    with torch.no_grad():
        fused = m(t(x, device=dev), ei, b, t(q, device=dev), t(lstm, device=dev), x_ctx_init=t(xc, device=dev))
    out_b, gin_b, gp_b = grads["bf16"]
    print("lcgn bf16 autograd: forward vs bf16 inference %.3e, vs fp64 oracle %.3e (scale %.3e)" % (maxabs(out_b, fused), maxabs(out_b, ref), scale))
    assert maxabs(out_b, fused) < 3e-3 * max(scale, 1.0)
    assert 1e-5 < maxabs(out_b, grads["fp32"][0]) < 3e-2 * scale
    floor = 1e-2 * max(float(v.grad.abs().max()) for v in list(rp.values()) + rs if v.grad is not None)
    bad = {}
    for got, r, name in zip(gin_b, rs, ("x", "q_encoding", "lstm_outputs")):
        bad[name] = _rel(got, r.grad, floor)
    for k, g in gp_b.items():
        rg = rp[k].grad if rp[k].grad is not None else torch.zeros_like(rp[k])
        bad[k] = _rel(g if g is not None else torch.zeros_like(rg), rg, floor)
    print("lcgn bf16 autograd: worst gradient deviation from the fp32 model %.3e" % max(bad.values()))
    bad = {k: e for k, e in bad.items() if not e < 5e-2}
    assert not bad, bad
    assert maxabs(gin_b[0], grads["fp32"][1][0]) > 0.0


def test_pooling_and_classifier_gradients_vs_oracle(dev):
    """MyConditionalGlobalAttention + logit_fc (pipeline_model_gat.py:149-181, 814-816) are differentiable: per-graph
    broadcast / softmax / sum on the HIP per-graph ops; gradients against autograd through the oracle."""
    from oracle import ref_torch as R
    from graphvqa_amd.pipeline_head import MyConditionalGlobalAttention, ShortAnswerClassifier
    gb = synth.make_graph_batch(7, seed=0x9001, nodes_lo=1, nodes_hi=25, rel_per_node=1.0)
    N, B, D, A = gb.num_nodes, gb.num_graphs, 32, 50
    pp, cp = synth.attention_pool_params(D, D, seed=3), synth.classifier_params(D, 24, A, seed=4)
    pool, clf = MyConditionalGlobalAttention(D, D), ShortAnswerClassifier(D, 24, A)
    pool.load_state_dict({k: t(v) for k, v in pp.items()}); clf.load_state_dict({k: t(v) for k, v in cp.items()})
    pool, clf = pool.to(dev).eval(), clf.to(dev).eval()
    x, u, w = synth.normal((N, D), 1), synth.normal((B, D), 2), synth.normal((B, A), 3)
    xs = [t(a, device=dev).requires_grad_(True) for a in (x, u)]
    logits = clf(pool(xs[0], xs[1], t(gb.batch, device=dev)), xs[1])
    (logits * t(w, device=dev)).sum().backward()
    rpp = {k: v.double().requires_grad_(True) for k, v in tparams(pp).items()}
    rcp = {k: v.double().requires_grad_(True) for k, v in tparams(cp).items()}
    rs = [t(a).double().requires_grad_(True) for a in (x, u)]
    ref = R.short_answer_logits(R.global_attention_pool(rs[0], rs[1], t(gb.batch), rpp, B), rs[1], rcp)
    (ref * t(w).double()).sum().backward()
    assert maxabs(logits, ref) < 1e-4
    # gate_nn.2.bias shifts every score of a softmax: its gradient is exactly zero -> compare against a floor
    floor = 1e-3 * max(float(v.grad.abs().max()) for v in list(rpp.values()) + list(rcp.values()) + rs)
    assert _rel(xs[0].grad, rs[0].grad, floor) < 1e-4 and _rel(xs[1].grad, rs[1].grad, floor) < 1e-4
    for k, v in pool.named_parameters():
        assert _rel(v.grad, rpp[k].grad, floor) < 1e-4, k
    for k, v in clf.named_parameters():
        assert _rel(v.grad, rcp[k].grad, floor) < 1e-4, k
    with torch.no_grad():                                    # and the fused inference kernels agree with the same weights
        fused = clf(pool(xs[0], xs[1], t(gb.batch, device=dev)), xs[1])
    assert maxabs(fused, logits) < 1e-4


def test_gine_gcn_as_written_modules_train(dev):
    """The reference's gine_seq / gcn_seq discard their conv results: the differentiable / training forward is the
    BatchNorm-ReLU-dropout chain on x (pipeline_model_gine.py:665-671), and agrees with the fused inference chain."""
    from graphvqa_amd.baseline_models import gine_seq, gcn_seq
    gb = synth.make_graph_batch(3, seed=5, nodes_lo=4, nodes_hi=9, rel_per_node=1.0)
    N, B = gb.num_nodes, gb.num_graphs
    x = t(synth.normal((N, 8), 1), device=dev)
    ei, b = t(gb.edge_index, device=dev), t(gb.batch, device=dev)
    ea, ins = t(synth.normal((gb.num_edges, 8), 2), device=dev), t(synth.normal((5, B, 8), 3), device=dev)
    for m, args in ((gine_seq(8, 8, 8, dropout=0.0), (x, ei, ea, ins, b)), (gcn_seq(8, 8, 8, dropout=0.0), (x, ei, ins, b))):
        m = m.to(dev).eval()
        xg = args[0].clone().requires_grad_(True)
        out = m(xg, *args[1:])
        out.sum().backward()
        assert xg.grad is not None and m.bns[0].weight.grad is not None
        with torch.no_grad():
            assert maxabs(m(*args), out) < 1e-5
        m.train()
        assert m(*args).shape == (N, 8) and int(m.bns[0].num_batches_tracked) == 1


def test_config1_pipeline_end_to_end_gradients(dev):
    """BASELINE config 1 shape, trainable end to end: encoder -> gat_seq -> pooling -> classifier on the differentiable
    paths (eval-mode statistics so that the oracle chain is the exact reference); the loss gradient reaches every
    parameter, the embedding table included, and matches autograd through the same chain on the oracle."""
    import types
    from oracle import ref_torch as R
    from tests.util import load_golden
    from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.pipeline_head import MyConditionalGlobalAttention, ShortAnswerClassifier
    meta, gg = load_golden("gat_seq_debug2_d300")
    ei, batch = gg["edge_index"], gg["batch"]
    N, E, B, V, D, Q, A = batch.shape[0], ei.shape[1], 2, 50, 64, 96, 40
    x_tok = synth.randint(N * 12, 71, 0, V, stream=9).reshape(N, 12)
    e_tok = synth.randint(E, 72, 1, V, stream=9).reshape(E, 1)
    added = np.array([3, 17, 60], dtype=np.int64)
    ins, q, w = synth.normal((3, B, Q), 73), synth.normal((B, Q), 74), synth.normal((B, A), 75)
    pe, pg = synth.encoder_params(V, D, seed=1), synth.gat_seq_params(D, D, D, Q, 3, 4, seed=2)
    pp, pc = synth.attention_pool_params(D, Q, seed=3), synth.classifier_params(Q, 48, A, seed=4)
    mods = [GroundTruth_SceneGraph_Encoder(V, 0, D), gat_seq(D, D, D, Q, 3, dropout=0.0, gat_heads=4),
            MyConditionalGlobalAttention(D, Q), ShortAnswerClassifier(Q, 48, A)]
    for m, p in zip(mods, (pe, pg, pp, pc)):
        m.load_state_dict({k: t(v) for k, v in p.items()})
        m.to(dev).eval()
    enc, gs, pool, clf = mods
    data = types.SimpleNamespace(x=t(x_tok, device=dev), edge_attr=t(e_tok, device=dev), edge_index=t(ei, device=dev),
                                 batch=t(batch, device=dev), added_sym_edge=t(added, device=dev))
    xe, ee, _ = enc(data)
    h = gs(xe, data.edge_index, ee, t(ins, device=dev), data.batch)
    logits = clf(pool(h, t(q, device=dev), data.batch), t(q, device=dev))
    (logits * t(w, device=dev)).sum().backward()

    rp = [{k: (v.double().requires_grad_("running" not in k) if v.is_floating_point() else v)
           for k, v in tparams(p).items()} for p in (pe, pg, pp, pc)]
    rxe, ree = R.scene_graph_encoder(t(x_tok), t(ei), t(e_tok), t(added), t(batch), B, rp[0])
    rh = R.gat_seq(rxe, t(ei), ree, t(ins).double(), t(batch), rp[1], heads=4)
    ref = R.short_answer_logits(R.global_attention_pool(rh, t(q).double(), t(batch), rp[2], B), t(q).double(), rp[3])
    (ref * t(w).double()).sum().backward()
    assert maxabs(logits, ref) < 1e-4
    grads = [v.grad for d in rp for v in d.values() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None]
    floor = 1e-3 * max(float(g.abs().max()) for g in grads)
    bad = {}
    for m, d in zip(mods, rp):
        for k, v in m.named_parameters():
            rg = d[k].grad
            if k.endswith("lin_l.weight"):
                rr = d[k.replace("lin_l", "lin_r")].grad
                rg = rg if rr is None else rg + rr
            if rg is None:
                rg = torch.zeros_like(d[k])
            if k == "sg_vocab_embedding.weight":      # nn.Embedding(padding_idx=0): the pad row receives no gradient
                rg = rg.clone()
                rg[0] = 0
            assert v.grad is not None, k
            e = _rel(v.grad, rg, floor)
            if not e < 5e-4:
                bad[type(m).__name__ + "." + k] = e
    assert not bad, bad


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_gat_seq_gradients_vs_reference_golden(dev, mode):
    """Forward + backward of gat_seq on the HIP differentiable path against gradients recorded from the REFERENCE's own
    gat_seq under autograd (tests/golden/gat_seq_small_grads.npz; ragged graphs incl. single-node ones)."""
    from tests.util import load_golden
    from graphvqa_amd.gat_skip import gat_seq
    meta, g0 = load_golden("gat_seq_small")
    _, g = load_golden("gat_seq_small_grads")
    dn, de, di, K, H = meta["dn"], meta["de"], meta["di"], meta["K"], meta["heads"]
    p = synth.gat_seq_params(dn, dn, de, di, K, H, seed=meta["param_seed"])
    m = gat_seq(dn, dn, de, di, K, dropout=0.0, gat_heads=H)
    m.load_state_dict({k: t(v) for k, v in p.items()})
    m = m.to(dev).train(mode == "train")
    xs = [t(g0[k], device=dev).requires_grad_(True) for k in ("x", "edge_attr", "instr")]
    out = m(xs[0], t(g0["edge_index"], device=dev), xs[1], xs[2], t(g0["batch"], device=dev))
    (out * t(g["w"], device=dev)).sum().backward()
    assert maxabs(out, g[f"{mode}.out"]) < 1e-4
    scale = max(float(np.abs(g[k]).max()) for k in g if k.startswith(mode + ".d_"))
    bad = {}
    for name, v in list(zip(("x", "edge_attr", "instr"), xs)) + list(m.named_parameters()):
        ref = g[f"{mode}.d_{name}"]
        err = maxabs(v.grad, ref) / max(float(np.abs(ref).max()), 1e-2 * scale)
        if not err < 3e-4:
            bad[name] = err
    assert not bad, bad


def test_differentiable_path_matches_fused_path_at_config3_size(dev):
    """BASELINE config 3 (64k nodes / 256k edges, d=512, K=5): the differentiable formulation reproduces the fused
    inference kernels' output, and a backward pass through it yields finite gradients for every parameter."""
    from graphvqa_amd.gat_skip import gat_seq
    D, H, K = 512, 4, 5
    gb = synth.config3_batch()
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    m = gat_seq(D, D, D, D, K, dropout=0.1, gat_heads=H)
    m.load_state_dict({k: t(v) for k, v in synth.gat_seq_params(D, D, D, D, K, H, seed=777).items()})
    m = m.to(dev).eval()
    x, ea, ins = [t(a, device=dev) for a in (synth.normal((N, D), 1), synth.normal((E, D), 2), synth.normal((K, B, D), 3))]
    ei, b = t(gb.edge_index, device=dev), t(gb.batch, device=dev)
    with torch.no_grad():
        fused = m(x, ei, ea, ins, b)
    out = m(x, ei, ea, ins, b)                       # parameters require grad -> differentiable path
    assert out.requires_grad and maxabs(out, fused) < 1e-4 * (1.0 + float(fused.abs().max()))
    out.square().mean().backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert float(m.convs[0].lin_l.weight.grad.abs().max()) > 0


def test_randomized_forward_and_input_gradient_vs_oracle(dev):
    """40 random batches -- empty graphs in the batch, nodes without in-edges (no self loops guaranteed), hubs taking
    half of a graph's edges, C in {8, 12, 32, 100}, H in {1, 2, 4}, K in 1..3 -- forward (fused kernels) and dL/dx
    (differentiable path) against the oracle in fp64."""
    from oracle import ref_torch as R
    from graphvqa_amd.gat_skip import gat_seq
    rng = np.random.default_rng(2024)
    worst_f = worst_g = 0.0
    for case in range(40):
        B, K, H = int(rng.integers(1, 12)), int(rng.integers(1, 4)), int(rng.choice([1, 2, 4]))
        C, De, Di = int(rng.choice([8, 12, 32, 100])), int(rng.choice([4, 8, 20])), int(rng.choice([4, 16]))
        sizes = rng.integers(0, 50, size=B)
        if sizes.sum() == 0:
            sizes[0] = 3
        batch = np.repeat(np.arange(B), sizes).astype(np.int64)
        N, offs = int(sizes.sum()), np.concatenate([[0], np.cumsum(sizes)])
        src, dst = [np.zeros(0, np.int64)], [np.zeros(0, np.int64)]
        for g in range(B):
            n = int(sizes[g])
            if n == 0:
                continue
            e = int(rng.integers(0, 4 * n + 1))
            s_, d_ = rng.integers(0, n, size=e) + offs[g], rng.integers(0, n, size=e) + offs[g]
            if rng.random() < 0.3 and n > 2:
                d_[: e // 2] = offs[g]
            src.append(s_); dst.append(d_)
        ei = np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)
        E = ei.shape[1]
        p = synth.gat_seq_params(C, C, De, Di, K, H, seed=case)
        for k in list(p):
            if "running_var" in k:
                p[k] = (0.5 + rng.random(p[k].shape)).astype(np.float32)
            elif "running_mean" in k or k.endswith("bias"):
                p[k] = (0.2 * rng.standard_normal(p[k].shape)).astype(np.float32)
        x, ea = rng.standard_normal((N, C)).astype(np.float32), rng.standard_normal((E, De)).astype(np.float32)
        ins, w = rng.standard_normal((K, B, Di)).astype(np.float32), rng.standard_normal((N, C)).astype(np.float32)
        m = gat_seq(C, C, De, Di, K, dropout=0.0, gat_heads=H)
        m.load_state_dict({k: t(v) for k, v in p.items()})
        m = m.to(dev).eval()
        args = [t(a, device=dev) for a in (x, ei, ea, ins, batch)]
        with torch.no_grad():
            out = m(*args)
        rp = {k: (v.double() if v.is_floating_point() else v) for k, v in tparams(p).items()}
        rx = t(x).double().requires_grad_(True)
        ref = R.gat_seq(rx, t(ei), t(ea).double(), t(ins).double(), t(batch), rp, heads=H)
        xg = args[0].clone().requires_grad_(True)
        (m(xg, *args[1:]) * t(w, device=dev)).sum().backward()
        (ref * t(w).double()).sum().backward()
        worst_f = max(worst_f, maxabs(out, ref))
        worst_g = max(worst_g, maxabs(xg.grad, rx.grad) / (float(rx.grad.abs().max()) + 1e-9))
    assert worst_f < 1e-4 and worst_g < 1e-4, (worst_f, worst_g)


@pytest.mark.parametrize("R,D,J,ld", [(1000, 300, 8, 300), (5000, 512, 20, 512), (33, 36, 3, 40), (257, 512, 32, 512), (0, 64, 4, 64),
                                      (4097, 128, 1, 128), (70000, 512, 8, 512)])
def test_tall_skinny_products_vs_fp64(dev, R, D, J, ld):
    """csrc/train.hip: Y = X V, dV = X^T G, dX = addend + G V^T against fp64 products (the logit products of the
    differentiable path and their autograd); X with a row stride, ragged R / D / J, the empty case."""
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import _SkinnyLinear, skinny_linear
    g = torch.Generator().manual_seed(R + D + J)
    Xfull = torch.randn((max(R, 1), ld), generator=g).to(dev)
    X = Xfull[:R, :D]
    V = torch.randn((D, J), generator=g).to(dev).requires_grad_(True)
    Xg = X.detach().clone().requires_grad_(True) if ld == D else None
    Xin = Xg if Xg is not None else X
    assert _SkinnyLinear.supported(Xin, V)
    Y = skinny_linear(Xin, V)
    ref = X.double() @ V.detach().double()
    scale = max(float(ref.abs().max()), 1.0) if R else 1.0
    assert Y.shape == (R, J)
    if R:
        assert float((Y.detach().double() - ref).abs().max()) <= 2e-6 * scale
    G = torch.randn((R, J), generator=g).to(dev)
    Y.backward(G)
    refV = X.double().t() @ G.double()
    sV = max(float(refV.abs().max()), 1.0)
    assert float((V.grad.double() - refV).abs().max()) <= 2e-6 * sV
    if Xg is not None and R:
        refX = G.double() @ V.detach().double().t()
        assert float((Xg.grad.double() - refX).abs().max()) <= 2e-6 * max(float(refX.abs().max()), 1.0)
    if R:       # the fused addend of the C entry and a strided destination
        lib = _lib.load()
        add = torch.randn((R, ld), generator=g).to(dev)
        out = torch.full((R, ld), 7.0, device=dev)
        Vc = V.detach().contiguous()
        _lib.check(lib.gvqa_skinny_backward_input(R, D, J, G.data_ptr(), Vc.data_ptr(), add.data_ptr(), ld, out.data_ptr(), ld,
                                                  torch.cuda.current_stream().cuda_stream))
        refX = add[:, :D].double() + G.double() @ Vc.double().t()
        assert float((out[:, :D].double() - refX).abs().max()) <= 2e-6 * max(float(refX.abs().max()), 1.0)
        if ld > D:
            assert bool((out[:, D:] == 7.0).all())          # nothing written past D


def test_edge_logits_of_all_hops_share_one_pass(dev):
    """The differentiable path computes the edge logits of all K hops in ONE tall-skinny product (and one in the backward): its
    lin_e / att_e gradients still match autograd through the oracle (covered by the gradient tests above); here: more columns than
    one kernel call takes (K * H > 32) are split into column groups."""
    from graphvqa_amd.gat_skip import skinny_linear
    g = torch.Generator().manual_seed(5)
    X = torch.randn((3000, 64), generator=g).to(dev)
    V = torch.randn((64, 40), generator=g).to(dev).requires_grad_(True)
    Y = skinny_linear(X, V)
    assert float((Y.detach().double() - X.double() @ V.detach().double()).abs().max()) <= 2e-5
    Y.square().sum().backward()
    ref = X.double().t() @ (2 * (X.double() @ V.detach().double()))
    assert float((V.grad.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("C,H,with_mask,hub", [(32, 4, False, False), (12, 4, True, False), (300, 4, True, True), (512, 8, True, False),
                                               (64, 1, True, True), (36, 3, False, True), (128, 2, True, False)])
def test_message_passing_with_per_graph_rows_kept_out_of_xp(dev, C, H, with_mask, hub):
    """gat_message_passing(xp, ..., graph_rows=R) == MP(xp + R[batch]) in value and in every gradient (xp, logits, R) -- the
    instruction half of lin_l folded out of the [N, H*C] tensor -- against fp64 autograd through the oracle's restatement of the
    explicit form; with attention dropout masks (s != 1: the logits feel R) and without; hub nodes beyond the tiled backward."""
    from graphvqa_amd.gat_skip import gat_message_passing
    from graphvqa_amd.graph import SceneGraphBatch
    gb = synth.make_graph_batch(7, seed=0xC00 + C, nodes_lo=1, nodes_hi=40, rel_per_node=2.0)
    ei = gb.edge_index.copy()
    if hub:
        n0 = int((gb.batch == 0).sum())
        ei = np.concatenate([ei, np.stack([np.arange(70) % n0, np.zeros(70, np.int64)])], axis=1)
    ei = ei[:, ei[1] % 5 != 3]                   # some nodes lose every in-edge (self loops included): their output is exactly 0
    N, E, B = gb.num_nodes, ei.shape[1], gb.num_graphs
    rng = np.random.default_rng(C * 11 + H)
    xp = rng.standard_normal((N, H * C)).astype(np.float32)
    a_node = rng.standard_normal((N, 2 * H)).astype(np.float32)
    a_edge = rng.standard_normal((E, H)).astype(np.float32)
    rows = rng.standard_normal((B, H * C)).astype(np.float32)
    mask = ((rng.random((E, H)) > 0.3) / 0.7).astype(np.float32) if with_mask else None
    w = rng.standard_normal((N, C)).astype(np.float32)
    g = SceneGraphBatch(t(ei, device=dev), t(gb.batch, device=dev), N, B)
    bias, skip = rng.standard_normal(C).astype(np.float32), rng.standard_normal((N, C)).astype(np.float32)
    xs = [t(a, device=dev).requires_grad_(True) for a in (xp, a_node, a_edge, rows, bias, skip)]
    out, alpha = gat_message_passing(xs[0], xs[1], xs[2], g, H, C, 0.2, None if mask is None else t(mask, device=dev), graph_rows=xs[3],
                                     bias=xs[4], skip=xs[5])
    (out * t(w, device=dev)).sum().backward()
    rs = [t(a).double().requires_grad_(True) for a in (xp, a_node, a_edge, rows, bias, skip)]
    ref_out, ref_alpha = _mp_reference(rs[0] + rs[3][t(gb.batch).long()], rs[1], rs[2], None if mask is None else t(mask).double(),
                                       t(ei), N, H, C, 0.2)
    ref_out = ref_out + rs[4] + rs[5]
    (ref_out * t(w).double()).sum().backward()
    assert maxabs(out, ref_out) < 2e-5 and maxabs(alpha, ref_alpha) < 1e-6
    for got, ref, name in zip(xs, rs, ("dxp", "da_node", "da_edge", "d_graph_rows", "d_bias", "d_skip")):
        assert _rel(got.grad, ref.grad) < 2e-5, name
    # bias / skip without per-graph rows
    ys = [t(a, device=dev).requires_grad_(True) for a in (xp, bias, skip)]
    out2, _ = gat_message_passing(ys[0], xs[1].detach(), xs[2].detach(), g, H, C, 0.2, None, bias=ys[1], skip=ys[2])
    (out2 * t(w, device=dev)).sum().backward()
    qs = [t(a).double().requires_grad_(True) for a in (xp, bias, skip)]
    ref2, _ = _mp_reference(qs[0], rs[1].detach(), rs[2].detach(), None, t(ei), N, H, C, 0.2)
    ((ref2 + qs[1] + qs[2]) * t(w).double()).sum().backward()
    assert maxabs(out2, ref2 + qs[1] + qs[2]) < 2e-5
    for got, ref, name in zip(ys, qs, ("dxp", "d_bias", "d_skip")):
        assert _rel(got.grad, ref.grad) < 2e-5, name


@pytest.mark.parametrize("R,M,N,ldx", [(5000, 256, 128, 256), (65536, 2048, 512, 2048), (1000, 36, 300, 40), (63, 512, 512, 512),
                                       (70001, 1200, 300, 1200), (0, 64, 32, 64)])
@pytest.mark.parametrize("direct", [1, 0])
def test_transposed_split_product_vs_fp64(dev, R, M, N, ldx, direct):
    """gvqa_linear_tn_split2h: C = X^T Y (the weight gradient of the hop projection) against an fp64 product, next to torch's own
    fp32 matmul on the same operands; ragged R / M / N, a strided X, columns of very different magnitude, the empty case.
    direct = 1: the product transposes the row-major operands itself (tn_direct.hip, the default); 0: transposed packs in HBM first."""
    from graphvqa_amd import _lib
    lib = _lib.load()
    old_direct = _lib.set_option(_lib.OPT_TN_DIRECT, direct)
    g = torch.Generator().manual_seed(R + M + N)
    X = torch.randn((max(R, 1), ldx), generator=g).to(dev)[:R]
    Y = torch.randn((max(R, 1), N), generator=g).to(dev)[:R]
    if R:
        X[:, : M // 2] *= 1e-3                       # half of the columns far below the operand's largest magnitude
        Y[:, ::3] *= 50.0
    C_ = torch.full((M, N), 3.0, device=dev)
    ws = torch.empty(max(lib.gvqa_linear_tn_workspace_bytes(R, M, N), 256), dtype=torch.uint8, device=dev)
    try:
        _lib.check(lib.gvqa_linear_tn_split2h(R, M, N, X.data_ptr(), ldx, Y.data_ptr(), N, None, 0, None, 0, C_.data_ptr(), N, ws.data_ptr(), ws.numel(),
                                              torch.cuda.current_stream().cuda_stream))
    finally:
        _lib.set_option(_lib.OPT_TN_DIRECT, old_direct)
    ref = X[:, :M].double().t() @ Y.double()
    if R == 0:
        assert bool((C_ == 0).all())
        return
    # error relative to what the sum can resolve: |X|^T |Y| (the bound fp32 matmul itself is held to)
    bound = (X[:, :M].double().abs().t() @ Y.double().abs())
    err = float(((C_.double() - ref).abs() / bound.clamp_min(1e-30)).max())
    err_torch = float((((X[:, :M].t() @ Y).double() - ref).abs() / bound.clamp_min(1e-30)).max())
    assert err <= max(2e-6, 2.0 * err_torch), (err, err_torch)
    assert float((C_.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_message_passing_backward_reports_the_largest_gradient_magnitude(dev):
    """gvqa_gat_mp_bwd_desc.dxp_absmax: the slices' maximum equals max|dxp| exactly, and the weight-gradient product fed with it
    equals the one that measures the operand itself."""
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_message_passing, _ProjectionLinear
    from graphvqa_amd.graph import SceneGraphBatch
    gb = synth.make_graph_batch(40, seed=77, nodes_lo=5, nodes_hi=40, rel_per_node=2.0)
    N, E, H, C = gb.num_nodes, gb.num_edges, 4, 64
    rng = np.random.default_rng(3)
    g = SceneGraphBatch(t(gb.edge_index, device=dev), t(gb.batch, device=dev), N, gb.num_graphs)
    xp = t(rng.standard_normal((N, H * C)).astype(np.float32), device=dev).requires_grad_(True)
    seen = {}
    xp.register_hook(lambda gr: seen.setdefault("g", gr))
    out, _ = gat_message_passing(xp, t(rng.standard_normal((N, 2 * H)).astype(np.float32), device=dev),
                                 t(rng.standard_normal((E, H)).astype(np.float32), device=dev), g, H, C)
    (out * t(rng.standard_normal((N, C)).astype(np.float32), device=dev)).sum().backward()
    from graphvqa_amd.gat_skip import _absmax_hint
    am = _absmax_hint(seen["g"])          # (the hint rides with the tensor's storage address and version counter: ADVICE r03)
    assert am is not None and am.numel() == _lib.ABSMAX_SLOTS
    assert float(am.max()) == float(xp.grad.abs().max())
    x = t(rng.standard_normal((N, 128)).astype(np.float32), device=dev)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        with_hint = _ProjectionLinear._weight_grad(seen["g"], x)
        without = _ProjectionLinear._weight_grad(seen["g"].clone(), x)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    assert torch.equal(with_hint, without)
    ref = seen["g"].double().t() @ x.double()
    assert float((with_hint.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    # a stale hint is not honoured: after an in-place change of the gradient tensor (a second consumer's accumulation, a hook)
    # the version counter differs, the consumer measures the operand itself and the product stays right -- with the old hint the
    # fp16 operand would overflow
    gbig = seen["g"]
    gbig.mul_(4096.0)
    assert _absmax_hint(gbig) is None
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        big = _ProjectionLinear._weight_grad(gbig, x)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    assert torch.isfinite(big).all() and float((big.double() - 4096.0 * ref).abs().max()) <= 2e-6 * 4096.0 * float(ref.abs().max())


@pytest.mark.parametrize("H,C,Kin,two", [(4, 512, 1024, True), (4, 300, 812, True), (1, 30, 17, False), (8, 64, 100, True), (2, 5, 3, False)])
def test_fold_attention_vs_fp64_einsum(dev, H, C, Kin, two):
    """gvqa_fold_attention_forward / _backward against the fp64 einsum they replace: V = att folded through W, dW, datt."""
    from graphvqa_amd.gat_skip import fold_attention
    g = torch.Generator().manual_seed(H * 1000 + C + Kin)
    W = torch.randn((H * C, Kin), generator=g).to(dev).requires_grad_(True)
    a = torch.randn((1, H, C), generator=g).to(dev).requires_grad_(True)
    b = torch.randn((1, H, C), generator=g).to(dev).requires_grad_(True) if two else None
    G = torch.randn((Kin, H * (2 if two else 1)), generator=g).to(dev)
    V = fold_attention(W, a, b, H)
    (V * G).sum().backward()
    Wd, ad = W.detach().double().requires_grad_(True), a.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if two else None
    ref = torch.einsum("hck,hc->kh", Wd.view(H, C, Kin), ad.view(H, C))
    if two:
        ref = torch.cat((ref, torch.einsum("hck,hc->kh", Wd.view(H, C, Kin), bd.view(H, C))), dim=1)
    (ref * G.double()).sum().backward()
    assert _rel(V.detach(), ref) < 2e-6 and _rel(W.grad, Wd.grad) < 2e-6 and _rel(a.grad, ad.grad) < 2e-6
    assert a.grad.shape == a.shape
    if two:
        assert _rel(b.grad, bd.grad) < 2e-6


@pytest.mark.parametrize("R,M,K,ldw,which", [(5000, 256, 128, 128, "both"), (65536, 2048, 512, 1024, "both"), (1000, 36, 300, 300, "both"),
                                             (63, 512, 512, 512, "dx"), (70001, 1200, 300, 812, "dw"), (4097, 100, 64, 64, "both")])
@pytest.mark.parametrize("direct", [1, 0])
def test_projection_backward_in_one_call_vs_fp64(dev, R, M, K, ldw, which, direct):
    """gvqa_linear_backward_split2h: dx = dy W and dW = dy^T x from ONE pass over dy, against fp64 products and next to torch's fp32
    matmuls; W as a column slice of a wider weight (row stride), ragged sizes, rows of dy of very different magnitude (the packed
    rows share the operand's one scale), either gradient alone."""
    from graphvqa_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(R + M + K)
    dy = torch.randn((R, M), generator=g).to(dev)
    dy[::7] *= 1e-3
    dy[::11] *= 30.0
    W = torch.randn((M, ldw), generator=g).to(dev)[:, :K]
    x = torch.randn((R, K), generator=g).to(dev)
    dx = torch.full((R, K), 5.0, device=dev) if which in ("both", "dx") else None
    dW = torch.full((M, K), 5.0, device=dev) if which in ("both", "dw") else None
    ws = torch.empty(lib.gvqa_linear_backward_workspace_bytes(R, M, K), dtype=torch.uint8, device=dev)
    p = lambda a: None if a is None else a.data_ptr()
    old_direct = _lib.set_option(_lib.OPT_TN_DIRECT, direct)        # (1: dW reads dy and x as they are -- tn_direct.hip; 0: transposed packs first)
    try:
        _lib.check(lib.gvqa_linear_backward_split2h(R, M, K, dy.data_ptr(), M, W.data_ptr(), ldw, x.data_ptr(), K, None, 0, p(dx), K, 0, p(dW), K,
                                                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    finally:
        _lib.set_option(_lib.OPT_TN_DIRECT, old_direct)
    if dx is not None:
        ref = dy.double() @ W.double()
        bound = dy.double().abs() @ W.double().abs()
        err = float(((dx.double() - ref).abs() / bound.clamp_min(1e-30)).max())
        err_t = float((((dy @ W).double() - ref).abs() / bound.clamp_min(1e-30)).max())
        assert err <= max(2e-6, 2.0 * err_t), (err, err_t)
    if dW is not None:
        ref = dy.double().t() @ x.double()
        bound = dy.double().abs().t() @ x.double().abs()
        err = float(((dW.double() - ref).abs() / bound.clamp_min(1e-30)).max())
        err_t = float((((dy.t() @ x).double() - ref).abs() / bound.clamp_min(1e-30)).max())
        assert err <= max(2e-6, 2.0 * err_t), (err, err_t)
    if dx is not None:       # dx_accumulate: the product lands on top of what dx holds (the logit products' input gradient in the hop)
        add = torch.randn((R, K), generator=g).to(dev)
        acc = add.clone()
        _lib.check(lib.gvqa_linear_backward_split2h(R, M, K, dy.data_ptr(), M, W.data_ptr(), ldw, x.data_ptr(), K, None, 0, acc.data_ptr(), K, 1,
                                                    None, K, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
        assert float((acc - (add + dx)).abs().max()) <= 1e-6 * max(float(dx.abs().max()), 1.0)


def test_in_kernel_dropout_draws_equal_their_explicit_mask(dev):
    """gvqa_bn_relu_dropout_train_*_rng (feature dropout of gat_skip.py:276 drawn inside the BatchNorm passes, Philox keyed on (seed, offset)):
    forward and backward equal, bit for bit, the explicit-mask entry points fed with gvqa_dropout_keep_mask's bytes; the kept fraction is 1 - p;
    the same (seed, offset) reproduces, another offset does not; and the module path reserves its counters on torch's generator, so
    torch.manual_seed reproduces a training forward."""
    from graphvqa_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(11)
    N, Cc, p, seed, off = 3001, 300, 0.1, 0x1234567890ABCDEF, 4096
    x, dy = torch.randn((N, Cc), generator=g).to(dev), torch.randn((N, Cc), generator=g).to(dev)
    w, b = (torch.rand(Cc, generator=g) + 0.5).to(dev), torch.randn(Cc, generator=g).to(dev)
    keep = torch.empty((N, Cc), dtype=torch.uint8, device=dev)
    _lib.check(lib.gvqa_dropout_keep_mask(N, Cc, seed, off, p, keep.data_ptr(), st))
    frac = float(keep.float().mean())
    assert abs(frac - (1 - p)) < 4e-3 and set(keep.unique().tolist()) <= {0, 1}
    keep2 = torch.empty_like(keep)
    _lib.check(lib.gvqa_dropout_keep_mask(N, Cc, seed, off, p, keep2.data_ptr(), st))
    assert torch.equal(keep, keep2)
    _lib.check(lib.gvqa_dropout_keep_mask(N, Cc, seed, off + 4, p, keep2.data_ptr(), st))
    assert not torch.equal(keep, keep2)
    ws = torch.empty(lib.gvqa_bn_train_workspace_bytes(N, Cc), dtype=torch.uint8, device=dev)
    outs = []
    for rng in (True, False):
        y, mean, var = torch.empty_like(x), torch.empty_like(w), torch.empty_like(w)
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(w)
        if rng:
            _lib.check(lib.gvqa_bn_relu_dropout_train_forward_rng(N, Cc, x.data_ptr(), w.data_ptr(), b.data_ptr(), 1e-5, seed, off, p, y.data_ptr(),
                                                                  mean.data_ptr(), var.data_ptr(), ws.data_ptr(), ws.numel(), st))
            _lib.check(lib.gvqa_bn_relu_dropout_train_backward_rng(N, Cc, x.data_ptr(), w.data_ptr(), b.data_ptr(), mean.data_ptr(), var.data_ptr(), 1e-5,
                                                                   seed, off, p, dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                                   ws.data_ptr(), ws.numel(), st))
        else:
            ks = 1.0 / (1.0 - p)
            _lib.check(lib.gvqa_bn_relu_dropout_train_forward(N, Cc, x.data_ptr(), w.data_ptr(), b.data_ptr(), 1e-5, keep.data_ptr(), ks, y.data_ptr(),
                                                              mean.data_ptr(), var.data_ptr(), ws.data_ptr(), ws.numel(), st))
            _lib.check(lib.gvqa_bn_relu_dropout_train_backward(N, Cc, x.data_ptr(), w.data_ptr(), b.data_ptr(), mean.data_ptr(), var.data_ptr(), 1e-5,
                                                               keep.data_ptr(), ks, dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                               ws.data_ptr(), ws.numel(), st))
        outs.append((y, dx, dw, db))
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)
    assert float((outs[0][0] == 0).float().mean()) > p * 0.9          # (dropped or clipped by the ReLU)
    # the attention-dropout mask (gvqa_dropout_scale_mask): the same decisions as floats 0 | 1 / (1 - p)
    fm = torch.empty(N * Cc, device=dev)
    _lib.check(lib.gvqa_dropout_scale_mask(N * Cc, seed, off, p, fm.data_ptr(), st))
    assert torch.equal(fm.view(N, Cc) > 0, keep > 0) and float((fm[fm > 0] - 1.0 / (1.0 - p)).abs().max()) < 1e-6
    # module path: the generator's counters are reserved, the seed reproduces
    from graphvqa_amd.gat_skip import _bn_relu_train
    bn = torch.nn.BatchNorm1d(Cc).to(dev).train()
    xm = x[:3000].contiguous()
    torch.manual_seed(77); o0 = torch.cuda.default_generators[dev.index or 0].get_offset(); y1 = _bn_relu_train(bn, xm, p)
    assert torch.cuda.default_generators[dev.index or 0].get_offset() > o0
    y2 = _bn_relu_train(bn, xm, p)
    torch.manual_seed(77); y3 = _bn_relu_train(bn, xm, p)
    assert torch.equal(y1, y3) and not torch.equal(y1, y2)


def test_operand_pack_leaves_slice_maxima_and_the_backward_takes_them(dev):
    """gvqa_split2h_pack_absmax: the packed operand is the one gvqa_split2h_pack writes and the slice maxima's maximum is max|x| exactly;
    gvqa_linear_backward_split2h_hint fed with them returns the same dW, bit for bit, as the call that measures x itself."""
    from graphvqa_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    R, M, K = 5000, 256, 132
    x = torch.randn((R, K), generator=g).to(dev)
    x[::13] *= 40.0
    dy = torch.randn((R, M), generator=g).to(dev)
    W = torch.randn((M, K), generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.gvqa_split2h_packed_bytes(R, K)
    p0, p1 = torch.zeros(nb, dtype=torch.uint8, device=dev), torch.zeros(nb, dtype=torch.uint8, device=dev)
    am = torch.full((_lib.ABSMAX_SLOTS,), 7.0, device=dev)
    _lib.check(lib.gvqa_split2h_pack(R, K, x.data_ptr(), K, p0.data_ptr(), st))
    _lib.check(lib.gvqa_split2h_pack_absmax(R, K, x.data_ptr(), K, p1.data_ptr(), am.data_ptr(), st))
    assert torch.equal(p0, p1)
    assert float(am.max()) == float(x.abs().max())
    # gvqa_split2h_pack_logits: the same packed operand and maxima, plus x Vn^T (J = 8) from the same pass
    Vn = torch.randn((8, K), generator=g).to(dev)
    p2, am2, lg = torch.zeros(nb, dtype=torch.uint8, device=dev), torch.full((_lib.ABSMAX_SLOTS,), 7.0, device=dev), torch.empty((R, 8), device=dev)
    _lib.check(lib.gvqa_split2h_pack_logits(R, K, x.data_ptr(), K, p2.data_ptr(), am2.data_ptr(), Vn.data_ptr(), 8, lg.data_ptr(), st))
    assert torch.equal(p0, p2) and torch.equal(am, am2)
    refl = x.double() @ Vn.double().t()
    assert float((lg.double() - refl).abs().max()) <= 2e-6 * float(refl.abs().max()) * K ** 0.5
    ws = torch.empty(lib.gvqa_linear_backward_workspace_bytes(R, M, K), dtype=torch.uint8, device=dev)
    outs = []
    for hint in (None, am):
        dW, dx = torch.empty((M, K), device=dev), torch.empty((R, K), device=dev)
        _lib.check(lib.gvqa_linear_backward_split2h_hint(R, M, K, dy.data_ptr(), M, W.data_ptr(), K, x.data_ptr(), K, None, 0,
                                                         None if hint is None else hint.data_ptr(), 0 if hint is None else hint.numel(),
                                                         dx.data_ptr(), K, 0, dW.data_ptr(), K, ws.data_ptr(), ws.numel(), st))
        outs.append((dW, dx))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = dy.double().t() @ x.double()
    assert float((outs[1][0].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    # dy as a column block of a wider buffer (row stride > M): the direct products' descriptors cover exactly the rows' extents
    wide = torch.full((R, M + 48), float("nan"), device=dev)
    wide[:, :M] = dy
    dW3, dx3 = torch.empty((M, K), device=dev), torch.empty((R, K), device=dev)
    _lib.check(lib.gvqa_linear_backward_split2h_hint(R, M, K, wide.data_ptr(), M + 48, W.data_ptr(), K, x.data_ptr(), K, None, 0, am.data_ptr(), am.numel(),
                                                     dx3.data_ptr(), K, 0, dW3.data_ptr(), K, ws.data_ptr(), ws.numel(), st))
    assert torch.equal(dW3, outs[1][0]) and torch.equal(dx3, outs[1][1])
    # gvqa_linear_backward_split2h_ex: dx = dy W + g v^T + addend in the product's epilogue (M % 16 == 0: the direct product applies), against the
    # separate passes; the packed form refuses the epilogue terms before launching anything
    J = 8
    gl, vl, add = torch.randn((R, J), generator=g).to(dev), torch.randn((K, J), generator=g).to(dev), torch.randn((R, K), generator=g).to(dev)
    ex = _lib.LinearBackwardExtras()
    ex.lowrank_g, ex.lowrank_v, ex.J, ex.addend, ex.ld_addend = gl.data_ptr(), vl.data_ptr(), J, add.data_ptr(), K
    dx2 = torch.empty((R, K), device=dev)
    _lib.check(lib.gvqa_linear_backward_split2h_ex(R, M, K, dy.data_ptr(), M, W.data_ptr(), K, x.data_ptr(), K, None, 0, dx2.data_ptr(), K, 0, None, K,
                                                   C.byref(ex), ws.data_ptr(), ws.numel(), st))
    want = outs[0][1].double() + gl.double() @ vl.double().t() + add.double()
    assert float((dx2.double() - want).abs().max()) <= 1e-5 * max(float(want.abs().max()), 1.0)
    old = _lib.set_option(_lib.OPT_TN_DIRECT, 0)
    try:
        rc = lib.gvqa_linear_backward_split2h_ex(R, M, K, dy.data_ptr(), M, W.data_ptr(), K, x.data_ptr(), K, None, 0, dx2.data_ptr(), K, 0, None, K,
                                                 C.byref(ex), ws.data_ptr(), ws.numel(), st)
    finally:
        _lib.set_option(_lib.OPT_TN_DIRECT, old)
    assert rc == _lib.E_UNSUPPORTED


@pytest.mark.parametrize("train", [False, True])
def test_gat_seq_gradients_on_the_library_products(dev, train):
    """The gradient checks above run below the size at which the projection leaves torch (GVQA_OPT_SPLIT3_MIN_MFLOP); here the
    threshold is 0, so the forward projection, dx = dy W and dW = dy^T x (gvqa_linear_backward_split2h), the tall-skinny logit
    products and the attention folds are ALL the library's kernels -- at the small dims and at the reference's real widths
    (d = 300, ins 512, H = 4, 60 ragged graphs) -- and every parameter / input gradient is held to the oracle's fp64 autograd."""
    from graphvqa_amd import _lib
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        _grads_vs_oracle(dev, train=train)
        _grads_vs_oracle(dev, train=train, dims=(300, 300, 512, 2, 4), seed=9, graphs=60)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)


def test_gat_seq_gradients_with_dropout_masks_on_the_library_products(dev):
    """The same with attention and feature dropout masks (s != 1: the per-graph rows reach the logits' gradient)."""
    from graphvqa_amd import _lib
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        test_gat_seq_gradients_with_dropout_masks(dev)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)


@pytest.mark.parametrize("ab", [1, 2, 4, 8, 16, 64, 111])
def test_gat_seq_gradients_with_each_training_fusion_switched(dev, ab, monkeypatch):
    """Every alternative form of the differentiable path (gat_skip._TRAIN_AB: 1 |h| maxima measured by the backward, 2 the skip's gradient through
    autograd, 4 head rows / bias / skip as a second pass, 8 tiny per-graph products on the tiled kernel, 16 dh written by ONE epilogue
    (gvqa_linear_backward_split2h_ex's rank-J + addend terms), 64 node logits as a pass of their own, 111 = round 4's path) against the oracle's fp64 autograd, with and without dropout
    masks, at the reference's widths -- the default forms are the tests above."""
    from graphvqa_amd import _lib, gat_skip
    monkeypatch.setattr(gat_skip, "_TRAIN_AB", ab)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        test_gat_seq_gradients_with_dropout_masks(dev)
        _grads_vs_oracle(dev, train=True, dims=(300, 300, 512, 2, 4), seed=9, graphs=60)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)


def test_lcgn_seq_gradients_on_the_library_products(dev):
    """lcgn_seq's node-sized layers (init / proj_x_loc / proj_x_ctx / the stacked lin_l|lin_r|cal_x with its iteration-invariant
    x_loc block / output_layer / fin_layer, lcgn.py:305-322) forced onto the library's products and one-call backward."""
    from graphvqa_amd import _lib
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        test_lcgn_seq_gradients_vs_oracle(dev)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)


@pytest.mark.parametrize("threshold", [None, 0])
def test_gat_seq_gradients_at_widths_the_library_products_do_not_take(dev, threshold):
    """Widths that are not multiples of 4 (d = 30, ins 22, H = 2): every product of the differentiable path takes its torch
    form, the message passing and the per-graph rows stay on the kernels; all gradients against the oracle's fp64 autograd."""
    from graphvqa_amd import _lib
    old = None if threshold is None else _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, threshold)
    try:
        _grads_vs_oracle(dev, train=True, dims=(30, 18, 22, 2, 2), seed=13, graphs=9)
        _grads_vs_oracle(dev, train=False, dims=(36, 20, 24, 3, 1), seed=14, graphs=9)
    finally:
        if old is not None:
            _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)


@pytest.mark.parametrize("N,C", [(257, 64), (1000, 300), (4097, 512), (33, 30)])
def test_bn_relu_with_feature_dropout_in_the_same_passes(dev, N, C):
    """_BatchNormReluTrain with a keep mask: y = relu(bn(x)) * keep / (1 - p) and its gradients (x, weight, bias) against torch
    autograd applying the same mask; widths with and without the 16-byte form."""
    from graphvqa_amd.gat_skip import _BatchNormReluTrain
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn((N, C), generator=g).to(dev).requires_grad_(True)
    w = (1.0 + 0.3 * torch.randn(C, generator=g)).to(dev).requires_grad_(True)
    b = (0.2 * torch.randn(C, generator=g)).to(dev).requires_grad_(True)
    keep = (torch.rand((N, C), generator=g) > 0.25).to(torch.uint8).to(dev)
    gy = torch.randn((N, C), generator=g).to(dev)
    y, mean, var = _BatchNormReluTrain.apply(x, w, b, 1e-5, keep, 1.0 / 0.75)
    y.backward(gy)
    xr, wr, br = [v.detach().double().requires_grad_(True) for v in (x, w, b)]
    ref = torch.relu(torch.nn.functional.batch_norm(xr, None, None, wr, br, True, 0.0, 1e-5)) * keep.double() / 0.75
    ref.backward(gy.double())
    assert maxabs(y, ref) < 2e-5
    assert maxabs(mean, xr.mean(0)) < 1e-5 and maxabs(var, xr.var(0, unbiased=False)) < 1e-5
    for got, r, name in ((x.grad, xr.grad, "dx"), (w.grad, wr.grad, "dw"), (b.grad, br.grad, "db")):
        assert _rel(got, r) < 2e-5, name


def test_config1_pipeline_end_to_end_gradients_on_the_library_products(dev):
    """The end-to-end gradient check (encoder -> gat_seq -> pooling -> classifier against the oracle's autograd) with the size
    threshold at 0: the encoder's and the pooling head's differentiable paths then run every product on the library's kernels
    (split GEMMs forward, one-call backward, gather-sum / gather-add-relu ops), not torch's."""
    from graphvqa_amd import _lib
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        test_config1_pipeline_end_to_end_gradients(dev)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
