"""CPU oracle for the scene-graph execution path -- TEST INFRASTRUCTURE ONLY.

A dependency-free (plain torch CPU ops) restatement of the reference's
algorithm for the hot path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this file, and only as the checker
or the timed CPU baseline.  Nothing under `graphvqa_amd/` imports it; the product
path fails loudly when the HIP library is missing.

Parity status: PINNED against the reference's own code (gat_skip.py, lcgn.py,
pipeline_model_{gcn,gine}.py imported unmodified from /root/reference in the
build container) through `tests/golden/*.npz` (generator:
`tests/golden/make_golden.py`).  The third-party half of the reference
(torch_geometric / torch_scatter, version unpinned by the reference, PyG
1.6/1.7-era API) is absent from the image, so those semantics are restated from
their documented behaviour (oracle/pyg_shim/README.md) -- parity against the real
PyG binaries is UNPINNED.

All functions take a `params` dict of tensors keyed exactly like the reference
module's state_dict and compute in the dtype of the inputs
(float32 for parity runs, float64 for error analysis).  The op sequence follows
the reference line by line so that fp32 rounding matches its CPU execution as
closely as a restatement can (gather -> elementwise -> sequential scatter-add in
COO edge order).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# third-party primitives, restated (PyG 1.6/1.7 semantics)
# ----------------------------------------------------------------------------
def segment_softmax(src: torch.Tensor, index: torch.Tensor, num_segments: int) -> torch.Tensor:
    """torch_geometric.utils.softmax: exp(src - segmax[index]) / (segsum[index] + 1e-16).

    Called at gat_skip.py:188 / lcgn.py:210 with index = destination node.
    Empty segments take max 0 (torch_scatter fills untouched outputs with 0).
    """
    shape = (num_segments,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    seg_max = torch.full(shape, float("-inf"), dtype=src.dtype)
    seg_max = seg_max.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    seg_max = torch.where(torch.isinf(seg_max), torch.zeros_like(seg_max), seg_max)
    out = (src - seg_max.index_select(0, index)).exp()
    seg_sum = torch.zeros(shape, dtype=src.dtype).scatter_add_(0, idx, out)
    return out / (seg_sum.index_select(0, index) + 1e-16)


def scatter_add_rows(msg: torch.Tensor, index: torch.Tensor, num_rows: int) -> torch.Tensor:
    """MessagePassing 'add' aggregation along node_dim=0 (torch_scatter.scatter_add)."""
    idx = index.view(-1, *([1] * (msg.dim() - 1))).expand_as(msg)
    return torch.zeros((num_rows,) + tuple(msg.shape[1:]), dtype=msg.dtype).scatter_add_(0, idx, msg)


# ----------------------------------------------------------------------------
# gat  (gat_skip.py:16-213)
# ----------------------------------------------------------------------------
def gat_conv(x, edge_index, edge_attr, p, prefix="", heads=4, negative_slope=0.2,
             concat=False, return_attention_weights=False, alpha_mask=None):
    """One `gat` hop, eval mode (attention dropout inactive; gat_skip.py:190).

    x [N, in], edge_attr [E, edge_in]; lin_l is shared with lin_r for int
    in_channels (gat_skip.py:75-77), so x_l == x_r.
    """
    W_l = p[prefix + "lin_l.weight"]
    W_e = p[prefix + "lin_e.weight"]
    att_l, att_r, att_e = p[prefix + "att_l"], p[prefix + "att_r"], p[prefix + "att_e"]
    bias = p.get(prefix + "bias", None)
    H = heads
    C = W_l.shape[0] // H
    N = (x[0] if isinstance(x, (tuple, list)) else x).shape[0]
    src, dst = edge_index[0], edge_index[1]

    if isinstance(x, (tuple, list)):                           # (x_l, x_r): tuple in_channels, separate lin_r (gat_skip.py:78-80,136-143)
        x_l, x_r = x
        N = x_r.shape[0] if x_r is not None else x_l.shape[0]
        xp = F.linear(x_l, W_l).view(-1, H, C)                 # :138
        a_l = (xp * att_l).sum(dim=-1)                         # :139
        a_r = (F.linear(x_r, p[prefix + "lin_r.weight"]).view(-1, H, C) * att_r).sum(dim=-1) if x_r is not None else 0.0 * a_l   # :141-142
    else:
        xp = F.linear(x, W_l).view(-1, H, C)                   # gat_skip.py:133
        a_l = (xp * att_l).sum(dim=-1)                         # :134
        a_r = (xp * att_r).sum(dim=-1)                         # :135
    e = F.linear(edge_attr, W_e).view(-1, H, C)                # :150
    a_e = (e * att_e).sum(dim=-1)                              # :151

    alpha = a_l.index_select(0, src) + a_r.index_select(0, dst)   # :183 (alpha_j + alpha_i)
    alpha = alpha + a_e                                        # :185
    alpha = F.leaky_relu(alpha, negative_slope)                # :187
    alpha = segment_softmax(alpha, dst, N)                     # :188
    alpha_d = alpha if alpha_mask is None else alpha * alpha_mask   # :205 F.dropout(alpha) with a given mask / (1 - p)
    msg = xp.index_select(0, src) * alpha_d.unsqueeze(-1)      # :208
    out = scatter_add_rows(msg, dst, N)                        # aggregate (aggr='add')
    out = out.view(-1, H * C) if concat else out.mean(dim=1)   # :162-165
    if bias is not None:
        out = out + bias                                       # :168
    if return_attention_weights:
        return out, alpha
    return out


def batchnorm_eval(h, p, prefix, eps=1e-5):
    """torch.nn.BatchNorm1d in eval mode (running statistics)."""
    return F.batch_norm(h, p[prefix + "running_mean"], p[prefix + "running_var"],
                        p[prefix + "weight"], p[prefix + "bias"], False, 0.0, eps)


def batchnorm_train(h, p, prefix, eps=1e-5):
    """BatchNorm1d forward in train mode: batch statistics over all N rows, biased variance."""
    return F.batch_norm(h, None, None, p[prefix + "weight"], p[prefix + "bias"], True, 0.0, eps)


def gat_seq(x, edge_index, edge_attr, instr_vectors, batch, p, heads=4, negative_slope=0.2,
            training_bn=False, return_all=False, alpha_masks=None, feature_masks=None):
    """`gat_seq.forward` (gat_skip.py:249-279): K hops of instruction-conditioned GAT with
    skip connection, then BN -> ReLU (-> dropout, inactive) on all but the last hop."""
    K = instr_vectors.shape[0]
    h = x
    hs, alphas = [], []
    for i in range(K):
        ins = instr_vectors[i]
        edge_batch = batch.index_select(0, edge_index[0])                       # :257
        edge_cat = torch.cat((edge_attr, ins.index_select(0, edge_batch)), -1)  # :259-260
        x_cat = torch.cat((h, ins.index_select(0, batch)), -1)                  # :263-264
        conv, alpha = gat_conv(x_cat, edge_index, edge_cat, p, f"convs.{i}.", heads,
                               negative_slope, False, True, None if alpha_masks is None else alpha_masks[i])
        h = conv + h                                                            # :270
        if i != K - 1:                                                          # :273-276
            h = (batchnorm_train if training_bn else batchnorm_eval)(h, p, f"bns.{i}.")
            h = F.relu(h)
            if feature_masks is not None:                                       # :276 F.dropout with a given mask
                h = h * feature_masks[i]
        hs.append(h)
        alphas.append(alpha)
    if return_all:
        return h, hs, alphas
    return h


# ----------------------------------------------------------------------------
# GINE / GCN variants (baseline_and_test_models/pipeline_model_{gine,gcn}.py)
# ----------------------------------------------------------------------------
def gine_conv(x, edge_index, edge_attr, p, prefix="", eps=0.0):
    """PyG GINEConv with nn = Lin -> ReLU -> Lin (pipeline_model_gine.py:628):
    nn((1 + eps) * x_i + sum_{j->i} relu(x_j + e_ji))."""
    src, dst = edge_index[0], edge_index[1]
    msg = F.relu(x.index_select(0, src) + edge_attr)
    out = scatter_add_rows(msg, dst, x.shape[0])
    out = out + (1 + eps) * x
    hid = F.relu(F.linear(out, p[prefix + "nn.0.weight"], p[prefix + "nn.0.bias"]))
    return F.linear(hid, p[prefix + "nn.2.weight"], p[prefix + "nn.2.bias"])


def gcn_norm(edge_index, num_nodes, dtype):
    """PyG gcn_norm with add_remaining_self_loops (unit weights, fill 1)."""
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop = torch.arange(num_nodes, dtype=row.dtype)
    row = torch.cat([row[mask], loop])
    col = torch.cat([col[mask], loop])
    w = torch.ones(row.shape[0], dtype=dtype)
    deg = torch.zeros(num_nodes, dtype=dtype).scatter_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
    return row, col, dis[row] * w * dis[col]


def gcn_conv(x, edge_index, p, prefix=""):
    """PyG 1.6/1.7 GCNConv (pipeline_model_gcn.py:628): D^-1/2 (A~) D^-1/2 (x W) + b,
    weight stored [in, out]."""
    N = x.shape[0]
    row, col, norm = gcn_norm(edge_index, N, x.dtype)
    xw = torch.matmul(x, p[prefix + "weight"])
    out = scatter_add_rows(norm.view(-1, 1) * xw.index_select(0, row), col, N)
    return out + p[prefix + "bias"]


def _bn_relu_chain(x, p, K):
    h = x
    for i in range(K - 1):
        h = F.relu(batchnorm_eval(h, p, f"bns.{i}."))
    return h


def gine_seq(x, edge_index, edge_attr, instr_vectors, batch, p, return_convs=False):
    """`gine_seq.forward` as written (pipeline_model_gine.py:641-674): conv_res is computed and
    DISCARDED (h is never reassigned from it), so the module output is x through 4x(BN, ReLU).
    The per-hop conv results are returned separately for kernel-level parity."""
    K = instr_vectors.shape[0]
    h = x
    convs = []
    for i in range(K):
        ins = instr_vectors[i]
        edge_cat = torch.cat((edge_attr, ins.index_select(0, batch.index_select(0, edge_index[0]))), -1)
        x_cat = torch.cat((h, ins.index_select(0, batch)), -1)
        convs.append(gine_conv(x_cat, edge_index, edge_cat, p, f"convs.{i}."))
        if i != K - 1:
            h = F.relu(batchnorm_eval(h, p, f"bns.{i}."))
    return (h, convs) if return_convs else h


def gcn_seq(x, edge_index, instr_vectors, batch, p, return_convs=False):
    """`gcn_seq.forward` as written (pipeline_model_gcn.py:641-669); same discarded-conv quirk."""
    K = instr_vectors.shape[0]
    h = x
    convs = []
    for i in range(K):
        ins = instr_vectors[i]
        x_cat = torch.cat((h, ins.index_select(0, batch)), -1)
        convs.append(gcn_conv(x_cat, edge_index, p, f"convs.{i}."))
        if i != K - 1:
            h = F.relu(batchnorm_eval(h, p, f"bns.{i}."))
    return (h, convs) if return_convs else h


# ----------------------------------------------------------------------------
# LCGN variant (baseline_and_test_models/lcgn.py)
# ----------------------------------------------------------------------------
def lcgn_conv(x_joint, edge_index, cmd, batch, p, prefix="lcgn.", heads=1, negative_slope=0.2,
              return_attention_weights=False, node_store=None):
    """`gat_lcgn.forward` + `message` (lcgn.py:120-238), eval mode, concat=False.
    node_store: see lcgn_seq (applied to x_l, x_r, x_val and the aggregated message)."""
    st = node_store if node_store is not None else (lambda v: v)
    H = heads
    C = p[prefix + "lin_l.weight"].shape[0] // H
    N = x_joint.shape[0]
    src, dst = edge_index[0], edge_index[1]
    x_l = st(F.linear(x_joint, p[prefix + "lin_l.weight"])).view(-1, H, C)      # :144
    x_r = st(F.linear(x_joint, p[prefix + "lin_r.weight"])).view(-1, H, C)      # :145
    proj_cmd = F.linear(cmd, p[prefix + "proj_cmd.weight"])                 # :148
    cal_cmd = F.linear(cmd, p[prefix + "cal_cmd.weight"])                   # :149
    onehot = F.one_hot(batch).to(x_joint.dtype)                             # :150
    proj_cmd = onehot.matmul(proj_cmd).view(-1, H, C)                       # :152
    cal_cmd = onehot.matmul(cal_cmd).view(-1, H, C)                         # :153
    x_mul = proj_cmd * x_r                                                  # :154
    alpha = torch.sum(x_l.index_select(0, src) * x_mul.index_select(0, dst), dim=-1)  # :207
    alpha = F.leaky_relu(alpha, negative_slope)
    alpha = segment_softmax(alpha, dst, N)
    x_val = st(F.linear(x_joint.index_select(0, src), p[prefix + "cal_x.weight"])).view(-1, H, C)  # :230
    x_fin = x_val * cal_cmd.index_select(0, src)                            # :231
    out = scatter_add_rows(x_fin * alpha.unsqueeze(-1), dst, N).mean(dim=1)  # concat=False
    bias = p.get(prefix + "bias", None)
    if bias is not None:
        out = out + bias
    out = st(out)
    if return_attention_weights:
        return out, alpha
    return out


def lcgn_extract_command(q_emb, lstm_outputs, t, p):
    """`lcgn_seq.extract_textual_command` (lcgn.py:292-300)."""
    lo = lstm_outputs.transpose(1, 0)                                        # [B, L, D]
    q_cmd = F.linear(q_emb, p[f"qInput2_{t}.weight"], p[f"qInput2_{t}.bias"])
    raw = F.linear(q_cmd[:, None, :] * lo, p["cmd_inter2logits.weight"],
                   p["cmd_inter2logits.bias"]).squeeze(-1)
    att = F.softmax(raw, dim=-1)
    return torch.bmm(att[:, None, :], lo).squeeze(1)


def lcgn_seq(x, edge_index, batch, q_encoding, lstm_outputs, p, x_ctx_init, max_iter=4, heads=1,
             return_all=False, node_store=None):
    """`lcgn_seq.forward` (lcgn.py:303-323), eval mode.  `x_ctx_init` stands in for the
    reference's `torch.randn(x_loc.size())` drawn on the CPU generator (lcgn.py:306): the caller
    draws it with the same seed/call so both sides see identical noise.

    node_store (BASELINE config 5, "bf16 node features"): a callable applied to every PER-NODE tensor
    at the point where the reference materialises it -- the input features, x_loc, proj_x_loc(x_loc),
    x_ctx (initial and after every iteration), the proj_x_ctx * proj_x_loc product, gat_lcgn's x_l /
    x_r / x_val and its aggregated message -- e.g. `bf16_storage` below.  Everything else (weights,
    per-question tensors, logits, softmax, accumulation) stays in the dtype of the inputs: run in
    float64 this is "the reference's arithmetic done exactly, on node tensors that are stored in
    bf16", the oracle-side model of the storage choice (the reference itself has no such mode)."""
    st = node_store if node_store is not None else (lambda v: v)
    x = st(x)
    x_loc = st(F.linear(x, p["init_sg_emb_input.0.weight"], p["init_sg_emb_input.0.bias"]))
    x_ctx = st(x_ctx_init)
    q_emb = F.relu(F.linear(q_encoding, p["qInput1.weight"], p["qInput1.bias"]))
    proj_x_loc = st(F.linear(x_loc, p["proj_x_loc.1.weight"], p["proj_x_loc.1.bias"]))
    ctxs = []
    for t in range(max_iter):
        cmd = lcgn_extract_command(q_emb, lstm_outputs, t, p)
        proj_x_ctx = F.linear(x_ctx, p["proj_x_ctx.1.weight"], p["proj_x_ctx.1.bias"])
        x_joint = torch.cat([x_loc, x_ctx, st(proj_x_ctx * proj_x_loc)], dim=-1)
        msg = lcgn_conv(x_joint, edge_index, cmd, batch, p, "lcgn.", heads, node_store=node_store)
        x_ctx = st(F.linear(torch.cat([x_ctx, msg], dim=-1), p["output_layer.weight"],
                            p["output_layer.bias"]))
        ctxs.append(x_ctx)
    out = F.linear(torch.cat([x_loc, x_ctx], dim=-1), p["fin_layer.weight"], p["fin_layer.bias"])
    return (out, ctxs) if return_all else out


def bf16_storage(v):
    """node_store for lcgn_seq: round to bfloat16 (nearest even) and come back in the working dtype."""
    return v.to(torch.bfloat16).to(v.dtype)


# ----------------------------------------------------------------------------
# Step after the path: language-conditioned global attention pooling + classifier
# ----------------------------------------------------------------------------
def _mlp2(x, p, prefix):
    """Seq(Lin, ReLU, Lin) with state_dict keys prefix.0.* / prefix.2.*"""
    h = F.relu(F.linear(x, p[prefix + "0.weight"], p[prefix + "0.bias"]))
    return F.linear(h, p[prefix + "2.weight"], p[prefix + "2.bias"])


def global_attention_pool(x, u, batch, p, num_graphs):
    """`MyConditionalGlobalAttention.forward` (pipeline_model_gat.py:149-181):
    x' = node_nn(x); gate = gate_nn(ques_nn(u)[batch] * x'); softmax over the nodes of each graph;
    out[g] = sum_n gate[n] x'[n]."""
    xn = _mlp2(x, p, "node_nn.")
    gate = _mlp2(_mlp2(u, p, "ques_nn.").index_select(0, batch) * xn, p, "gate_nn.")
    gate = segment_softmax(gate, batch, num_graphs)
    return scatter_add_rows(gate * xn, batch, num_graphs)


def short_answer_logits(g_feat, q, p, prefix="logit_fc."):
    """pipeline_model_gat.py:814-816 with logit_fc = Seq(Dropout, Lin(3Q,512), ELU, Dropout, Lin(512,A))
    (:722-728), eval mode."""
    feat = torch.cat((g_feat, q, g_feat * q), dim=-1)
    h = F.elu(F.linear(feat, p[prefix + "1.weight"], p[prefix + "1.bias"]))
    return F.linear(h, p[prefix + "4.weight"], p[prefix + "4.bias"])


# ----------------------------------------------------------------------------
# Step before the path: ground-truth scene-graph encoder
# ----------------------------------------------------------------------------
def graph_layernorm(x, batch, num_graphs, weight, bias, eps=1e-5):
    """graph_utils/my_graph_layernorm.py:52-78: per-graph mean / variance over nodes x channels;
    out = x / (sqrt(var) + eps) (eps OUTSIDE the sqrt), then * weight + bias (1-element tensors)."""
    C = x.shape[-1]
    cnt = torch.zeros(num_graphs, dtype=x.dtype).scatter_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    norm = cnt.clamp(min=1).mul(C).view(-1, 1)
    mean = scatter_add_rows(x, batch, num_graphs).sum(dim=-1, keepdim=True) / norm
    x = x - mean.index_select(0, batch)
    var = scatter_add_rows(x * x, batch, num_graphs).sum(dim=-1, keepdim=True) / norm
    out = x / (var.sqrt().index_select(0, batch) + eps)
    return out * weight + bias


def scene_graph_encoder(x_tokens, edge_index, edge_tokens, added_sym_edge, batch, num_graphs, p):
    """`GroundTruth_SceneGraph_Encoder.forward` (pipeline_model_gat.py:575-610) with its MetaLayer
    (EdgeModel :65-76, NodeModel :78-98, torch_scatter.scatter_mean :96).  Returns
    (x_encoded [N, D], edge_attr_encoded [E, D])."""
    emb = p["sg_vocab_embedding.weight"]
    x = emb.index_select(0, x_tokens.reshape(-1)).view(x_tokens.shape[0], x_tokens.shape[1], -1).sum(dim=-2)
    e = emb.index_select(0, edge_tokens.reshape(-1)).view(edge_tokens.shape[0], edge_tokens.shape[1], -1).clone()
    e[added_sym_edge] = e[added_sym_edge] * -1                                   # :590
    e = e.sum(dim=-2)
    src, dst = edge_index[0], edge_index[1]
    pre = "scene_graph_encoding_layer."
    e2 = _mlp2(torch.cat([x.index_select(0, src), x.index_select(0, dst), e], 1), p, pre + "edge_model.edge_mlp.")
    m = _mlp2(torch.cat([x.index_select(0, src), e2], 1), p, pre + "node_model.node_mlp_1.")
    N = x.shape[0]
    s = scatter_add_rows(m, dst, N)
    cnt = torch.zeros(N, dtype=x.dtype).scatter_add_(0, dst, torch.ones(dst.shape[0], dtype=x.dtype))
    agg = s / cnt.clamp(min=1).view(-1, 1)                                       # scatter_mean
    x2 = _mlp2(torch.cat([x, agg], 1), p, pre + "node_model.node_mlp_2.")
    x2 = graph_layernorm(x2, batch, num_graphs, p["graph_layer_norm.weight"], p["graph_layer_norm.bias"])
    return x2, e2
