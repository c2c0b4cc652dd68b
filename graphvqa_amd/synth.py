"""Deterministic synthetic scene-graph batches and tensors.

Everything here is a pure function of integer seeds, built on a counter-based
generator (splitmix64 -> uniform / Box-Muller) evaluated with numpy uint64
arithmetic.  The same call gives bit-identical arrays in this container and on
the GPU box, so golden fixtures only need to store inputs/outputs that are not
cheaply regenerable (and never multi-MB weight tensors).

Shapes follow SURVEY.md section 8(d): GQA-shaped batches of small graphs with one
explicit self-loop per node (reference: gqa_dataset_entry.py:292-297) followed
by random intra-graph relation edges, edges grouped by source node like the
reference's converter emits them (gqa_dataset_entry.py:255-332).  Multi-edges
are allowed and kept (they are real edges with their own features).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on an array of uint64 counters."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _bits(n: int, seed: int, stream: int = 0) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([(seed * 0x100000001B3 + stream) & 0xFFFFFFFFFFFFFFFF],
                                    dtype=np.uint64))[0]
        ctr = np.arange(n, dtype=np.uint64) + base
    return _splitmix64(ctr)


def uniform01(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """n float64 values in [0, 1) (53 random bits each)."""
    return (_bits(n, seed, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform(shape, seed: int, lo: float = 0.0, hi: float = 1.0, dtype=np.float32) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(n, seed, 0)
    return (lo + (hi - lo) * u).astype(dtype).reshape(shape)


def normal(shape, seed: int, dtype=np.float32) -> np.ndarray:
    """Standard normal via Box-Muller on two independent streams."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(n, seed, 1)
    u2 = uniform01(n, seed, 2)
    r = np.sqrt(-2.0 * np.log1p(-u1))          # 1-u1 in (0, 1]
    return (r * np.cos(2.0 * math.pi * u2)).astype(dtype).reshape(shape)


def randint(n: int, seed: int, lo, hi, stream: int = 3) -> np.ndarray:
    """n int64 values, element i uniform in [lo_i, hi_i) (lo/hi scalars or arrays)."""
    u = uniform01(n, seed, stream)
    lo = np.asarray(lo, dtype=np.int64)
    hi = np.asarray(hi, dtype=np.int64)
    return (lo + np.floor(u * (hi - lo)).astype(np.int64)).astype(np.int64)


def glorot(shape, seed: int) -> np.ndarray:
    """PyG `inits.glorot`: U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))
    (reference call sites: gat_skip.py:101-107)."""
    a = math.sqrt(6.0 / (shape[-2] + shape[-1]))
    return uniform(shape, seed, -a, a)


@dataclass
class GraphBatch:
    """COO batch in the reference's input contract.

    edge_index[0] = source, edge_index[1] = destination (int64), batch[n] = graph
    id of node n (non-decreasing), nodes of one graph contiguous.
    """
    edge_index: np.ndarray   # [2, E] int64
    batch: np.ndarray        # [N] int64
    num_graphs: int

    @property
    def num_nodes(self) -> int:
        return int(self.batch.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])


def make_graph_batch(num_graphs: int, seed: int, nodes_lo: int = 20, nodes_hi: int = 40,
                     rel_per_node: float = 1.0, fixed_nodes: int | None = None,
                     fixed_rel: int | None = None) -> GraphBatch:
    """GQA-shaped batch: per graph n_i nodes, n_i self-loops + r_i random
    relation edges (src != dst when n_i > 1, duplicates allowed).

    Edges are emitted per graph grouped by source node: the node's self-loop
    first, then that node's outgoing relation edges (reference order,
    gqa_dataset_entry.py:292-332).
    """
    if fixed_nodes is not None:
        n = np.full(num_graphs, fixed_nodes, dtype=np.int64)
    else:
        n = randint(num_graphs, seed, nodes_lo, nodes_hi + 1, stream=4)
    if fixed_rel is not None:
        r = np.full(num_graphs, fixed_rel, dtype=np.int64)
    else:
        r = np.floor(n * rel_per_node + 0.5).astype(np.int64)
    r = np.where(n > 1, r, 0)
    node_off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    rel_off = np.concatenate([[0], np.cumsum(r)]).astype(np.int64)
    N, R = int(node_off[-1]), int(rel_off[-1])

    # relation edges: graph id per relation, local src/dst
    g_of_rel = np.repeat(np.arange(num_graphs, dtype=np.int64), r)
    n_rel = n[g_of_rel]
    src_l = randint(R, seed, 0, n_rel, stream=5)
    dst_l = randint(R, seed, 0, np.maximum(n_rel - 1, 1), stream=6)
    dst_l = np.where(dst_l >= src_l, dst_l + 1, dst_l)          # src != dst
    dst_l = np.minimum(dst_l, n_rel - 1)
    rel_src = node_off[g_of_rel] + src_l
    rel_dst = node_off[g_of_rel] + dst_l

    # merge: key = source node, self-loop first (tag 0) then relations (tag 1, stable)
    self_src = np.arange(N, dtype=np.int64)
    src = np.concatenate([self_src, rel_src])
    dst = np.concatenate([self_src, rel_dst])
    tag = np.concatenate([np.zeros(N, np.int64), np.ones(R, np.int64)])
    order = np.lexsort((np.arange(N + R), tag, src))
    edge_index = np.stack([src[order], dst[order]]).astype(np.int64)
    batch = np.repeat(np.arange(num_graphs, dtype=np.int64), n)
    return GraphBatch(edge_index=edge_index, batch=batch, num_graphs=num_graphs)


# Named workloads (BASELINE.json configs / SURVEY 8d).
def config2_batch() -> GraphBatch:
    """1k graphs, ~30 nodes / ~60 edges each (n_i ~ U{20..40}, e_i = 2 n_i)."""
    return make_graph_batch(1000, seed=0x5EED0002, nodes_lo=20, nodes_hi=40, rel_per_node=1.0)


def config3_batch(num_graphs: int = 2048) -> GraphBatch:
    """64k nodes / 256k edges: 2048 graphs x 32 nodes x (32 self-loops + 96 relations)."""
    return make_graph_batch(num_graphs, seed=0x5EED0003, fixed_nodes=32, fixed_rel=96)


# ----------------------------------------------------------------------------
# Parameters, keyed like the reference modules' state_dicts
# ----------------------------------------------------------------------------
def _affine(C: int, seed: int, randomize: bool):
    """BatchNorm1d tensors: default init (1, 0, 0, 1) or seeded non-trivial values."""
    if not randomize:
        return (np.ones(C, np.float32), np.zeros(C, np.float32),
                np.zeros(C, np.float32), np.ones(C, np.float32))
    return (uniform((C,), seed + 1, 0.5, 1.5), uniform((C,), seed + 2, -0.5, 0.5),
            uniform((C,), seed + 3, -0.5, 0.5), uniform((C,), seed + 4, 0.5, 2.0))


def gat_seq_params(in_channels: int, out_channels: int, edge_attr_dim: int, ins_dim: int,
                   num_ins: int, heads: int, seed: int, randomize_affine: bool = True) -> dict:
    """`gat_seq` parameters (gat_skip.py:224-235) with the reference's init rule
    (glorot on lin/att, gat_skip.py:101-107); bias / BN tensors are optionally
    randomised so that tests exercise them (the reference initialises them to 0 / (1,0,0,1))."""
    p = {}
    H, C = heads, out_channels
    for i in range(num_ins):
        s = seed + 1000 * (i + 1)
        W_l = glorot((H * C, in_channels + ins_dim), s + 1)
        p[f"convs.{i}.lin_l.weight"] = W_l
        p[f"convs.{i}.lin_r.weight"] = W_l          # alias of lin_l (gat_skip.py:76-77)
        p[f"convs.{i}.lin_e.weight"] = glorot((H * C, edge_attr_dim + ins_dim), s + 2)
        p[f"convs.{i}.att_l"] = glorot((1, H, C), s + 3)
        p[f"convs.{i}.att_r"] = glorot((1, H, C), s + 4)
        p[f"convs.{i}.att_e"] = glorot((1, H, C), s + 5)
        p[f"convs.{i}.bias"] = (uniform((C,), s + 6, -0.1, 0.1) if randomize_affine
                                else np.zeros(C, np.float32))
    for j in range(num_ins - 1):
        w, b, rm, rv = _affine(C, seed + 500000 + 10 * j, randomize_affine)
        p[f"bns.{j}.weight"], p[f"bns.{j}.bias"] = w, b
        p[f"bns.{j}.running_mean"], p[f"bns.{j}.running_var"] = rm, rv
        p[f"bns.{j}.num_batches_tracked"] = np.zeros((), np.int64)
    return p


def _linear(out_f: int, in_f: int, seed: int, bias: bool = True, prefix: str = "") -> dict:
    a = 1.0 / math.sqrt(in_f)
    d = {prefix + "weight": uniform((out_f, in_f), seed, -a, a)}
    if bias:
        d[prefix + "bias"] = uniform((out_f,), seed + 7, -a, a)
    return d


def gine_seq_params(in_channels: int, out_channels: int, ins_dim: int, seed: int,
                    num_layers: int = 5, randomize_affine: bool = True) -> dict:
    """`gine_seq` (pipeline_model_gine.py:622-634): GINEConv(Seq(Lin, ReLU, Lin)) x5 + 4 BN."""
    p = {}
    D = in_channels + ins_dim
    for i in range(num_layers):
        s = seed + 1000 * (i + 1)
        p.update(_linear(out_channels, D, s + 1, prefix=f"convs.{i}.nn.0."))
        p.update(_linear(out_channels, out_channels, s + 2, prefix=f"convs.{i}.nn.2."))
        p[f"convs.{i}.eps"] = np.zeros(1, np.float32)
    for j in range(num_layers - 1):
        w, b, rm, rv = _affine(out_channels, seed + 500000 + 10 * j, randomize_affine)
        p[f"bns.{j}.weight"], p[f"bns.{j}.bias"] = w, b
        p[f"bns.{j}.running_mean"], p[f"bns.{j}.running_var"] = rm, rv
        p[f"bns.{j}.num_batches_tracked"] = np.zeros((), np.int64)
    return p


def gcn_seq_params(in_channels: int, out_channels: int, ins_dim: int, seed: int,
                   num_layers: int = 5, randomize_affine: bool = True) -> dict:
    """`gcn_seq` (pipeline_model_gcn.py:622-634): GCNConv(in+ins, out) x5 (weight [in, out]) + 4 BN."""
    p = {}
    D = in_channels + ins_dim
    for i in range(num_layers):
        s = seed + 1000 * (i + 1)
        p[f"convs.{i}.weight"] = glorot((D, out_channels), s + 1)
        p[f"convs.{i}.bias"] = (uniform((out_channels,), s + 2, -0.1, 0.1) if randomize_affine
                                else np.zeros(out_channels, np.float32))
    for j in range(num_layers - 1):
        w, b, rm, rv = _affine(out_channels, seed + 500000 + 10 * j, randomize_affine)
        p[f"bns.{j}.weight"], p[f"bns.{j}.bias"] = w, b
        p[f"bns.{j}.running_mean"], p[f"bns.{j}.running_var"] = rm, rv
        p[f"bns.{j}.num_batches_tracked"] = np.zeros((), np.int64)
    return p


def lcgn_seq_params(in_channels: int, out_channels: int, seed: int, cmd_dim: int = 512,
                    question_dim: int = 512, max_iter: int = 4, heads: int = 1, num_ins: int = 5,
                    randomize_affine: bool = True) -> dict:
    """`lcgn_seq` (lcgn.py:255-282) incl. its dead `bns` entries (kept: state_dict contract)."""
    p = {}
    O = out_channels
    p.update(_linear(O, in_channels, seed + 1, prefix="init_sg_emb_input.0."))
    p.update(_linear(O, question_dim, seed + 2, prefix="qInput1."))
    for t in range(max_iter):
        p.update(_linear(O, O, seed + 10 + t, prefix=f"qInput2_{t}."))
    p.update(_linear(1, O, seed + 20, prefix="cmd_inter2logits."))
    p.update(_linear(O, O, seed + 21, prefix="proj_x_loc.1."))
    p.update(_linear(O, O, seed + 22, prefix="proj_x_ctx.1."))
    p.update(_linear(O, 2 * O, seed + 23, prefix="output_layer."))
    p.update(_linear(O, 2 * O, seed + 24, prefix="fin_layer."))
    HC = heads * O
    p["lcgn.lin_l.weight"] = glorot((HC, 3 * O), seed + 30)
    p["lcgn.lin_r.weight"] = glorot((HC, 3 * O), seed + 31)
    p["lcgn.cal_x.weight"] = glorot((HC, 3 * O), seed + 32)
    p["lcgn.proj_cmd.weight"] = glorot((HC, cmd_dim), seed + 33)
    p["lcgn.cal_cmd.weight"] = glorot((HC, cmd_dim), seed + 34)
    p["lcgn.bias"] = (uniform((O,), seed + 35, -0.1, 0.1) if randomize_affine
                      else np.zeros(O, np.float32))
    for j in range(num_ins - 1):
        w, b, rm, rv = _affine(O, seed + 500000 + 10 * j, randomize_affine)
        p[f"bns.{j}.weight"], p[f"bns.{j}.bias"] = w, b
        p[f"bns.{j}.running_mean"], p[f"bns.{j}.running_var"] = rm, rv
        p[f"bns.{j}.num_batches_tracked"] = np.zeros((), np.int64)
    return p


def attention_pool_params(num_node_features: int, channels: int, seed: int) -> dict:
    """`MyConditionalGlobalAttention(num_node_features, num_out_features)` (pipeline_model_gat.py:134-147)."""
    p = {}
    p.update(_linear(channels, channels, seed + 1, prefix="gate_nn.0."))
    p.update(_linear(1, channels, seed + 2, prefix="gate_nn.2."))
    p.update(_linear(channels, num_node_features, seed + 3, prefix="node_nn.0."))
    p.update(_linear(channels, channels, seed + 4, prefix="node_nn.2."))
    p.update(_linear(channels, channels, seed + 5, prefix="ques_nn.0."))
    p.update(_linear(channels, channels, seed + 6, prefix="ques_nn.2."))
    return p


def classifier_params(question_dim: int, hidden: int, num_answers: int, seed: int, prefix: str = "logit_fc.") -> dict:
    """`logit_fc` = Seq(Dropout, Lin(3Q, hidden), ELU, Dropout, Lin(hidden, A)) (pipeline_model_gat.py:718-728)."""
    p = {}
    p.update(_linear(hidden, 3 * question_dim, seed + 1, prefix=prefix + "1."))
    p.update(_linear(num_answers, hidden, seed + 2, prefix=prefix + "4."))
    return p


def encoder_params(vocab_size: int, dim: int, seed: int, pad_idx: int = 0) -> dict:
    """`GroundTruth_SceneGraph_Encoder` (pipeline_model_gat.py:553-573): embedding [V, D] (padding row
    zero), MetaLayer MLPs (EdgeModel 3D->D->D, NodeModel 2D->D->D twice), graph LayerNorm with
    1-element weight / bias (my_graph_layernorm.py:38-39)."""
    p = {}
    emb = normal((vocab_size, dim), seed + 1)
    emb[pad_idx] = 0.0
    p["sg_vocab_embedding.weight"] = emb
    pre = "scene_graph_encoding_layer."
    p.update(_linear(dim, 3 * dim, seed + 2, prefix=pre + "edge_model.edge_mlp.0."))
    p.update(_linear(dim, dim, seed + 3, prefix=pre + "edge_model.edge_mlp.2."))
    p.update(_linear(dim, 2 * dim, seed + 4, prefix=pre + "node_model.node_mlp_1.0."))
    p.update(_linear(dim, dim, seed + 5, prefix=pre + "node_model.node_mlp_1.2."))
    p.update(_linear(dim, 2 * dim, seed + 6, prefix=pre + "node_model.node_mlp_2.0."))
    p.update(_linear(dim, dim, seed + 7, prefix=pre + "node_model.node_mlp_2.2."))
    p["graph_layer_norm.weight"] = uniform((1,), seed + 8, 0.5, 1.5)
    p["graph_layer_norm.bias"] = uniform((1,), seed + 9, -0.5, 0.5)
    return p
