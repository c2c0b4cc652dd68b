"""GQA scene-graph JSON -> COO graph in the hot path's input contract.

Host-side restatement of the topology rules of the reference's converter
(gqa_dataset_entry.py:231-332) and of PyG `Batch.from_data_list` offsets
(gqa_dataset_entry.py:654): nodes = object ids sorted as strings; per node, in
order: one self-loop, then for each outgoing relation the forward edge and --
only if the reverse (dst, src) pair is absent from the relation set -- a reverse
edge flagged in `added_sym_edge`.  Multi-edges are kept.

`convert_scene_graph` adds the token features of the same converter (:259-316): 12 token ids per node
(object name, then its de-duplicated attributes, `<pad>` after them), one token id per edge (`<self>` for
the self-loop, the relation name for a relation AND for its added reverse edge -- the encoder flips the sign of the
latter's embedding through `added_sym_edge`, pipeline_model_gat.py:590).  `collate_scene_graphs` is the
`Batch.from_data_list` step (:654) as a Python restatement; `flatten_scene_graphs` + `collate_flat_scene_graphs` are the feed's
form: one tokenising pass, then the library's native collate (`gvqa_scene_graph_collate`, csrc/collate.hip).  And `DeviceSceneGraphs` puts the batch on the GPU together with its CSR handle, built
from the loader-side layout (no device read-back).  Pinned on the four debug graphs against the reference's own converter
(tests/golden/sg_builder_debug4.npz).
"""
from __future__ import annotations

import numpy as np

from .synth import GraphBatch


def scene_graph_topology(sg: dict):
    """Returns (num_nodes, edge_index [2,E] int64, added_sym_edge [A] int64)."""
    objects = sg["objects"]
    if len(objects) == 0:
        # the reference substitutes a 2-node dummy graph (gqa_dataset_entry.py:196-224)
        objects = {"0": {"relations": [{"object": "1"}]}, "1": {"relations": [{"object": "0"}]}}
    obj_ids = sorted(objects.keys())
    node_of = {o: i for i, o in enumerate(obj_ids)}
    pairs = set()
    for i, o in enumerate(obj_ids):
        for rel in objects[o]["relations"]:
            pairs.add((i, node_of[rel["object"]]))
    edges, added = [], []
    for i, o in enumerate(obj_ids):
        edges.append((i, i))
        for rel in objects[o]["relations"]:
            j = node_of[rel["object"]]
            edges.append((i, j))
            if (j, i) not in pairs:
                edges.append((j, i))
                added.append(len(edges) - 1)
    ei = np.asarray(edges, dtype=np.int64).T.copy()
    return len(obj_ids), ei, np.asarray(added, dtype=np.int64)


MAX_OBJ_TOKEN_LEN = 12      # gqa_dataset_entry.py:263

_DUMMY = {"objects": {      # what the reference substitutes for an empty scene graph (gqa_dataset_entry.py:196-224)
    "0": {"name": "<UNK>", "relations": [{"object": "1", "name": "<UNK>"}], "attributes": ["<UNK>"]},
    "1": {"name": "<UNK>", "relations": [{"object": "0", "name": "<UNK>"}], "attributes": ["<UNK>"]}}}


def convert_scene_graph(sg: dict, stoi, pad_token: str = "<pad>", self_token: str = "<self>"):
    """One GQA scene graph -> (x_tokens [N, 12], edge_index [2, E], edge_tokens [E, 1], added_sym_edge [A]), int64, exactly
    the tensors of the reference's `convert_one_gqa_scene_graph` (gqa_dataset_entry.py:190-372).  `stoi` maps a string
    to its id and must return the unknown id for out-of-vocabulary strings (torchtext's defaultdict behaviour; a plain dict
    is wrapped with unknown = 0).  Attribute tokens keep first-occurrence order (the reference iterates a Python `set`, whose
    order is not defined -- the encoder sums the 12 embeddings, so only the multiset matters)."""
    if len(sg["objects"]) == 0:
        sg = _DUMMY
    lookup = stoi.__getitem__ if hasattr(stoi, "default_factory") else (lambda w: stoi.get(w, 0))
    objects = sg["objects"]
    obj_ids = sorted(objects.keys())
    node_of = {o: i for i, o in enumerate(obj_ids)}
    n, ei, added = scene_graph_topology(sg)
    x = np.full((n, MAX_OBJ_TOKEN_LEN), lookup(pad_token), dtype=np.int64)
    pairs = _relation_pairs(objects, obj_ids, node_of)          # once per graph
    e_tok = []
    for i, o in enumerate(obj_ids):
        obj = objects[o]
        x[i, 0] = lookup(obj["name"])
        for a_idx, attr in enumerate(dict.fromkeys(obj["attributes"])):      # de-duplicated (the reference: set(...), :282)
            x[i, a_idx + 1] = lookup(attr)
        e_tok.append(lookup(self_token))
        for rel in obj["relations"]:
            e_tok.append(lookup(rel["name"]))
            if (node_of[rel["object"]], i) not in pairs:
                e_tok.append(lookup(rel["name"]))                             # the added reverse edge re-uses the name (:327)
    e_tok = np.asarray(e_tok, dtype=np.int64).reshape(-1, 1)
    assert e_tok.shape[0] == ei.shape[1]
    return x, ei, e_tok, added


def _relation_pairs(objects, obj_ids, node_of):
    pairs = set()
    for i, o in enumerate(obj_ids):
        for rel in objects[o]["relations"]:
            pairs.add((i, node_of[rel["object"]]))
    return pairs


class CollatedSceneGraphs:
    """Host-side batch in the path's input contract (what `GQATorchDataset_collate_fn` hands to the model,
    gqa_dataset_entry.py:631-675): block-diagonal `edge_index`, `batch`, token features, `added_sym_edge` rebased to the
    batched edge list, and the per-graph layout the loader knows for free."""

    def __init__(self, x_tokens, edge_index, edge_tokens, added_sym_edge, batch, nodes_per_graph, edges_per_graph):
        self.x, self.edge_index, self.edge_attr, self.added_sym_edge, self.batch = x_tokens, edge_index, edge_tokens, added_sym_edge, batch
        self.nodes_per_graph, self.edges_per_graph = nodes_per_graph, edges_per_graph
        self.num_graphs, self.num_nodes, self.num_edges = len(nodes_per_graph), int(x_tokens.shape[0]), int(edge_index.shape[1])

    def host_layout(self):
        from .graph import HostLayout
        if getattr(self, "_layout", None) is not None:          # the native collate hands the layout over with the batch
            return HostLayout(*self._layout, coo_grouped=True)
        deg = int(np.bincount(self.edge_index[1]).max()) if self.num_edges else 0
        # in-edges by DESTINATION graph: every edge stays inside its graph, so they equal the per-graph edge counts
        # (the collate concatenates the graphs' edge lists in graph order: the COO edges are grouped by graph)
        return HostLayout(np.concatenate([[0], np.cumsum(self.nodes_per_graph)]), np.concatenate([[0], np.cumsum(self.edges_per_graph)]), deg,
                          coo_grouped=True)

    def to(self, device):
        return DeviceSceneGraphs(self, device)


def collate_scene_graphs(sgs, stoi, pad_token: str = "<pad>", self_token: str = "<self>") -> CollatedSceneGraphs:
    xs, eis, ets, adds, batch, npg, epg = [], [], [], [], [], [], []
    n_off = e_off = 0
    for g, sg in enumerate(sgs):
        x, ei, et, added = convert_scene_graph(sg, stoi, pad_token, self_token)
        xs.append(x); eis.append(ei + n_off); ets.append(et); adds.append(added + e_off)
        batch.append(np.full(x.shape[0], g, dtype=np.int64))
        npg.append(x.shape[0]); epg.append(ei.shape[1])
        n_off += x.shape[0]; e_off += ei.shape[1]
    return CollatedSceneGraphs(np.concatenate(xs), np.concatenate(eis, axis=1), np.concatenate(ets), np.concatenate(adds),
                               np.concatenate(batch), np.asarray(npg, np.int64), np.asarray(epg, np.int64))


class FlatSceneGraphs:
    """Pre-tokenised, flattened scene graphs: the input of the native collate (`gvqa_scene_graph_collate`, csrc/collate.hip).
    This is what a loader caches per image next to the JSON: the string -> id lookups are vocabulary work (out of the path's
    scope), everything after them -- node order, self-loops, reverse edges, `added_sym_edge`, batching -- is the C function's."""

    def __init__(self, graph_obj_ptr, name_tok, attr_ptr, attr_tok, rel_ptr, rel_dst, rel_tok, pad_tok, self_tok, unk_tok):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        self.graph_obj_ptr, self.attr_ptr, self.rel_ptr, self.rel_dst = i32(graph_obj_ptr), i32(attr_ptr), i32(rel_ptr), i32(rel_dst)
        self.name_tok, self.attr_tok, self.rel_tok = i64(name_tok), i64(attr_tok), i64(rel_tok)
        self.pad_tok, self.self_tok, self.unk_tok = int(pad_tok), int(self_tok), int(unk_tok)
        self.num_graphs = len(self.graph_obj_ptr) - 1


def flatten_scene_graphs(sgs, stoi, pad_token: str = "<pad>", self_token: str = "<self>", unk_token: str = "<unk>") -> FlatSceneGraphs:
    """Tokenise a list of GQA scene-graph dicts into the flat arrays of `FlatSceneGraphs` (one pass over the JSON; object ids
    sorted as strings like the converter, gqa_dataset_entry.py:231-232; attribute STRINGS de-duplicated in first-occurrence
    order, :282)."""
    lookup = stoi.__getitem__ if hasattr(stoi, "default_factory") else (lambda w: stoi.get(w, 0))
    gptr, name, aptr, atok, rptr, rdst, rtok = [0], [], [0], [], [0], [], []
    for sg in sgs:
        objects = sg["objects"]
        obj_ids = sorted(objects.keys())
        node_of = {o: i for i, o in enumerate(obj_ids)}
        for o in obj_ids:
            obj = objects[o]
            name.append(lookup(obj["name"]))
            atok.extend(lookup(a) for a in dict.fromkeys(obj["attributes"]))
            aptr.append(len(atok))
            for rel in obj["relations"]:
                rdst.append(node_of[rel["object"]])
                rtok.append(lookup(rel["name"]))
            rptr.append(len(rdst))
        gptr.append(len(name))
    return FlatSceneGraphs(gptr, name, aptr, atok, rptr, rdst, rtok, lookup(pad_token), lookup(self_token), lookup(unk_token))


def collate_flat_scene_graphs(flat: FlatSceneGraphs) -> CollatedSceneGraphs:
    """The batch of `collate_scene_graphs`, produced by the library's native collate from flattened scene graphs: two C calls
    (sizes, fill), no Python loop."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    sizes = np.zeros(3, np.int64)
    _lib.check(lib.gvqa_scene_graph_collate_sizes(flat.num_graphs, p(flat.graph_obj_ptr), p(flat.rel_ptr), p(flat.rel_dst), p(sizes)))
    N, E, A = (int(v) for v in sizes)
    B = flat.num_graphs
    x = np.empty((N, MAX_OBJ_TOKEN_LEN), np.int64)
    ei, et = np.empty((2, E), np.int64), np.empty((E, 1), np.int64)
    added, batch = np.empty(A, np.int64), np.empty(N, np.int64)
    gptr, eptr = np.empty(B + 1, np.int32), np.empty(B + 1, np.int32)
    maxdeg = C.c_int32(0)
    _lib.check(lib.gvqa_scene_graph_collate(B, p(flat.graph_obj_ptr), p(flat.name_tok), p(flat.attr_ptr), p(flat.attr_tok), p(flat.rel_ptr),
                                            p(flat.rel_dst), p(flat.rel_tok), flat.pad_tok, flat.self_tok, flat.unk_tok, N, E, A, p(x), p(ei),
                                            p(et), p(added), p(batch), p(gptr), p(eptr), C.byref(maxdeg)))
    c = CollatedSceneGraphs(x, ei, et, added, batch, np.diff(gptr).astype(np.int64), np.diff(eptr).astype(np.int64))
    c._layout = (gptr.astype(np.int64), eptr.astype(np.int64), int(maxdeg.value))
    return c


class DeviceSceneGraphs:
    """The collated batch on the GPU -- attribute names as the reference's `Batch` (x, edge_index, edge_attr, batch,
    added_sym_edge), so `GroundTruth_SceneGraph_Encoder.forward(gt_scene_graphs)` / `gat_seq.forward(...)` take it as is --
    plus `graph`: the destination-sorted CSR handle built on the device from the COO arrays, finalized from the loader-side
    layout (no synchronisation)."""

    def __init__(self, c: CollatedSceneGraphs, device):
        import torch
        from .graph import SceneGraphBatch
        dev = torch.device(device)
        put = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
        self.x, self.edge_index, self.edge_attr = put(c.x), put(c.edge_index), put(c.edge_attr)
        self.batch, self.added_sym_edge = put(c.batch), put(c.added_sym_edge)
        self.num_graphs, self.num_nodes, self.num_edges = c.num_graphs, c.num_nodes, c.num_edges
        self.graph = SceneGraphBatch(self.edge_index, self.batch, c.num_nodes, c.num_graphs, host_layout=c.host_layout())


def batch_scene_graphs(sgs) -> GraphBatch:
    """Block-diagonal batching (node-id offsets per graph, `batch` vector)."""
    eis, batch, off = [], [], 0
    for g, sg in enumerate(sgs):
        n, ei, _ = scene_graph_topology(sg)
        eis.append(ei + off)
        batch.append(np.full(n, g, dtype=np.int64))
        off += n
    return GraphBatch(edge_index=np.concatenate(eis, axis=1), batch=np.concatenate(batch),
                      num_graphs=len(sgs))
