"""GQA scene-graph JSON -> COO graph in the hot path's input contract.

Host-side restatement of the topology rules of the reference's converter
(gqa_dataset_entry.py:231-332) and of PyG `Batch.from_data_list` offsets
(gqa_dataset_entry.py:654): nodes = object ids sorted as strings; per node, in
order: one self-loop, then for each outgoing relation the forward edge and --
only if the reverse (dst, src) pair is absent from the relation set -- a reverse
edge flagged in `added_sym_edge`.  Multi-edges are kept.  Token features are
vocabulary-dependent and not produced here (the builder is a
"next" row; only the topology is needed by the execution path and its tests).
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
