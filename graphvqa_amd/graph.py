"""Device-resident scene-graph batch (destination-sorted CSR) built by the HIP library.

Host-side handle for `struct gvqa_graph`.  Replaces the COO indexing PyG's `MessagePassing`
performs on every hop for the reference (gat_skip.py:155-156); the input is exactly what the
reference's collate function emits (gqa_dataset_entry.py:361-369, :654).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


class HostLayout:
    """Per-graph layout of a batch as the LOADER knows it on the host (the reference collates its Batch on the CPU,
    gqa_dataset_entry.py:631-675): first node of every graph and running in-edge counts by destination graph, int32
    [B + 1] each.  Passing it to SceneGraphBatch skips the device -> host read-back of the graph statistics (the only
    synchronisation of the path); the caller then vouches for an intra-graph batch with in-range indices."""

    def __init__(self, graph_ptr, edge_ptr, max_in_degree: int = 0, coo_grouped: bool = False):
        import numpy as np
        self.graph_ptr = np.ascontiguousarray(graph_ptr, dtype=np.int32)
        self.edge_ptr = np.ascontiguousarray(edge_ptr, dtype=np.int32)
        self.max_in_degree = int(max_in_degree)
        # the loader also vouches that the COO edges are grouped by graph (graph g's edges = COO positions [edge_ptr[g],
        # edge_ptr[g + 1]): what Batch.from_data_list yields, gqa_dataset_entry.py:654) -> the one-launch CSR build
        self.coo_grouped = bool(coo_grouped)

    @classmethod
    def from_numpy(cls, edge_index, batch, num_graphs: int):
        """From host copies of the COO batch (edge_index [2, E], batch [N]): two bincounts."""
        import numpy as np
        if edge_index.shape[1] and not np.array_equal(batch[edge_index[0]], batch[edge_index[1]]):
            raise ValueError("HostLayout: an edge joins two graphs of the batch (the loader-side layout is for intra-graph batches)")
        nodes = np.bincount(batch, minlength=num_graphs)
        edges = np.bincount(batch[edge_index[1]], minlength=num_graphs) if edge_index.shape[1] else np.zeros(num_graphs, np.int64)
        deg = int(np.bincount(edge_index[1]).max()) if edge_index.shape[1] else 0
        eg = batch[edge_index[1]]
        grouped = bool(eg.shape[0] == 0 or np.all(eg[1:] >= eg[:-1]))
        return cls(np.concatenate([[0], np.cumsum(nodes)]), np.concatenate([[0], np.cumsum(edges)]), deg, coo_grouped=grouped)


class SceneGraphBatch:
    """CSR-by-destination view of (edge_index [2,E] int64, batch [N] int64, num_graphs)."""

    def __init__(self, edge_index: torch.Tensor, batch: torch.Tensor | None, num_nodes: int,
                 num_graphs: int | None = None, host_layout: "HostLayout | None" = None):
        lib = _lib.load()
        if not edge_index.is_cuda:
            raise ValueError("SceneGraphBatch needs CUDA/HIP tensors (edge_index is on %s)" % edge_index.device)
        if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
            raise ValueError("edge_index must be int64 [2, E]")
        dev = edge_index.device
        edge_index = edge_index.contiguous()
        if batch is not None:
            if batch.dtype != torch.int64 or batch.dim() != 1 or batch.shape[0] != num_nodes:
                raise ValueError("batch must be int64 [N]")
            batch = batch.to(dev).contiguous()
        if num_graphs is None:
            # same host sync the reference's pipeline performs (pipeline_model_gat.py:152)
            num_graphs = int(batch[-1].item()) + 1 if (batch is not None and num_nodes > 0) else 1
        N, E, B = int(num_nodes), int(edge_index.shape[1]), int(num_graphs)
        self.device = dev
        self.num_nodes, self.num_edges, self.num_graphs = N, E, B
        nbytes = lib.gvqa_graph_workspace_bytes(N, E, B)
        self._ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        self._keep = (edge_index, batch)
        self.c = _lib.Graph()
        if host_layout is not None and host_layout.graph_ptr.shape[0] != B + 1:
            raise ValueError("host_layout does not match num_graphs")
        with torch.cuda.device(dev):
            if host_layout is not None and host_layout.coo_grouped:
                # COO edges grouped by graph: build + plan as one upload and one launch (falls through when out of its reach)
                rc = lib.gvqa_graph_build_grouped(N, E, B, _ptr(edge_index), _ptr(batch), host_layout.graph_ptr.ctypes.data,
                                                  host_layout.edge_ptr.ctypes.data, int(host_layout.max_in_degree),
                                                  self._ws.data_ptr(), self._ws.numel(), _stream(dev), C.byref(self.c))
                if rc != _lib.E_UNSUPPORTED:
                    _lib.check(rc)
                    self._host_layout = host_layout
                    return
            _lib.check(lib.gvqa_graph_build(N, E, B, _ptr(edge_index), _ptr(batch), self._ws.data_ptr(),
                                            self._ws.numel(), _stream(dev), C.byref(self.c)))
            if host_layout is not None:       # loader-side layout: no device synchronisation
                self._host_layout = host_layout
                _lib.check(lib.gvqa_graph_finalize_host(C.byref(self.c), host_layout.graph_ptr.ctypes.data,
                                                        host_layout.edge_ptr.ctypes.data, int(host_layout.max_in_degree),
                                                        _stream(dev)))
            else:
                _lib.check(lib.gvqa_graph_finalize(C.byref(self.c), _stream(dev)))

    def check_valid(self):
        """Deferred validation of a handle built from a loader-side layout (synchronises): raises GvqaError if the batch violates
        the input contract, has cross-graph edges, or exceeds the layout's statistics."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gvqa_graph_check_valid(C.byref(self.c), _stream(self.device)))
        return self

    def transposed(self) -> "SceneGraphBatch":
        """CSR by SOURCE of the same batch (flipped edge_index; same COO edge ids): what the backward of the
        message passing walks to scatter gradients to source nodes without atomics.  Built once, cached."""
        if getattr(self, "_transposed", None) is None:
            edge_index, batch = self._keep
            hl = getattr(self, "_host_layout", None)        # intra-graph batch: the same per-graph counts by source
            if hl is not None:
                hl = HostLayout(hl.graph_ptr, hl.edge_ptr, 0, coo_grouped=hl.coo_grouped)
            self._transposed = SceneGraphBatch(edge_index.flip(0), batch, self.num_nodes, self.num_graphs, host_layout=hl)
        return self._transposed

    # statistics ------------------------------------------------------------------------------
    @property
    def max_graph_nodes(self): return self.c.max_graph_nodes

    @property
    def max_graph_edges(self): return self.c.max_graph_edges

    @property
    def max_in_degree(self): return self.c.max_in_degree

    @property
    def intra_graph(self): return bool(self.c.intra_graph)

    # device arrays as torch views (tests / debugging) ------------------------------------------
    def _view(self, ptr, n):
        off = ptr - self._ws.data_ptr()
        return self._ws[off:off + 4 * n].view(torch.int32)

    @property
    def rowptr(self): return self._view(self.c.rowptr, self.num_nodes + 1)

    @property
    def csr_src(self): return self._view(self.c.csr_src, self.num_edges)

    @property
    def csr_eid(self): return self._view(self.c.csr_eid, self.num_edges)

    @property
    def graph_ptr(self): return self._view(self.c.graph_ptr, self.num_graphs + 1)

    @property
    def node_graph(self): return self._view(self.c.node_graph, self.num_nodes)
