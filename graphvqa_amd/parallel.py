"""Data-parallel sharding of scene-graph batches over the GPUs of one node.

Graphs are independent units (block-diagonal batch, no edge crosses graphs -- the reference
uses `batch[edge_index[0]]` as *the* graph of an edge, gat_skip.py:257), and eval-mode BatchNorm is
a per-channel affine, so the K hops need NO communication: each rank (one process per GPU) runs
a contiguous range of graphs.  The only exchange is an all-gather of per-graph result rows
(the north star's "all-gather of per-batch logits"; the reference itself never gathers --
every rank overwrites the same dump file, mainExplain_gat.py:938-942).  Backend: RCCL on GPUs
(`nccl` in torch.distributed), gloo in the CPU tests.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def partition_graphs(edges_per_graph: np.ndarray, world_size: int) -> np.ndarray:
    """Contiguous graph ranges balanced by edge count.  Returns bounds[world_size + 1]: rank r owns
    graphs [bounds[r], bounds[r+1]).  Deterministic; every rank computes the same answer."""
    B = int(edges_per_graph.shape[0])
    csum = np.concatenate([[0], np.cumsum(edges_per_graph.astype(np.int64))])
    total = csum[-1]
    bounds = np.zeros(world_size + 1, dtype=np.int64)
    for r in range(1, world_size):
        target = total * r / world_size
        bounds[r] = int(np.searchsorted(csum, target, side="left"))
    bounds[world_size] = B
    return np.maximum.accumulate(np.minimum(bounds, B))


def shard_batch(edge_index: np.ndarray, batch: np.ndarray, num_graphs: int, rank: int, world_size: int):
    """Local shard of a COO batch for `rank`: (node slice, edge mask, local edge_index, local batch,
    graph range).  Node ids and graph ids are rebased to the shard."""
    edge_graph = batch[edge_index[0]]
    epg = np.bincount(edge_graph, minlength=num_graphs)
    bounds = partition_graphs(epg, world_size)
    g0, g1 = int(bounds[rank]), int(bounds[rank + 1])
    n0, n1 = int(np.searchsorted(batch, g0, side="left")), int(np.searchsorted(batch, g1, side="left"))
    emask = (edge_graph >= g0) & (edge_graph < g1)
    ei = edge_index[:, emask] - n0
    return slice(n0, n1), emask, ei, batch[n0:n1] - g0, (g0, g1)


class BatchShard:
    """One rank's share of ONE scene-graph batch (the `DistributedSampler` role of the reference's drivers,
    mainExplain_gat.py:197-198,226-227, at batch granularity): contiguous graph range balanced by edge count,
    node / edge / instruction tensors sliced to it, ids rebased.  Tensors live on `device`."""

    def __init__(self, edge_index: np.ndarray, batch: np.ndarray, num_graphs: int, x, edge_attr, instr, rank: int,
                 world_size: int, device):
        if edge_index.shape[1] and not np.array_equal(batch[edge_index[0]], batch[edge_index[1]]):
            raise ValueError("BatchShard: an edge joins two graphs of the batch; graphs are the unit of sharding")
        nsl, emask, ei, b, (g0, g1) = shard_batch(edge_index, batch, num_graphs, rank, world_size)
        epg = np.bincount(batch[edge_index[0]], minlength=num_graphs)
        bounds = partition_graphs(epg, world_size)
        self.counts = [int(bounds[r + 1] - bounds[r]) for r in range(world_size)]     # graphs per rank (every rank agrees)
        self.graph_range = (g0, g1)
        self.num_graphs, self.num_nodes, self.num_edges = g1 - g0, nsl.stop - nsl.start, int(ei.shape[1])
        as_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        self.x = as_t(x)[nsl].contiguous().to(device)
        self.edge_attr = as_t(edge_attr)[as_t(emask)].contiguous().to(device)
        self.instr = as_t(instr)[:, g0:g1].contiguous().to(device)
        self.edge_index = as_t(ei).contiguous().to(device)
        self.batch = as_t(b).contiguous().to(device)
        # the loader-side per-graph layout of the shard (host): lets the batch handle skip its statistics read-back
        self.host_nodes_per_graph = np.bincount(b, minlength=g1 - g0)
        self.host_edges_per_graph = epg[g0:g1]
        self.host_max_in_degree = int(np.bincount(ei[1]).max()) if ei.shape[1] else 0
        eg = b[ei[1]] if ei.shape[1] else np.zeros(0, np.int64)
        self.host_coo_grouped = bool(eg.shape[0] == 0 or np.all(eg[1:] >= eg[:-1]))      # COO edges in graph order (one-launch CSR build)

    def host_layout(self):
        from .graph import HostLayout
        return HostLayout(np.concatenate([[0], np.cumsum(self.host_nodes_per_graph)]),
                          np.concatenate([[0], np.cumsum(self.host_edges_per_graph)]), self.host_max_in_degree,
                          coo_grouped=self.host_coo_grouped)


def sharded_step(shard: BatchShard, forward, pool=None, force: bool = False) -> torch.Tensor:
    """One data-parallel step over one batch: `forward(shard) -> h [N_r, C]` on the local graphs (no communication inside
    the K hops), per-graph rows `pool(h, shard) -> [B_r, C]` (default: node mean), then ONE all-gather of the rows in
    graph order -> [B, C] on every rank.  bench.py drives the HIP path through this; the gloo CPU test drives the oracle
    through the same function."""
    h = forward(shard)
    rows = pool(h, shard) if pool is not None else graph_mean_pool(h, shard.batch, shard.num_graphs)
    return all_gather_graph_rows(rows, counts=shard.counts, force=force)


def graph_mean_pool(h: torch.Tensor, batch: torch.Tensor, num_graphs: int, graph=None) -> torch.Tensor:
    """Per-graph mean of node rows -> [B, C]; the small per-graph payload that is all-gathered when the answer head is
    not run.  With the batch handle (`graph`, a SceneGraphBatch) on a GPU this is one pass of the HIP segment-mean kernel
    over h (deterministic); the torch formulation (index_add_) serves the CPU tests."""
    if graph is not None and h.is_cuda:
        from .gat_skip import _segment_sum_raw, _f32c
        return _segment_sum_raw(_f32c(h, "h"), graph, mean=True)       # gvqa_graph_segment_mean: one launch
    out = torch.zeros((num_graphs, h.shape[1]), dtype=h.dtype, device=h.device)
    out.index_add_(0, batch, h)
    cnt = torch.bincount(batch, minlength=num_graphs).clamp(min=1).to(h.dtype)
    return out / cnt[:, None]


class GatheredRows:
    """Result of an all-gather of per-graph rows that may still be in flight: `wait()` orders the current stream (host
    thread for gloo) after the collective and returns the [sum B_r, C] rows in graph order."""

    def __init__(self, out, work=None, counts=None, mx=0):
        self._out, self._work, self._counts, self._mx = out, work, counts, mx

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        counts, mx, out = self._counts, self._mx, self._out
        if counts is None or all(c == mx for c in counts):
            return out
        return torch.cat([out[r * mx:r * mx + counts[r]] for r in range(len(counts))])


def _gather_algo(algo=None) -> str:
    """`collective` = one `all_gather_into_tensor` (RCCL picks ring / tree + protocol); `direct` = every rank pushes its rows to
    all peers at once (one grouped batch of point-to-point sends / receives: on the fully connected xGMI topology of an 8-GPU node
    the 7 sends leave on 7 different links in ONE hop, where a ring serialises 7 steps each bound by one link).
    Default from GVQA_ALLGATHER (collective)."""
    algo = algo or os.environ.get("GVQA_ALLGATHER", "collective")
    if algo not in ("collective", "direct"):
        raise ValueError(f"all-gather algorithm {algo!r}: expected 'collective' or 'direct'")
    return algo


class _Works:
    """Several in-flight point-to-point requests behind one wait()."""

    def __init__(self, reqs, keep=None):
        self._reqs, self._keep = reqs, keep          # (`keep`: the send buffer stays alive until the requests completed)

    def wait(self):
        for r in self._reqs:
            r.wait()
        self._keep = None


def _direct_all_gather(out: torch.Tensor, rows: torch.Tensor, mx: int, async_op: bool):
    """out[r * mx:(r + 1) * mx] <- rank r's rows, by ONE grouped batch of isend / irecv to and from every peer (own slot: a
    device copy).  Receives land in place in `out`: no staging buffer, no second pass."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    out[rank * mx:(rank + 1) * mx].copy_(rows)
    ops = []
    for d in range(1, world):                      # peer order rotated by rank: at every position of the batch the world's sends form a permutation
        to, frm = (rank + d) % world, (rank - d) % world
        ops.append(dist.P2POp(dist.isend, rows, to))
        ops.append(dist.P2POp(dist.irecv, out[frm * mx:(frm + 1) * mx], frm))
    if not ops:
        return None
    works = _Works(dist.batch_isend_irecv(ops), rows)
    if async_op:
        return works
    works.wait()
    return None


def all_gather_graph_rows(rows: torch.Tensor, counts=None, force: bool = False, async_op: bool = False, algo=None):
    """All-gather per-graph rows [B_r, C] from every rank into [sum B_r, C] (rank order).
    Ragged shards are padded to the largest B_r (one collective, latency-bound at these sizes).
    async_op=True: the collective is enqueued on the backend's own stream (ordered after the current stream's work so far)
    and a GatheredRows handle is returned instead of the tensor -- the caller's stream goes on with the next batch.
    algo: 'collective' | 'direct' (see _gather_algo; default from GVQA_ALLGATHER)."""
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return GatheredRows(rows) if async_op else rows
    algo = _gather_algo(algo)
    world = dist.get_world_size()
    if counts is None:
        c = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
        cl = [torch.zeros_like(c) for _ in range(world)]
        dist.all_gather(cl, c)
        counts = [int(v.item()) for v in cl]
    mx = max(counts)
    if rows.shape[0] < mx:
        rows = torch.cat([rows, rows.new_zeros((mx - rows.shape[0], rows.shape[1]))])
    rows = rows.contiguous()
    if rows.is_cuda and dist.get_backend() == "gloo":      # (gloo has no device all-gather: test set-ups with N ranks on one GPU go through the host)
        host = torch.empty((world * mx, rows.shape[1]), dtype=rows.dtype)
        if algo == "direct":
            _direct_all_gather(host, rows.cpu(), mx, False)
        else:
            dist.all_gather_into_tensor(host, rows.cpu())
        res = GatheredRows(host.to(rows.device), None, counts, mx)
        return res if async_op else res.wait()
    out = torch.empty((world * mx, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    if algo == "direct":
        work = _direct_all_gather(out, rows, mx, async_op)
    else:
        work = dist.all_gather_into_tensor(out, rows, async_op=async_op)
    res = GatheredRows(out, work if async_op else None, counts, mx)
    return res if async_op else res.wait()


class PipelinedSteps:
    """The steady-state loop over batches (the reference's eval loop, mainExplain_gat.py:226-227 + :791 per batch) with the
    exchange off the critical path: the all-gather of batch i's per-graph rows runs on the collective's stream while the K
    hops of batch i + 1 run on the compute stream (graphs are independent units: nothing in batch i + 1 reads batch i's
    gathered rows).  `step` returns the PREVIOUS batch's gathered rows (None on the first call), `drain` the last one's."""

    def __init__(self):
        self._pending = None

    def step(self, shard: BatchShard, forward, pool=None, force: bool = False):
        h = forward(shard)
        rows = pool(h, shard) if pool is not None else graph_mean_pool(h, shard.batch, shard.num_graphs)
        nxt = all_gather_graph_rows(rows, counts=shard.counts, force=force, async_op=True)
        prev = self.drain()
        self._pending = nxt
        return prev

    def drain(self):
        prev, self._pending = self._pending, None
        return None if prev is None else prev.wait()


def allreduce_gradients(parameters, bucket_bytes: int = 64 << 20, average: bool = True) -> int:
    """Data-parallel training step glue (the reference's DDP all-reduce, mainExplain_gat.py:259-263):
    sum (or average) the gradients of `parameters` over all ranks with a few LARGE flat all-reduces.

    Gradients are packed into contiguous buckets of about `bucket_bytes` (one RCCL ring all-reduce per bucket: on
    xGMI's point-to-point links a ring collective is per-link bandwidth bound, so few large messages beat many small
    ones; gat_seq's 9.8 M fp32 parameters are a single 39 MB bucket), reduced in place and scattered back.  Parameters
    without a gradient on this rank (e.g. unused on an empty shard) contribute zeros so that every rank issues the
    same collectives.  Returns the number of all-reduces issued.  No-op without an initialised process group or with
    one rank."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    params = [p for p in parameters if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    calls, i = 0, 0
    while i < len(params):
        j, nbytes = i, 0
        while j < len(params) and (j == i or nbytes + params[j].numel() * params[j].element_size() <= bucket_bytes) \
                and params[j].dtype == params[i].dtype and params[j].device == params[i].device:
            nbytes += params[j].numel() * params[j].element_size()
            j += 1
        grads = [p.grad for p in params[i:j]]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(world)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        calls += 1
        i = j
    return calls
