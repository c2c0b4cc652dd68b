"""N>1 path on CPU: world_size-2 gloo processes.  Each rank takes its shard of a batch
(graphs partitioned by edge count), runs the hops on its shard (oracle stands in for the HIP
path here -- no GPU in this tier), and the per-graph rows are all-gathered; the result must equal
the single-process result on the whole batch (graphs are independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graphvqa_amd import synth
from graphvqa_amd.parallel import (partition_graphs, shard_batch, graph_mean_pool, all_gather_graph_rows, BatchShard,
                                   sharded_step, PipelinedSteps)
from tests.util import t, tparams


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case():
    dn, de, di, K, H = 16, 12, 8, 3, 4
    gb = synth.make_graph_batch(11, seed=31, nodes_lo=1, nodes_hi=14, rel_per_node=1.7)   # ragged, odd count
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(dn, dn, de, di, K, H, seed=55)
    x, ea, ins = synth.normal((N, dn), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    return gb, p, x, ea, ins, H


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_torch as R
        torch.set_num_threads(1)
        gb, p, x, ea, ins, H = _case()
        # the bench's step (parallel.sharded_step) with the oracle standing in for the HIP forward
        shard = BatchShard(gb.edge_index, gb.batch, gb.num_graphs, x, ea, ins, rank, world, torch.device("cpu"))
        fwd = lambda s: R.gat_seq(s.x, s.edge_index, s.edge_attr, s.instr, s.batch, tparams(p), heads=H)
        gathered = sharded_step(shard, fwd)               # ragged shards: padded all-gather
        nsl, emask, ei, b, (g0, g1) = shard_batch(gb.edge_index, gb.batch, gb.num_graphs, rank, world)
        assert (shard.num_graphs, shard.num_nodes, shard.num_edges) == (g1 - g0, nsl.stop - nsl.start, ei.shape[1])
        if rank == 0:
            q.put(gathered.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_partition_is_contiguous_balanced_and_total():
    epg = np.array([5, 1, 9, 3, 3, 7, 2, 8, 1, 1, 40])
    for w in (1, 2, 3, 4, 8, 16):
        b = partition_graphs(epg, w)
        assert b[0] == 0 and b[-1] == len(epg) and np.all(np.diff(b) >= 0)
    b2 = partition_graphs(np.full(2048, 128), 8)
    assert np.array_equal(b2, np.arange(9) * 256)           # config 3: 256 graphs per GPU
    assert np.array_equal(partition_graphs(np.zeros(4, np.int64), 2)[[0, -1]], [0, 4])


def test_shards_cover_the_batch_exactly():
    gb, *_ = _case()
    seen_n, seen_e = 0, 0
    for r in range(3):
        nsl, emask, ei, b, (g0, g1) = shard_batch(gb.edge_index, gb.batch, gb.num_graphs, r, 3)
        seen_n += nsl.stop - nsl.start
        seen_e += int(emask.sum())
        if ei.size:
            assert ei.min() >= 0 and ei.max() < (nsl.stop - nsl.start)
        assert b.size == 0 or (b.min() == 0 and b.max() == g1 - g0 - 1)
    assert seen_n == gb.num_nodes and seen_e == gb.num_edges


def test_two_rank_gloo_matches_single_process():
    from oracle import ref_torch as R
    gb, p, x, ea, ins, H = _case()
    full = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H)
    want = graph_mean_pool(full, t(gb.batch), gb.num_graphs).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-5


def _pipeline_worker(rank, world, port, q):
    """Three batches through PipelinedSteps (the all-gather of batch i in flight while batch i + 1 is computed): every
    batch's gathered rows must equal the blocking sharded_step's on the same batch."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_torch as R
        torch.set_num_threads(1)
        gb, p, x, ea, ins, H = _case()
        fwd = lambda s: R.gat_seq(s.x, s.edge_index, s.edge_attr, s.instr, s.batch, tparams(p), heads=H)
        shards = [BatchShard(gb.edge_index, gb.batch, gb.num_graphs, x * (1.0 + 0.5 * i), ea, ins, rank, world, torch.device("cpu"))
                  for i in range(3)]
        want = [sharded_step(s, fwd) for s in shards]
        pipe, got = PipelinedSteps(), []
        for s in shards:
            prev = pipe.step(s, fwd)
            if prev is not None:
                got.append(prev)
        assert len(got) == 2
        got.append(pipe.drain())
        assert pipe.drain() is None
        err = max(float((g - w).abs().max()) for g, w in zip(got, want))
        shapes = [tuple(g.shape) for g in got]
        if rank == 0:
            q.put((err, shapes, gb.num_graphs))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pipelined_steps_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    err, shapes, B = q.get(timeout=180)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert err == 0.0 and all(sh[0] == B for sh in shapes), (err, shapes)


def test_pipelined_steps_without_a_process_group():
    pipe = PipelinedSteps()
    gb, p, x, ea, ins, H = _case()
    shard = BatchShard(gb.edge_index, gb.batch, gb.num_graphs, x, ea, ins, 0, 1, torch.device("cpu"))
    fwd = lambda s: s.x
    assert pipe.step(shard, fwd) is None
    second = pipe.step(shard, fwd)
    want = graph_mean_pool(shard.x, shard.batch, shard.num_graphs)
    assert torch.equal(second, want) and torch.equal(pipe.drain(), want) and pipe.drain() is None


def _grad_worker(rank, world, port, q):
    """Data-parallel gradient exchange: each rank back-propagates its shard's loss (sum over its graphs) through the
    oracle (stand-in for the HIP path on this tier), then allreduce_gradients; the summed gradient must equal the
    single-process gradient of the whole batch's loss (eval BatchNorm: graphs are independent)."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_torch as R
        from graphvqa_amd.parallel import allreduce_gradients
        torch.set_num_threads(1)
        gb, p, x, ea, ins, H = _case()
        nsl, emask, ei, b, (g0, g1) = shard_batch(gb.edge_index, gb.batch, gb.num_graphs, rank, world)
        params = {k: torch.nn.Parameter(v.double()) for k, v in tparams(p).items() if v.is_floating_point() and "running" not in k}
        full = dict(tparams(p, torch.float64))
        full.update(params)
        h = R.gat_seq(t(x[nsl]).double(), t(ei), t(ea[emask]).double(), t(ins[:, g0:g1]).double(), t(b), full, heads=H)
        h.square().sum().backward()
        # tiny buckets: several collectives, identical on every rank (lin_r aliases lin_l in the module; here it has its own grad)
        n = allreduce_gradients(params.values(), bucket_bytes=4096, average=False)
        if rank == 0:
            q.put((n, {k: v.grad.numpy() for k, v in params.items()}))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_matches_single_process():
    from oracle import ref_torch as R
    gb, p, x, ea, ins, H = _case()
    params = {k: torch.nn.Parameter(v.double()) for k, v in tparams(p).items() if v.is_floating_point() and "running" not in k}
    full = dict(tparams(p, torch.float64))
    full.update(params)
    R.gat_seq(t(x).double(), t(gb.edge_index), t(ea).double(), t(ins).double(), t(gb.batch), full, heads=H).square().sum().backward()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    ncalls, got = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert ncalls > 1
    for k, v in params.items():
        ref = np.zeros_like(got[k]) if v.grad is None else v.grad.numpy()
        assert np.abs(got[k] - ref).max() < 1e-9 * (1.0 + np.abs(ref).max()), k


def _direct_worker(rank, world, port, q):
    """The direct one-hop exchange (every rank pushes its rows to all peers in one grouped batch of isend / irecv) against the
    collective, on ragged per-rank row counts; blocking, in flight (async_op) and through the environment switch."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        counts = [3 + 2 * r for r in range(world)]                  # ragged: 3, 5, 7 ...
        rows = torch.arange(counts[rank] * 6, dtype=torch.float32).reshape(counts[rank], 6) + 1000.0 * rank
        want = torch.cat([torch.arange(c * 6, dtype=torch.float32).reshape(c, 6) + 1000.0 * r for r, c in enumerate(counts)])
        coll = all_gather_graph_rows(rows, counts=counts, algo="collective")
        direct = all_gather_graph_rows(rows, counts=counts, algo="direct")
        inflight = all_gather_graph_rows(rows, counts=None, algo="direct", async_op=True)      # (counts exchanged by the function)
        os.environ["GVQA_ALLGATHER"] = "direct"
        by_env = all_gather_graph_rows(rows, counts=counts)
        del os.environ["GVQA_ALLGATHER"]
        ok = all(torch.equal(v, want) for v in (coll, direct, inflight.wait(), by_env))
        bad = False
        try:
            all_gather_graph_rows(rows, counts=counts, algo="ring-of-fire")
        except ValueError:
            bad = True
        dist.barrier()
        if rank == 0:
            q.put((ok, bad, tuple(direct.shape)))
    finally:
        dist.destroy_process_group()


def test_direct_one_hop_all_gather_equals_the_collective_three_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker, args=(r, 3, port, q)) for r in range(3)]
    for pr in procs:
        pr.start()
    ok, bad, shape = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert ok and bad and shape == (15, 6)
