"""Parity of the HIP path (through the C ABI) with the oracle and the golden vectors.  GPU only.

Tolerance: north_star states <= 1e-4 max-abs in fp32 against the reference forward; the tests
assert that bound against the golden vectors (recorded from the reference's code) and a tighter
one against the fp64 oracle where sizes allow.
"""
import os
import numpy as np
import pytest
import torch

from graphvqa_amd import synth
from tests.util import load_golden, t, tparams, maxabs

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(autouse=True)
def _inference_mode():
    """These tests pin the fused inference kernels: with gradients enabled gat_seq.forward takes the differentiable
    formulation instead (tests/test_gpu_backward.py)."""
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _csr_reference(edge_index, N):
    src, dst = edge_index
    order = np.lexsort((np.arange(dst.shape[0]), dst))       # by dst, then original edge id
    rowptr = np.zeros(N + 1, np.int64)
    np.add.at(rowptr, dst + 1, 1)
    return np.cumsum(rowptr), src[order], order


@pytest.mark.parametrize("maker", ["small", "config2", "hub", "empty_edges"])
def test_graph_build_matches_numpy(dev, maker):
    from graphvqa_amd.graph import SceneGraphBatch
    if maker == "small":
        gb = synth.make_graph_batch(8, seed=21, nodes_lo=1, nodes_hi=12, rel_per_node=1.5)
    elif maker == "config2":
        gb = synth.config2_batch()
    elif maker == "hub":   # one graph, node 0 receives 3000 edges (exercises the in-row rank sort)
        E = 3000
        src = synth.randint(E, 5, 0, 50)
        ei = np.stack([src, np.zeros(E, np.int64)])
        gb = synth.GraphBatch(ei, np.zeros(50, np.int64), 1)
    else:
        gb = synth.GraphBatch(np.zeros((2, 0), np.int64), np.array([0, 0, 2, 2, 2], np.int64), 4)
    g = SceneGraphBatch(t(gb.edge_index, device=dev), t(gb.batch, device=dev), gb.num_nodes, gb.num_graphs)
    rowptr, csr_src, eid = _csr_reference(gb.edge_index, gb.num_nodes)
    assert np.array_equal(g.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(g.csr_eid.cpu().numpy(), eid)
    assert np.array_equal(g.csr_src.cpu().numpy(), csr_src)
    gp = np.searchsorted(gb.batch, np.arange(gb.num_graphs + 1))
    assert np.array_equal(g.graph_ptr.cpu().numpy(), gp)
    sizes = np.diff(gp)
    assert g.max_graph_nodes == (sizes.max() if len(sizes) else 0)
    assert g.max_in_degree == (np.diff(rowptr).max() if gb.num_nodes else 0)
    assert g.intra_graph


def test_graph_contract_violations(dev):
    from graphvqa_amd.graph import SceneGraphBatch
    from graphvqa_amd._lib import GvqaError
    ei = torch.tensor([[0, 1], [1, 7]], device=dev)
    with pytest.raises(GvqaError):
        SceneGraphBatch(ei, torch.zeros(3, dtype=torch.int64, device=dev), 3, 1)
    ei = torch.tensor([[0, 1], [1, 2]], device=dev)
    with pytest.raises(GvqaError):   # batch not sorted
        SceneGraphBatch(ei, torch.tensor([1, 0, 1], device=dev), 3, 2)
    g = SceneGraphBatch(ei, torch.tensor([0, 0, 1], device=dev), 3, 2)
    assert not g.intra_graph        # edge 1->2 crosses graphs


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (33, 20, 300), (130, 8, 512), (257, 1200, 300), (1000, 2048, 512),
                                   (64, 129, 37), (4096, 1208, 812)])
def test_linear_f32(dev, M, N, K):
    import ctypes as C
    from graphvqa_amd import _lib
    lib = _lib.load()
    A = t(synth.normal((M, K), 1), device=dev)
    B = t(synth.normal((N, K), 2), device=dev)
    bias = t(synth.normal((N,), 3), device=dev)
    out = torch.empty((M, N), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.gvqa_linear_f32(M, N, K, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), 1, out.data_ptr(), N, st))
    ref = torch.relu(A.double() @ B.double().T + bias.double())
    scale = float(ref.abs().max()) + 1.0
    assert maxabs(out, ref) < 2e-6 * scale * np.sqrt(K)
    # strided operands (sub-matrix of a wider weight, as the node half of lin_l)
    if K >= 8:
        K2 = K // 2 // 4 * 4 or 4
        out2 = torch.empty((M, N), device=dev)
        _lib.check(lib.gvqa_linear_f32(M, N, K2, A.data_ptr(), K, B.data_ptr(), K, None, 0, out2.data_ptr(), N, st))
        ref2 = A[:, :K2].double() @ B[:, :K2].double().T
        assert maxabs(out2, ref2) < 2e-6 * (float(ref2.abs().max()) + 1) * np.sqrt(K)


@pytest.mark.parametrize("M,N,K,ldc", [(20000, 20, 512, 20), (16385, 4, 96, 8), (65536 + 77, 32, 128, 32), (40000, 8, 300 // 32 * 32, 12), (59570, 20, 300, 20), (16400, 32, 100, 32), (20000, 12, 132, 12)])
def test_linear_tall_skinny_stream(dev, M, N, K, ldc):
    """The LDS-DMA streaming kernel for tall skinny plain products (k_linear_f32_skinny_dma: M >= 16384, N <= 32, N % 4 == 0, K % 32 == 0 --
    the all-hops edge logits [E, De] x [De, K H]): ragged last row tile, N from 4 to 32, result rows wider than N (the columns beyond stay
    untouched), against fp64."""
    from graphvqa_amd import _lib
    lib = _lib.load()
    A = t(synth.normal((M, K), 21), device=dev)
    B = t(synth.normal((N, K), 22), device=dev)
    out = torch.full((M, ldc), 7.0, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.gvqa_linear_f32(M, N, K, A.data_ptr(), K, B.data_ptr(), K, None, 0, out.data_ptr(), ldc, st))
    ref = A.double() @ B.double().T
    assert maxabs(out[:, :N], ref) < 2e-6 * (float(ref.abs().max()) + 1.0) * np.sqrt(K)
    if ldc > N:
        assert bool((out[:, N:] == 7.0).all())


@pytest.mark.parametrize("vendor", [0, 1])
def test_linear_large_epilogue_branches(dev, vendor):
    """Large products with no epilogue, an accumulate (separate or in place) or a bias: on the hand-written kernels
    (default) and on the OPT-IN vendor route (GVQA_OPT_VENDOR_GEMM: rocBLAS for plain / accumulate products, bias rows
    pre-written + beta = 1); all must agree with fp64 within the same tolerance."""
    from graphvqa_amd import _lib
    lib = _lib.load()
    M, N, K = 8192, 1024, 512                        # 8.6 GFLOP: above both vendor thresholds
    A = t(synth.normal((M, K), 11), device=dev)
    B = t(synth.normal((N, K), 12), device=dev)
    bias = t(synth.normal((N,), 13), device=dev)
    add = t(synth.normal((M, N + 8), 14), device=dev)            # wider addend rows (ld_add != ldc)
    st = torch.cuda.current_stream().cuda_stream
    ref = A.double() @ B.double().T
    tol = 2e-6 * (float(ref.abs().max()) + 1.0) * np.sqrt(K)
    out = torch.empty((M, N), device=dev)
    ex = lambda bias_, add_, ld_add, C_: _lib.check(lib.gvqa_linear_f32_ex(
        M, N, K, A.data_ptr(), K, B.data_ptr(), K, bias_, add_, ld_add, None, 0, 0, C_.data_ptr(), N, st))
    old = _lib.set_option(_lib.OPT_VENDOR_GEMM, vendor)
    try:
        assert (b"OPT-IN vendor" in lib.gvqa_gemm_backend()) == bool(vendor)
        ex(None, None, 0, out)
        assert maxabs(out, ref) < tol
        ex(bias.data_ptr(), None, 0, out)
        assert maxabs(out, ref + bias.double()) < tol
        ex(None, add.data_ptr(), N + 8, out)
        assert maxabs(out, ref + add[:, :N].double()) < tol
        acc = add[:, :N].contiguous()
        ex(None, acc.data_ptr(), N, acc)                              # in place: C += A.B^T
        assert maxabs(acc, ref + add[:, :N].double()) < tol
    finally:
        _lib.set_option(_lib.OPT_VENDOR_GEMM, old)
    assert b"gvqa::k_linear" in lib.gvqa_gemm_backend()


def _bf16_pieces(W, pieces):
    """The weight the bf16 matrix-core GEMM actually multiplies by: bf16(W) [+ bf16(W - bf16(W))]."""
    hi = W.bfloat16().float()
    return hi if pieces == 1 else hi + (W - hi).bfloat16().float()


@pytest.mark.parametrize("M,N,K", [(300, 200, 64), (129, 100, 72), (1, 8, 8), (1000, 512, 512), (4100, 1536, 1024),
                                   (5637, 1536, 576),                            # two pieces: the wide 256 x 128 kernel
                                   (8197, 1536, 1024), (33001, 516, 512),        # two pieces: the 256 x 256 kernel (edge tiles in M and N)
                                   (29785, 520, 544),                            # ... an odd number of its 32-deep K steps
                                   (29785, 512, 32), (29785, 512, 64)])          # ... one and two steps (shorter than its ring)
@pytest.mark.parametrize("pieces", [1, 2])
def test_linear_bf16(dev, M, N, K, pieces):
    """k_linear_bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate) against fp64 on the SAME bf16 operands:
    products of bf16 values are exact in fp32, so only the accumulation order differs."""
    from graphvqa_amd import _lib
    lib = _lib.load()
    A = t(synth.normal((M, K + 8), 21), device=dev).bfloat16()          # rows wider than K (lda != K)
    W = t(synth.normal((N, K), 22), device=dev)
    bias = t(synth.normal((N,), 23), device=dev)
    add = t(synth.normal((M, N), 24), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    Wpk = torch.empty((N, pieces * K), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.gvqa_pack_weight_bf16(N, K, pieces, W.data_ptr(), K, Wpk.data_ptr(), st))
    Weff = _bf16_pieces(W, pieces)
    assert torch.equal(Wpk[:, :K].float(), W.bfloat16().float())
    if pieces == 2:
        assert torch.equal(Wpk[:, :K].float() + Wpk[:, K:].float(), Weff)
        assert float((Weff - W).abs().max()) <= 2.0 ** -16 * float(W.abs().max())
    ref = A[:, :K].double() @ Weff.double().T
    tol = 2e-6 * (float(ref.abs().max()) + 1.0) * np.sqrt(K)
    out = torch.empty((M, N), device=dev)
    _lib.check(lib.gvqa_linear_bf16(M, N, K, pieces, A.data_ptr(), K + 8, Wpk.data_ptr(), None, None, 0, None, 0, 0,
                                    out.data_ptr(), N, 0, st))
    assert maxabs(out, ref) < tol
    # full epilogue, fp32 output: relu((acc + bias + addend) * mul)
    mul = t(synth.normal((M, N), 25), device=dev)
    _lib.check(lib.gvqa_linear_bf16(M, N, K, pieces, A.data_ptr(), K + 8, Wpk.data_ptr(), bias.data_ptr(), add.data_ptr(), N,
                                    mul.data_ptr(), N, 1, out.data_ptr(), N, 0, st))
    ref2 = torch.relu((ref + bias.double() + add.double()) * mul.double())
    assert maxabs(out, ref2) < tol * (1.0 + float(mul.abs().max()))
    # bf16 output with bf16 addend: equal to the rounded reference up to one bf16 ulp where rounding flips
    add16 = add.bfloat16()
    out16 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.gvqa_linear_bf16(M, N, K, pieces, A.data_ptr(), K + 8, Wpk.data_ptr(), bias.data_ptr(), add16.data_ptr(), N,
                                    None, 0, 0, out16.data_ptr(), N, 1, st))
    ref3 = ref + bias.double() + add16.double()
    assert maxabs(out16.float(), ref3) < 2.0 ** -8 * (float(ref3.abs().max()) + 1.0)
    assert float((out16.float().cpu() == ref3.float().bfloat16().float().cpu()).float().mean()) > 0.99


def test_mfma_stream_probe_runs_and_reports_its_flops(dev):
    """gvqa_mfma_stream (bench.py's `matrix_rate_measured`): the launch succeeds on random and on zero operands, reports cus x 8 x iters x 64 MFMAs
    of 2 x 32 x 32 x 16 flops, leaves finite sums in the sink (zeros for zero operands) and rejects a sink that is too small."""
    import ctypes
    from graphvqa_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    sink = torch.full((cus * 512,), float("nan"), device=dev)
    for bf in (0, 1):
        ops = torch.empty(1 << 16, dtype=torch.bfloat16 if bf else torch.float16, device=dev).normal_()
        fl = ctypes.c_int64(0)
        _lib.check(lib.gvqa_mfma_stream(ops.data_ptr(), ops.numel() * 2, sink.data_ptr(), sink.numel(), 3, bf, ctypes.byref(fl), st))
        torch.cuda.synchronize()
        assert fl.value == cus * 8 * 3 * 64 * 2 * 32 * 32 * 16
        assert bool(torch.isfinite(sink).all()) and float(sink.abs().max()) > 0
    zeros = torch.zeros(1 << 16, dtype=torch.float16, device=dev)
    _lib.check(lib.gvqa_mfma_stream(zeros.data_ptr(), zeros.numel() * 2, sink.data_ptr(), sink.numel(), 2, 0, None, st))
    torch.cuda.synchronize()
    assert float(sink.abs().max()) == 0.0
    assert lib.gvqa_mfma_stream(zeros.data_ptr(), zeros.numel() * 2, sink.data_ptr(), 16, 2, 0, None, st) != 0


def _load_module(m, params, dev):
    sd = {k: t(v) for k, v in params.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m.to(dev).eval()


def test_gat_conv_small_golden(dev):
    from graphvqa_amd.gat_skip import gat
    meta, g = load_golden("gat_conv_small")
    p = synth.gat_seq_params(20, 8, 16, 4, 1, 4, seed=meta["param_seed"])
    conv = gat(24, 8, 20, heads=4, concat=False, negative_slope=0.2, dropout=0.0, bias=True)
    _load_module(conv, {k[len("convs.0."):]: v for k, v in p.items() if k.startswith("convs.0.")}, dev)
    ei = t(g["edge_index"], device=dev)
    out, (ei2, alpha) = conv(t(g["x"], device=dev), ei, t(g["edge_attr"], device=dev), return_attention_weights=True)
    assert ei2 is ei
    assert maxabs(out, g["out"]) < TOL
    assert maxabs(alpha, g["alpha"]) < 1e-5
    out_only = conv(t(g["x"], device=dev), ei, t(g["edge_attr"], device=dev))
    assert torch.equal(out_only, out)      # deterministic, run to run


def _run_gat_seq(dev, dims, params, x, ei, ea, ins, batch, **kw):
    from graphvqa_amd.gat_skip import gat_seq
    dn, de, di, K, H = dims
    m = gat_seq(dn, dn, de, di, K, dropout=0.1, gat_heads=H)
    _load_module(m, params, dev)
    return m(t(x, device=dev), t(ei, device=dev), t(ea, device=dev), t(ins, device=dev), t(batch, device=dev), **kw)


def test_gat_seq_small_golden_all_hops(dev):
    meta, g = load_golden("gat_seq_small")
    dims = (meta["dn"], meta["de"], meta["di"], meta["K"], meta["heads"])
    p = synth.gat_seq_params(dims[0], dims[0], dims[1], dims[2], dims[3], dims[4], seed=meta["param_seed"])
    out, alpha, hops = _run_gat_seq(dev, dims, p, g["x"], g["edge_index"], g["edge_attr"], g["instr"], g["batch"],
                                    return_attention_weights=True, return_hops=True)
    assert maxabs(out, g["out"]) < TOL
    assert maxabs(hops, g["hs"]) < TOL
    assert maxabs(alpha, g["alphas"]) < 1e-5


@pytest.mark.parametrize("name", ["gat_seq_debug2_d300", "gat_seq_debug4_d300"])
def test_gat_seq_real_dims_golden(dev, name):
    meta, g = load_golden(name)
    s = meta["input_seeds"]
    N, E, B = g["batch"].shape[0], g["edge_index"].shape[1], int(g["batch"].max()) + 1
    x, ea = synth.normal((N, 300), s["x"]), synth.normal((E, 300), s["edge_attr"])
    ins = synth.normal((5, B, 512), s["instr"])
    p = synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=meta["param_seed"])
    out, alpha, hops = _run_gat_seq(dev, (300, 300, 512, 5, 4), p, x, g["edge_index"], ea, ins, g["batch"],
                                    return_attention_weights=True, return_hops=True)
    assert maxabs(out, g["out"]) < TOL
    assert maxabs(hops, g["hs"]) < TOL
    assert maxabs(alpha[0], g["alpha0"]) < 1e-5 and maxabs(alpha[4], g["alpha4"]) < 1e-5


def test_gat_concat_and_pair_inputs_golden(dev):
    """The parts of the reference's public `gat` class that gat_seq does not use -- concat=True (heads side by side, bias [H C]) and
    tuple in_channels with a pair (x_l, x_r) of node tensors (gat_skip.py:78-80,136-143,162-163) -- against outputs recorded from
    the reference's own class; also state_dict keys / shapes, and gradients flowing (the general form is differentiable)."""
    from graphvqa_amd.gat_skip import gat
    meta, g = load_golden("gat_conv_concat_pair")
    H, C, e_in = meta["heads"], meta["out_channels"], meta["edge_in"]
    base = {"lin_l.weight": g["p_lin_l_weight"], "lin_e.weight": g["p_lin_e_weight"], "att_l": g["p_att_l"], "att_r": g["p_att_r"], "att_e": g["p_att_e"]}
    x, xr, ei, ea = t(g["x"], device=dev), t(g["x_r"], device=dev), t(g["edge_index"], device=dev), t(g["edge_attr"], device=dev)
    conv = _load_module(gat(24, C, e_in, heads=H, concat=True, dropout=0.0), dict(base, **{"lin_r.weight": g["p_lin_l_weight"], "bias": g["p_bias_hc"]}), dev)
    out, (_, alpha) = conv(x, ei, ea, return_attention_weights=True)
    assert out.shape == (12, H * C) and maxabs(out, g["out_concat"]) < TOL and maxabs(alpha, g["alpha_concat"]) < 1e-5
    conv_b = gat((24, 16), C, e_in, heads=H, concat=False, dropout=0.0)
    assert conv_b.state_dict()["lin_r.weight"].shape == (H * C, 16) and conv_b.lin_r is not conv_b.lin_l
    _load_module(conv_b, dict(base, **{"lin_r.weight": g["p_lin_r_weight"], "bias": g["p_bias_c"]}), dev)
    out, (_, alpha) = conv_b((x, xr), ei, ea, return_attention_weights=True)
    assert maxabs(out, g["out_pair"]) < TOL and maxabs(alpha, g["alpha_pair"]) < 1e-5
    conv_c = _load_module(gat((24, 16), C, e_in, heads=H, concat=True, dropout=0.0), dict(base, **{"lin_r.weight": g["p_lin_r_weight"], "bias": g["p_bias_hc"]}), dev)
    assert maxabs(conv_c((x, xr), ei, ea), g["out_pair_concat"]) < TOL
    with torch.enable_grad():
        xg = x.clone().requires_grad_(True)
        conv_c((xg, xr), ei, ea).square().sum().backward()
        assert xg.grad is not None and torch.isfinite(xg.grad).all() and conv_c.lin_r.weight.grad is not None
    with pytest.raises(NotImplementedError):
        conv_c((x, xr[:5]), ei, ea)


@pytest.mark.parametrize("name", ["gat_seq_debug2_d300", "gat_seq_debug4_d300"])
def test_integration_md_stub_runs(dev, name):
    """INTEGRATION.md section 2 -- the ctypes stub a maintainer of the reference would copy in place of the body of
    gat_seq.forward (gat_skip.py:249-279) -- is extracted from the document and EXECUTED on the reference-recorded goldens.
    A stale structure in the document (round 3: two missing members) overruns the library's writes / reads garbage options;
    here it fails the build instead.  The module handed to the stub has the reference's attribute structure (convs / bns)."""
    from tests.util import integration_md_stub, ROOT
    from graphvqa_amd.gat_skip import gat_seq
    cwd = os.getcwd()
    os.chdir(ROOT)                       # the document loads the library by its repo-relative path
    try:
        ns = {}
        exec(compile(integration_md_stub(), "INTEGRATION.md#2", "exec"), ns)
    finally:
        os.chdir(cwd)
    meta, g = load_golden(name)
    s = meta["input_seeds"]
    N, E, B = g["batch"].shape[0], g["edge_index"].shape[1], int(g["batch"].max()) + 1
    x, ea = synth.normal((N, 300), s["x"]), synth.normal((E, 300), s["edge_attr"])
    ins = synth.normal((5, B, 512), s["instr"])
    p = synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=meta["param_seed"])
    mod = gat_seq(300, 300, 300, 512, 5, dropout=0.1, gat_heads=4)
    mod.load_state_dict(tparams(p))
    mod = mod.to(dev).eval()
    # canary bytes behind the caller-side structures would not help (the overrun was inside ctypes' own allocation), so the
    # check is the result itself plus a second call after unrelated allocations
    for _ in range(2):
        out = ns["gat_seq_forward"](mod, t(x, device=dev), t(g["edge_index"], device=dev), t(ea, device=dev),
                                    t(ins, device=dev), t(g["batch"], device=dev))
        torch.cuda.synchronize()
        assert maxabs(out, g["out"]) < TOL
        _ = [torch.empty(1 << 16, device=dev) for _ in range(4)]
    import ctypes as C
    from graphvqa_amd import _lib
    assert C.sizeof(ns["Graph"]) == C.sizeof(_lib.Graph) and C.sizeof(ns["Dims"]) == C.sizeof(_lib.GatDims)
    assert C.sizeof(ns["HopParams"]) == C.sizeof(_lib.GatConvParams)


@pytest.mark.parametrize("H,C", [(1, 16), (2, 24), (4, 100), (8, 8)])
def test_gat_seq_vs_fp64_oracle_heads(dev, H, C):
    """Other head counts / widths against the oracle evaluated in float64."""
    from oracle import ref_torch as R
    de, di, K = 12, 20, 3
    gb = synth.make_graph_batch(13, seed=77, nodes_lo=1, nodes_hi=30, rel_per_node=2.0)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=900 + H)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    out = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch)
    ref = R.gat_seq(t(x, torch.float64), t(gb.edge_index), t(ea, torch.float64), t(ins, torch.float64),
                    t(gb.batch), tparams(p, torch.float64), heads=H)
    assert maxabs(out, ref) < 2e-5


def test_tiled_and_general_kernels_agree(dev):
    """Same inputs through the LDS-tiled and the general CSR kernel (C ABI `force` switch)."""
    import ctypes as C_
    from graphvqa_amd import _lib
    from graphvqa_amd.graph import SceneGraphBatch
    lib = _lib.load()
    H, C = 4, 64
    gb = synth.make_graph_batch(50, seed=5, nodes_lo=1, nodes_hi=40, rel_per_node=2.0)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    g = SceneGraphBatch(t(gb.edge_index, device=dev), t(gb.batch, device=dev), N, B)
    xp = t(synth.normal((N, H * C), 1), device=dev)
    a_node = t(synth.normal((N, 2 * H), 2), device=dev)
    a_edge = t(synth.normal((E, H), 3), device=dev)
    T = t(synth.normal((B, C + H), 4), device=dev)
    skip = t(synth.normal((N, C), 5), device=dev)
    vec = [t(synth.uniform((C,), 10 + i, 0.5, 1.5), device=dev) for i in range(5)]
    outs, alphas = [], []
    ws = torch.empty(E * H * 4 + 256, dtype=torch.uint8, device=dev)
    for force in (1, 2):
        out = torch.empty((N, C), device=dev)
        alpha = torch.empty((E, H), device=dev)
        d = _lib.GatMpDesc()
        d.C, d.H, d.negative_slope, d.bn_eps = C, H, 0.2, 1e-5
        d.xp, d.a_node, d.a_edge = xp.data_ptr(), a_node.data_ptr(), a_edge.data_ptr()
        d.graph_term, d.graph_term_ld, d.skip = T.data_ptr(), C + H, skip.data_ptr()
        d.bias, d.bn_weight, d.bn_bias, d.bn_mean, d.bn_var = [v.data_ptr() for v in vec]
        d.out, d.alpha_out, d.force = out.data_ptr(), alpha.data_ptr(), force
        _lib.check(lib.gvqa_gat_message_passing(C_.byref(g.c), C_.byref(d), ws.data_ptr(), ws.numel(),
                                                torch.cuda.current_stream().cuda_stream))
        outs.append(out)
        alphas.append(alpha)
    assert maxabs(alphas[0], alphas[1]) < 1e-6
    assert maxabs(outs[0], outs[1]) < 1e-5
    # softmax property: alpha sums to one over the incoming edges of every node with in-edges
    s = torch.zeros(N, H, device=dev).index_add_(0, t(gb.edge_index[1], device=dev), alphas[0])
    deg = torch.bincount(t(gb.edge_index[1], device=dev), minlength=N)
    assert maxabs(s[deg > 0], torch.ones_like(s[deg > 0])) < 1e-5


def test_inter_graph_edges_fall_back_to_unfolded(dev):
    """An edge crossing graphs (never produced by the reference's collate) must still follow the
    reference's literal formula (instruction of batch[src] for the edge, batch[n] for nodes)."""
    from oracle import ref_torch as R
    H, C, de, di, K = 4, 16, 8, 12, 3
    gb = synth.make_graph_batch(4, seed=9, nodes_lo=3, nodes_hi=6, rel_per_node=1.0)
    ei = np.concatenate([gb.edge_index, np.array([[0], [gb.num_nodes - 1]])], axis=1)   # graph 0 -> graph 3
    N, E, B = gb.num_nodes, ei.shape[1], gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=42)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    out = _run_gat_seq(dev, (C, de, di, K, H), p, x, ei, ea, ins, gb.batch)
    ref = R.gat_seq(t(x), t(ei), t(ea), t(ins), t(gb.batch), tparams(p), heads=H)
    assert maxabs(out, ref) < TOL


def test_large_single_graph_uses_general_kernel(dev):
    """One 5000-node graph does not fit an LDS tile: general CSR kernel, same answer as the oracle."""
    from oracle import ref_torch as R
    H, C, de, di, K = 4, 32, 8, 12, 2
    gb = synth.make_graph_batch(1, seed=3, fixed_nodes=5000, fixed_rel=15000)
    N, E = gb.num_nodes, gb.num_edges
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=43)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, 1, di), 3)
    out = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch)
    ref = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H)
    assert maxabs(out, ref) < TOL


@pytest.fixture(params=["fused", "fused2", "fused-split3", "split2h+mp", "split3+mp", "f32+mp"])
def projection_mode(request):
    """The ways a hop runs (GVQA_OPT_HOP_FUSION / GVQA_OPT_PROJECTION): projection + aggregation as ONE kernel on the split2h
    arithmetic (default) or on split3, a split projection followed by the message-passing kernel, f32-input MFMA projection
    (k_linear_f32_dma / k_linear_f32) followed by the message-passing kernel."""
    from graphvqa_amd import _lib
    proj = {"fused": _lib.PROJECTION_SPLIT2H, "fused2": _lib.PROJECTION_SPLIT2H, "fused-split3": _lib.PROJECTION_SPLIT3, "split2h+mp": _lib.PROJECTION_SPLIT2H,
            "split3+mp": _lib.PROJECTION_SPLIT3, "f32+mp": _lib.PROJECTION_F32}[request.param]
    old_p = _lib.set_option(_lib.OPT_PROJECTION, proj)
    old_f = _lib.set_option(_lib.OPT_HOP_FUSION, 2 if request.param == "fused2" else 1 if request.param.startswith("fused") else 0)
    yield request.param
    _lib.set_option(_lib.OPT_PROJECTION, old_p)
    _lib.set_option(_lib.OPT_HOP_FUSION, old_f)


def test_config2_shape_vs_oracle(dev, projection_mode):
    """BASELINE config 2 (1k graphs, ~30 nodes / ~60 edges, exact reference dims) against the fp32 oracle."""
    from oracle import ref_torch as R
    gb = synth.config2_batch()
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=303)
    x, ea, ins = synth.normal((N, 300), 1), synth.normal((E, 300), 2), synth.normal((5, B, 512), 3)
    out, alpha, _ = _run_gat_seq(dev, (300, 300, 512, 5, 4), p, x, gb.edge_index, ea, ins, gb.batch,
                                 return_attention_weights=True)
    ref, _, alphas = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), return_all=True)
    assert maxabs(out, ref) < TOL
    assert maxabs(alpha[4], alphas[4]) < 2e-5


def test_config3_slice_vs_oracle_on_the_benchmarked_kernel(dev, projection_mode):
    """The kernel instance bench.py times -- k_gat_mp_tiled<4,2> at C = 512, H = 4 with 128-channel ranges (20 stages per
    graph), one block per graph -- and the hop projection at d = 512, against the fp32 oracle (/root/reference
    gat_skip.py:249-279 restated) on the first 64 graphs of the config-3 batch, K = 5."""
    import ctypes, os
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.graph import SceneGraphBatch
    nb, d = 64, 512
    gb = synth.config3_batch(nb)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(d, d, d, d, 5, 4, seed=777)
    x, ea, ins = synth.normal((N, d), 1), synth.normal((E, d), 2), synth.normal((5, B, d), 3)
    ref, _, alphas = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), return_all=True)
    g = SceneGraphBatch(t(gb.edge_index, device=dev), t(gb.batch, device=dev), N, B)
    plan = _lib.MpPlan()
    for parts in ("1", None) if not projection_mode.startswith("fused") else (None,):   # one block per graph, then the small-batch split
        old_parts = _lib.set_option(_lib.OPT_MP_PARTS, int(parts) if parts else 0)
        try:
            _lib.check(_lib.load().gvqa_gat_mp_plan(ctypes.byref(g.c), d, 4, ctypes.byref(plan)))
            out, alpha, _ = _run_gat_seq(dev, (d, d, d, 5, 4), p, x, gb.edge_index, ea, ins, gb.batch,
                                         return_attention_weights=True)
        finally:
            _lib.set_option(_lib.OPT_MP_PARTS, old_parts)
        assert plan.tiled == 1 and plan.channel_range == 128 and plan.accumulators == 2 and plan.stages_per_graph == 20
        assert plan.blocks_per_graph == (1 if parts else 4)
        assert (g.c.num_row_groups, g.c.max_row_group_edges) == (16, 512)        # 4 graphs of 32 nodes / 128 edges per row group
        assert maxabs(out, ref) < TOL
        assert maxabs(alpha[4], alphas[4]) < 2e-5


@pytest.mark.parametrize("H,C,de,di,lo,hi", [(4, 64, 24, 16, 1, 40), (4, 300, 20, 12, 20, 40), (1, 32, 8, 8, 1, 128), (2, 136, 16, 0, 60, 128),
                                             (8, 48, 12, 20, 5, 70), (4, 640, 16, 8, 20, 60), (2, 1040, 8, 0, 30, 90)])
@pytest.mark.parametrize("scheme", ["split2h", "split3"])
def test_fused_hop_kernel_on_ragged_batches(dev, scheme, H, C, de, di, lo, hi):
    """The fused hop (projection + aggregation in one kernel, csrc/split3.hip EPI 2) forced onto small ragged batches:
    row groups that are not full, graphs of 1 and of exactly 128 nodes, channel counts that do not fill the last column
    block, every supported head count, train-mode BatchNorm -- against the oracle and against the unfused kernels."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_seq
    K = 3
    gb = synth.make_graph_batch(23, seed=1000 + H * 7 + C, nodes_lo=lo, nodes_hi=hi, rel_per_node=1.6)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=300 + H)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    ref, hs, alphas = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H, return_all=True)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    old_p = _lib.set_option(_lib.OPT_PROJECTION, _lib.PROJECTION_SPLIT2H if scheme == "split2h" else _lib.PROJECTION_SPLIT3)
    old_f0 = _lib.load().gvqa_get_option(_lib.OPT_HOP_FUSION)
    try:
        _lib.set_option(_lib.OPT_HOP_FUSION, 1)
        _lib.prof_enable(True); _lib.prof_collect()
        out, alpha, hops = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch,
                                        return_attention_weights=True, return_hops=True)
        prof = _lib.prof_collect(); _lib.prof_enable(False)
        assert prof["alpha"][1] == K and prof["mp"][1] == 0            # the fused path ran: no message-passing launches
        _lib.set_option(_lib.OPT_COEFF_KERNEL, 1)                      # coefficients by the per-(node, head) kernel instead of the
        try:                                                           # row-group kernel: same operations, bit-identical
            _, alpha_g, _ = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch,
                                         return_attention_weights=True, return_hops=True)
        finally:
            _lib.set_option(_lib.OPT_COEFF_KERNEL, 0)
        assert torch.equal(alpha, alpha_g)
        _lib.set_option(_lib.OPT_HOP_FUSION, 0)
        out_u = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch)
        _lib.set_option(_lib.OPT_HOP_FUSION, 1)
        m = _load_module(gat_seq(C, C, de, di, K, dropout=0.0, gat_heads=H), p, dev).train()      # batch-statistics BatchNorm
        out_t = m(*[t(a, device=dev) for a in (x, gb.edge_index, ea, ins, gb.batch)])
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
        _lib.set_option(_lib.OPT_PROJECTION, old_p)
        _lib.set_option(_lib.OPT_HOP_FUSION, old_f0)
        _lib.prof_enable(False)
    assert maxabs(out, ref) < TOL and maxabs(hops, torch.stack(hs)) < TOL and maxabs(alpha, torch.stack(alphas)) < 2e-5
    assert maxabs(out, out_u) < 2e-5
    ref_t = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H, training_bn=True)
    assert maxabs(out_t, ref_t) < TOL


@pytest.mark.parametrize("H,C,K,seed,graphs,lo,hi,rel", [(2, 64, 3, 2, 1, 20, 20, 1.3), (4, 256, 3, 1, 3, 5, 30, 1.3), (2, 128, 4, 2, 1, 20, 20, 1.3),
                                                         (4, 260, 3, 1, 3, 5, 30, 1.3), (1, 132, 5, 3, 23, 2, 24, 1.3), (2, 300, 3, 5, 2, 100, 128, 1.3),
                                                         (4, 300, 5, 7, 200, 20, 40, 1.0), (8, 64, 4, 4, 5, 3, 30, 0.5), (4, 512, 5, 9, 40, 1, 12, 0.3)])
def test_chained_hops_on_small_and_sparse_row_groups(dev, H, C, K, seed, graphs, lo, hi, rel):
    """Chained persistent hops (GVQA_OPT_HOP_FUSION = 2, three or more hops: at least one hop reads AND leaves packed rows) on
    row groups with FEW edges -- one to three small graphs, config-2-like sparse graphs (E/N = 2).  The kernel's LDS sub-arrays
    are then short and its per-graph maxima (cleared before the main loop, flushed at the top of the next item) used to land
    inside the operand ring's third stage: garbage maxima -> a garbage output scale -> rows flushed to zero two hops later
    (found in round 3; batches with >= 282 edges per row group at H = 4 -- config 3 -- were never affected)."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    de, di = 16, 8
    gb = synth.make_graph_batch(graphs, seed=seed, nodes_lo=lo, nodes_hi=hi, rel_per_node=rel)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=500 + H)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    ref, _, alphas = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H, return_all=True)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    old_f = _lib.set_option(_lib.OPT_HOP_FUSION, 2)
    try:
        _lib.prof_enable(True); _lib.prof_collect()
        out, alpha, _ = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch, return_attention_weights=True)
        out2 = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch)
        prof = _lib.prof_collect()
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
        _lib.set_option(_lib.OPT_HOP_FUSION, old_f)
        _lib.prof_enable(False)
    assert prof["mp"][1] == 0 and prof["alpha"][1] == 2 * K          # the fused kernels ran, both calls
    assert maxabs(out, ref) < TOL and maxabs(alpha, torch.stack(alphas)) < 2e-5
    assert torch.equal(out, out2)


@pytest.mark.parametrize("layout", ["device", "loader"])
def test_row_groups_never_span_more_than_128_graph_ids(dev, layout):
    """the chained hop kernels index per-graph LDS state (scales, output maxima: 128 entries) by
    node_graph[row] - node_graph[first row of the group]; EMPTY graphs between two non-empty ones take ids without rows, so
    'at most 128 rows' did not imply 'at most 128 ids'.  Both planners now close a group at 128 ids; a batch whose few small
    graphs sit 200 ids apart runs the chained kernel (hop_fusion = 2) and is held to the oracle."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.graph import SceneGraphBatch, HostLayout
    from graphvqa_amd.gat_skip import gat_seq
    H, C, K, de, di = 4, 64, 4, 16, 8
    small = synth.make_graph_batch(9, seed=31, nodes_lo=3, nodes_hi=9, rel_per_node=1.5)
    ids = np.array([0, 1, 2, 203, 204, 470, 471, 472, 800])            # the 9 graphs keep their order, ids far apart
    B = 801
    batch = ids[small.batch]
    ei = small.edge_index
    N, E = small.num_nodes, small.num_edges
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=77)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    ref = R.gat_seq(t(x), t(ei), t(ea), t(ins), t(batch), tparams(p), heads=H)
    hl = HostLayout.from_numpy(ei, batch, B) if layout == "loader" else None
    g = SceneGraphBatch(t(ei, device=dev), t(batch, device=dev), N, B, host_layout=hl)
    G = g.c.num_row_groups
    assert G >= 4                                                        # 128 rows would have held all 9 graphs in one group
    rg = g._view(g.c.row_group_ptr, G + 1).cpu().numpy()
    ng = g.node_graph.cpu().numpy()
    for r in range(G):
        assert ng[rg[r + 1] - 1] - ng[rg[r]] < 128
    m = _load_module(gat_seq(C, C, de, di, K, dropout=0.0, gat_heads=H), p, dev)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    old_f = _lib.set_option(_lib.OPT_HOP_FUSION, 2)
    try:
        _lib.prof_enable(True); _lib.prof_collect()
        out = m(t(x, device=dev), t(ei, device=dev), t(ea, device=dev), t(ins, device=dev), t(batch, device=dev), graph=g)
        prof = _lib.prof_collect()
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
        _lib.set_option(_lib.OPT_HOP_FUSION, old_f)
        _lib.prof_enable(False)
    assert prof["mp"][1] == 0 and prof["alpha"][1] == K
    assert maxabs(out, ref) < TOL


def test_grouped_one_launch_csr_build_equals_the_general_build(dev):
    """gvqa_graph_build_grouped (loader-side layout + COO edges grouped by graph: one upload, one launch) must produce the
    general pair's arrays bit for bit -- rowptr, csr_src, csr_eid (in-row order by edge id, multi-edges), node_graph, graph_ptr,
    row groups, row order, statistics -- on random batches (empty graphs, single nodes, 128-node graphs, dense rows), also for
    the transposed graph; shapes outside its reach fall through to the general pair; an edge outside its group's node range
    (a layout that is not grouped after all) is caught by check_valid."""
    from graphvqa_amd import _lib
    from graphvqa_amd.graph import SceneGraphBatch, HostLayout
    rng = np.random.default_rng(4242)
    for case in range(40):
        B = int(rng.choice([1, 2, 7, 60, 300]))
        hi = int(rng.choice([1, 5, 40, 128]))
        sizes = rng.integers(0, hi + 1, size=B)
        if sizes.sum() == 0:
            sizes[0] = 1
        batch = np.repeat(np.arange(B), sizes).astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)])
        dens = float(rng.choice([0.0, 1.0, 4.0, 12.0]))
        src, dst = [np.zeros(0, np.int64)], [np.zeros(0, np.int64)]
        for g in range(B):
            n = int(sizes[g])
            if n == 0:
                continue
            e = int(rng.integers(0, int(dens * n) + 1))
            loops = np.arange(n) if rng.integers(0, 2) else np.zeros(0, np.int64)
            src.append(np.concatenate([loops, rng.integers(0, n, size=e)]) + offs[g])
            dst.append(np.concatenate([loops, rng.integers(0, n, size=e)]) + offs[g])
        ei = np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)
        N, E = int(batch.shape[0]), int(ei.shape[1])
        hl = HostLayout.from_numpy(ei, batch, B)
        assert hl.coo_grouped
        eit, bt = t(ei, device=dev), t(batch, device=dev)
        g_new = SceneGraphBatch(eit, bt, N, B, host_layout=hl)
        g_old = SceneGraphBatch(eit, bt, N, B, host_layout=HostLayout(hl.graph_ptr, hl.edge_ptr, hl.max_in_degree))      # general pair
        g_new.check_valid(); g_old.check_valid()
        for a, b_ in ((g_new, g_old), (g_new.transposed(), g_old.transposed())):
            for f in ("max_graph_nodes", "max_graph_edges", "max_in_degree", "intra_graph", "valid", "finalized", "num_row_groups",
                      "max_row_group_edges"):
                assert getattr(a.c, f) == getattr(b_.c, f), (case, f)
            torch.cuda.synchronize()
            for name in ("rowptr", "csr_src", "csr_eid", "node_graph", "graph_ptr"):
                assert torch.equal(getattr(a, name), getattr(b_, name)), (case, name)
            if a.c.num_row_groups:
                G = a.c.num_row_groups
                assert torch.equal(a._view(a.c.row_group_ptr, G + 1), b_._view(b_.c.row_group_ptr, G + 1)), case
                assert torch.equal(a._view(a.c.row_group_order, N), b_._view(b_.c.row_group_order, N)), case
    # a layout that claims grouping for a shuffled COO list: flagged on the device, reported by the deferred check
    gb = synth.make_graph_batch(60, seed=5, nodes_lo=3, nodes_hi=20, rel_per_node=1.5)       # several row groups
    perm = np.random.default_rng(1).permutation(gb.num_edges)
    ei_s = gb.edge_index[:, perm]
    good = HostLayout.from_numpy(ei_s, gb.batch, gb.num_graphs)
    assert not good.coo_grouped
    g_ok = SceneGraphBatch(t(ei_s, device=dev), t(gb.batch, device=dev), gb.num_nodes, gb.num_graphs, host_layout=good).check_valid()
    lie = HostLayout(good.graph_ptr, good.edge_ptr, good.max_in_degree, coo_grouped=True)
    with pytest.raises(_lib.GvqaError):
        SceneGraphBatch(t(ei_s, device=dev), t(gb.batch, device=dev), gb.num_nodes, gb.num_graphs, host_layout=lie).check_valid()
    # an edge between two graphs of ONE row group (a group packs several graphs) passes the node-range check; the
    # handle says intra_graph = 1 and the fused hops rely on it -> must be flagged for check_valid, like the general build's flag
    sm = synth.make_graph_batch(6, seed=7, nodes_lo=4, nodes_hi=9, rel_per_node=1.0)          # 6 small graphs = one row group
    assert sm.num_nodes <= 128
    gp = np.searchsorted(sm.batch, np.arange(sm.num_graphs + 1))
    ei_x = sm.edge_index.copy()
    k = int(np.nonzero(sm.batch[ei_x[1]] == 2)[0][0])              # an in-edge of graph 2 ...
    ei_x[0, k] = gp[4]                                               # ... now comes from a node of graph 4 (COO position unchanged)
    claim = HostLayout.from_numpy(sm.edge_index, sm.batch, sm.num_graphs)       # the layout of the clean batch: still "grouped"
    assert claim.coo_grouped
    gx = SceneGraphBatch(t(ei_x, device=dev), t(sm.batch, device=dev), sm.num_nodes, sm.num_graphs, host_layout=claim)
    assert gx.c.num_row_groups == 1 and gx.intra_graph
    with pytest.raises(_lib.GvqaError, match="joins two graphs"):
        gx.check_valid()
    SceneGraphBatch(t(sm.edge_index, device=dev), t(sm.batch, device=dev), sm.num_nodes, sm.num_graphs, host_layout=claim).check_valid()
    # a graph of more than 128 nodes: out of the grouped build's reach -> the general pair, silently
    big = synth.make_graph_batch(2, seed=6, nodes_lo=150, nodes_hi=200, rel_per_node=1.0)
    hb = HostLayout.from_numpy(big.edge_index, big.batch, big.num_graphs)
    gbig = SceneGraphBatch(t(big.edge_index, device=dev), t(big.batch, device=dev), big.num_nodes, big.num_graphs, host_layout=hb).check_valid()
    assert hb.coo_grouped and gbig.c.num_row_groups == 0 and gbig.c.max_graph_nodes >= 150


def test_default_rule_takes_the_one_launch_aggregate_first_form_at_config3(dev):
    """Under DEFAULT options the benchmark's shape (2048 graphs x 32 nodes, d = 512, H = 4, K = 5: 512 row groups = two whole rounds of
    256 CUs) runs as: layout pass (+ hop 0's node logits), then ONE launch for the five hops -- no coefficient kernel, no
    message-passing launch; a forward that returns the attention weights takes one launch per hop, coefficients in the hop kernels'
    prologues, and gives the same rows; alpha sums to one per destination and head."""
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.graph import SceneGraphBatch
    d, H, K = 512, 4, 5
    gb = synth.config3_batch()
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(d, d, d, d, K, H, seed=777)
    x, ea, ins = synth.normal((N, d), 1), synth.normal((E, d), 2), synth.normal((K, B, d), 3)
    assert _lib.load().gvqa_get_option(_lib.OPT_HOP_FUSION) == 3
    m = _load_module(gat_seq(d, d, d, d, K, dropout=0.1, gat_heads=H), p, dev)
    args = [t(a, device=dev) for a in (x, gb.edge_index, ea, ins, gb.batch)]
    g = SceneGraphBatch(args[1], args[4], N, B)
    assert m.hop_kernel(g) == "aggregate_first_seq"
    first = m(*args)                                   # prepares the weight cache
    _lib.prof_enable(True); _lib.prof_collect()
    try:
        out = m(*args)
        prof = _lib.prof_collect()
        out_a, alpha, _ = m(*args, return_attention_weights=True)
        prof_a = _lib.prof_collect()
    finally:
        _lib.prof_enable(False)
    assert prof["mp"][1] == 0 and prof["alpha"][1] == 0 and prof["proj"][1] == 1 and prof["pack"][1] == 1 and prof["node_logit"][1] == 0, prof
    assert prof_a["mp"][1] == 0 and prof_a["alpha"][1] == 0 and prof_a["proj"][1] == K and prof_a["pack"][1] == 1, prof_a
    assert torch.equal(first, out)
    assert maxabs(out, out_a) < 2e-5
    a_last = alpha[-1]
    sums = torch.zeros(N, H, device=dev).index_add_(0, args[1][1], a_last)
    assert maxabs(sums, torch.ones_like(sums)) < 1e-5


def test_default_rule_takes_the_chained_kernel_on_a_large_sparse_batch(dev):
    """The regime the round-3 defect lived in, under DEFAULT options: 3400 config-2-like graphs (20-40 nodes, E/N = 2: about
    250 edges per row group) at d = 256 / H = 4 / K = 4 give six or more items per workgroup slot, so GVQA_OPT_HOP_FUSION = 3
    picks the persistent chained kernel by itself (one pack pass, no message-passing launches); every row against the oracle."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    H, C, de, di, K = 4, 256, 16, 32, 4
    gb = synth.make_graph_batch(3400, seed=0x5EED0002, nodes_lo=20, nodes_hi=40, rel_per_node=1.0)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=61)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    assert _lib.load().gvqa_get_option(_lib.OPT_HOP_FUSION) == 3
    from graphvqa_amd.gat_skip import gat_seq
    m = _load_module(gat_seq(C, C, de, di, K, dropout=0.1, gat_heads=H), p, dev)
    args = [t(a, device=dev) for a in (x, gb.edge_index, ea, ins, gb.batch)]
    first = m(*args)                                   # prepares the weight cache
    _lib.prof_enable(True); _lib.prof_collect()
    try:
        out = m(*args)
        prof = _lib.prof_collect()
    finally:
        _lib.prof_enable(False)
    # chained: ONE pack pass (hop 0), a coefficient and a hop launch per hop, no message-passing launches
    assert prof["mp"][1] == 0 and prof["alpha"][1] == K and prof["proj"][1] == K and prof["pack"][1] == 1, prof
    assert torch.equal(first, out)
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))      # (100k nodes on the CPU oracle; restored: hundreds of threads
    try:                                                              #  make the small cases of the tests that follow crawl)
        ref = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H)
    finally:
        torch.set_num_threads(threads)
    assert maxabs(out, ref) < TOL


def test_batch_sixteen_times_the_benchmark_keeps_its_indices(dev):
    """scripts/check_large_batch.py at 32768 graphs (1 M nodes, 4.2 M edges, d = 512: N H C = 2^31 elements, packed operands and
    edge tensors of several GB) through the default eval path, with and without a loader-side layout: first / middle / last
    64-graph windows against the oracle.  (131072 graphs -- 4.2 M nodes, 16.8 M edges, 165 ms -- checked by hand in round 3.)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_large_batch.py")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GRAPHS="32768"))
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["N*H*C"] == 2 ** 31 and res["read_back_max_abs_err"] < TOL and res["host_layout_max_abs_err"] < TOL, res


def test_randomised_parity_sweep(dev):
    """A fixed-seed sample of tests/fuzz.py: random head counts, widths, hop counts, batch shapes (single-node graphs to 128-node
    graphs, sparse to dense, 1 to 300 graphs), every hop kernel, library products forced or not, with and without the attention /
    per-hop outputs -- each against the oracle (scripts/fuzz_gat_seq.py runs any number of further cases)."""
    from tests.fuzz import case, run
    rng = np.random.default_rng(20260929)
    bad = []
    for _ in range(90):
        c = case(rng)
        ok, errs, sz = run(c, dev)
        if not ok:
            bad.append((c, errs, sz))
    assert not bad, bad[:3]


@pytest.mark.parametrize("stratum", ["tiny_groups", "sparse_e_over_n_1", "hubs", "partial_k_block", "many_small", "big_graphs"])
@pytest.mark.parametrize("fusion", [0, 1, 2, 3])
def test_stratified_parity_sweep(dev, fusion, stratum):
    """The parity net sized to the kernels' state space: every hop kernel (0 GEMM + message passing, 1 the
    8-wave fused kernel, 2 the persistent chained kernel, 3 the default rule) x every batch regime of tests/fuzz.STRATA x K = 1..5,
    five random cases each = 25 cases per test, 600 in all, library products forced (size threshold 0) so that the named kernel
    is the one that runs; each case against the oracle at 1e-4 (alpha 5e-5), twice (fresh and cached weights)."""
    from tests.fuzz import stratified_case, run, STRATA
    rng = np.random.default_rng(1000 * fusion + STRATA.index(stratum) + 20260930)
    bad = []
    for K in range(1, 6):
        for _ in range(5):
            c = stratified_case(rng, fusion, stratum, K)
            ok, errs, sz = run(c, dev)
            if not ok:
                bad.append((c, errs, sz))
    assert not bad, (len(bad), bad[:3])


@pytest.mark.parametrize("stratum", ["tiny_groups", "sparse_e_over_n_1", "hubs", "partial_k_block", "many_small", "big_graphs"])
def test_aggregate_first_hop_parity_sweep(dev, stratum):
    """GVQA_OPT_HOP_FUSION = 4, the aggregate-first hop kernel (csrc/hopagg.hip: heads concatenated along K, the attention-weighted
    neighbour sum formed inside the matrix-core loop, rows chunk-major between hops, per-output-column weight scales): H = 4 cases
    of every batch regime x K = 1..5, three cases each, against the oracle at 1e-4 (alpha 5e-5); widths 32..512 take the kernel
    (both wave layouts: <= 320 and <= 512 columns), narrower ones its fallback.  The kernel must actually have run."""
    from tests.fuzz import stratified_case, run, STRATA
    from graphvqa_amd import _lib
    rng = np.random.default_rng(4000 + STRATA.index(stratum))
    bad, ran = [], 0
    for K in range(1, 6):
        for rep in range(3):
            c = stratified_case(rng, 4, stratum, K)
            c["H"] = 4
            if rep == 0:
                c["C"] = [512, 300, 64, 320, 128][K - 1]          # every K once on a wide width
                c["di"] = 8
            if c["C"] % 4:
                c["C"] = 64
            _lib.prof_enable(True); _lib.prof_collect()
            ok, errs, sz = run(c, dev)
            pr = _lib.prof_collect(); _lib.prof_enable(False)
            if c["C"] >= 32:
                assert pr["mp"][1] == 0 and pr["proj"][1] == 2 * K and pr["node_logit"][1] == 0 and pr["alpha"][1] == 0, (c, pr)      # 2 forwards: K hop kernels each, no coefficient kernel, nothing unfused
                ran += 1
            if not ok:
                bad.append((c, errs, sz))
    assert ran >= 10 and not bad, (ran, len(bad), bad[:3])


@pytest.mark.parametrize("stratum", ["tiny_groups", "sparse_e_over_n_1", "hubs", "partial_k_block", "many_small", "big_graphs"])
def test_aggregate_first_one_launch_parity_sweep(dev, stratum):
    """GVQA_OPT_HOP_FUSION = 5: the K hops of the aggregate-first kernel as ONE launch (k_hopagg4<..., SEQ>: a workgroup walks all
    hops of its row group, rows ping-pong chunk-major between two buffers, the coefficients of hops 1 .. K - 1 -- node logits out
    of the accumulator registers, leaky-relu, segment softmax -- computed inside the workgroup).  H = 4 cases of every batch regime
    x K = 2..5 against the oracle at 1e-4; the plain-output forward must have been ONE hop launch + one coefficient launch, and a
    forward that asks for attention weights / per-hop rows the per-hop launches (same numbers)."""
    from tests.fuzz import stratified_case, run, STRATA
    from graphvqa_amd import _lib
    rng = np.random.default_rng(5000 + STRATA.index(stratum))
    bad, ran = [], 0
    for K in range(2, 6):
        for rep in range(3):
            c = stratified_case(rng, 5, stratum, K)
            c["H"] = 4
            if rep == 0:
                c["C"] = [300, 512, 64, 320][K - 2]               # every K once on a wide width
                c["di"] = 8
            if c["C"] % 4:
                c["C"] = 64
            extra = c["alpha"] or c["hops"]
            _lib.prof_enable(True); _lib.prof_collect()
            ok, errs, sz = run(c, dev)
            pr = _lib.prof_collect(); _lib.prof_enable(False)
            if c["C"] >= 32:
                # two forwards: the plain one = 1 hop launch; the other the same, or K hop launches when it returns more (both forms
                # compute their coefficients inside the hop kernels: no coefficient launch)
                assert pr["mp"][1] == 0 and pr["node_logit"][1] == 0 and pr["proj"][1] == (K + 1 if extra else 2) and pr["alpha"][1] == 0, (c, pr)
                ran += 1
            if not ok:
                bad.append((c, errs, sz))
    assert ran >= 8 and not bad, (ran, len(bad), bad[:3])


def _ragged_batch(rng, B, lo, hi, dens):
    sizes = rng.integers(lo, hi + 1, size=B)
    if sizes.sum() == 0:
        sizes[0] = 1
    batch = np.repeat(np.arange(B), sizes).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    src, dst = [np.zeros(0, np.int64)], [np.zeros(0, np.int64)]
    for g in range(B):
        n = int(sizes[g])
        if n == 0:
            continue
        e = int(rng.integers(0, int(dens * n) + 1))
        src.append(np.concatenate([np.arange(n), rng.integers(0, n, size=e)]) + offs[g])      # one self loop per node first, then random intra-graph edges
        dst.append(np.concatenate([np.arange(n), rng.integers(0, n, size=e)]) + offs[g])
    return np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64), batch, sizes


def test_default_rule_takes_the_packed_one_launch_form_at_config2(dev):
    """BASELINE config 2 under DEFAULT options (round 6): 1000 graphs of 20..40 nodes cut in order are 262 row groups -- two rounds of
    one-workgroup-per-CU launches on 256 CUs -- so finalize leaves the packed numbering (<= 256 fuller groups: one round) and the
    default rule runs the d = 300 forward as the layout pass + ONE aggregate-first launch on it.  Against the oracle at 1e-4 (plain
    output, attention weights, per-hop rows), and bit for bit against the in-order groups."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.graph import SceneGraphBatch, HostLayout
    gb = synth.config2_batch()
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=303)
    x, ea, ins = synth.normal((N, 300), 1), synth.normal((E, 300), 2), synth.normal((5, B, 512), 3)
    ref, hs, alphas = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), return_all=True)
    m = _load_module(gat_seq(300, 300, 300, 512, 5, dropout=0.1, gat_heads=4), p, dev)
    args = [t(a, device=dev) for a in (x, gb.edge_index, ea, ins, gb.batch)]
    lib = _lib.load()
    assert lib.gvqa_get_option(_lib.OPT_HOP_FUSION) == 3 and lib.gvqa_get_option(_lib.OPT_PACKED_GROUPS) == 1
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    for g in (SceneGraphBatch(args[1], args[4], N, B, host_layout=HostLayout.from_numpy(gb.edge_index, gb.batch, B)), SceneGraphBatch(args[1], args[4], N, B)):
        if -(-g.c.num_row_groups // cus) > -(-N // 128 // cus) and cus == 256:
            assert 0 < g.c.pk_num_row_groups <= cus < g.c.num_row_groups, (g.c.pk_num_row_groups, g.c.num_row_groups)
            assert m.hop_kernel(g) == "aggregate_first_seq"
        m(*args, graph=g)
        _lib.prof_enable(True); _lib.prof_collect()
        try:
            out = m(*args, graph=g)
            prof = _lib.prof_collect()
        finally:
            _lib.prof_enable(False)
        if g.c.pk_num_row_groups:
            assert prof["mp"][1] == 0 and prof["alpha"][1] == 0 and prof["proj"][1] == 1 and prof["pack"][1] == 1, prof
        out_a, alpha, hops = m(*args, graph=g, return_attention_weights=True, return_hops=True)
        assert maxabs(out, ref) < TOL and maxabs(out_a, ref) < TOL and maxabs(hops, torch.stack(hs)) < TOL
        assert maxabs(alpha, torch.stack(alphas)) < 2e-5
        if g.c.pk_num_row_groups:
            old = _lib.set_option(_lib.OPT_PACKED_GROUPS, 0)
            try:
                m.hop_fusion = 5
                assert torch.equal(m(*args, graph=g), out)
            finally:
                _lib.set_option(_lib.OPT_PACKED_GROUPS, old)
                m.hop_fusion = None


def test_packed_row_groups_plan_is_a_row_permuted_copy_of_the_csr(dev):
    """gvqa_graph::pk_* (round 6, graph.hip plan_packed / k_build_packed): with GVQA_OPT_PACKED_GROUPS = 2 every ragged batch whose graphs
    pack into fewer row groups gets the packed numbering.  Checked against numpy, exactly: pk_node_old is a permutation that keeps
    every graph's nodes consecutive and in order; groups hold whole graphs, <= 128 nodes and <= 1024 in-edges; every packed CSR row is
    the original row slot for slot (same COO edge ids in the same order, sources renumbered); pk_node_graph / pk_graph_old name the
    graph.  Both build paths (one-launch grouped build, general build + host finalize, general build + device finalize)."""
    from graphvqa_amd import _lib
    from graphvqa_amd.graph import SceneGraphBatch, HostLayout
    rng = np.random.default_rng(606)
    old = _lib.set_option(_lib.OPT_PACKED_GROUPS, 2)
    packed_cases = 0
    try:
        for case in range(36):
            B = int(rng.choice([3, 40, 300, 1000]))
            lo, hi = [(20, 40), (1, 128), (0, 9), (60, 70), (100, 128)][case % 5]
            ei, batch, sizes = _ragged_batch(rng, B, lo, hi, float(rng.choice([0.0, 1.0, 3.0, 7.0])))
            N, E = int(batch.shape[0]), int(ei.shape[1])
            hl = HostLayout.from_numpy(ei, batch, B)
            eit, bt = t(ei, device=dev), t(batch, device=dev)
            gs = [SceneGraphBatch(eit, bt, N, B, host_layout=hl),
                  SceneGraphBatch(eit, bt, N, B, host_layout=HostLayout(hl.graph_ptr, hl.edge_ptr, hl.max_in_degree)),
                  SceneGraphBatch(eit, bt, N, B)]
            torch.cuda.synchronize()
            G1s = {g.c.pk_num_row_groups for g in gs}
            assert len(G1s) == 1, (case, G1s)
            for g in gs:
                G0, G1 = g.c.num_row_groups, g.c.pk_num_row_groups
                if G1 == 0:
                    continue
                assert 0 < G1 < G0 and G1 >= -(-N // 128), (case, G0, G1)
                packed_cases += 1
                v = lambda p, n: g._view(p, n).cpu().numpy().astype(np.int64)
                gp, rp, cs, ce = v(g.c.pk_row_group_ptr, G1 + 1), v(g.c.pk_rowptr, N + 1), v(g.c.pk_csr_src, E), v(g.c.pk_csr_eid, E)
                ng, no, go = v(g.c.pk_node_graph, N), v(g.c.pk_node_old, N), v(g.c.pk_graph_old, B)
                rp0, cs0, ce0 = (a.cpu().numpy().astype(np.int64) for a in (g.rowptr, g.csr_src, g.csr_eid))
                assert np.array_equal(np.sort(no), np.arange(N)) and np.array_equal(np.sort(go), np.arange(B)), case
                assert gp[0] == 0 and gp[-1] == N and np.all(np.diff(gp) > 0) and np.all(np.diff(gp) <= 128), case
                assert np.array_equal(batch[no], go[ng]), case                       # the packed graph index names the node's graph
                assert np.all(np.diff(ng) >= 0), case                                 # graphs consecutive in the packed order ...
                same = ng[1:] == ng[:-1]
                assert np.all(no[1:][same] == no[:-1][same] + 1), case                # ... their nodes in order
                # a group holds whole graphs: no graph index on both sides of a cut
                cuts = gp[1:-1]
                assert np.all(ng[cuts] != ng[cuts - 1]), case
                assert rp[0] == 0 and rp[-1] == E and np.array_equal(np.diff(rp), np.diff(rp0)[no]), case
                assert g.c.pk_max_row_group_edges == int(np.max(rp[gp[1:]] - rp[gp[:-1]])) <= 1024, case
                old2new = np.empty(N, np.int64); old2new[no] = np.arange(N)
                # slot for slot: packed row n = original row no[n]
                idx0 = np.concatenate([np.arange(rp0[o], rp0[o + 1]) for o in no]) if E else np.zeros(0, np.int64)
                assert np.array_equal(ce, ce0[idx0]) and np.array_equal(cs, old2new[cs0[idx0]]), case
    finally:
        _lib.set_option(_lib.OPT_PACKED_GROUPS, old)
    assert packed_cases >= 20, packed_cases


def test_packed_row_groups_are_not_built_where_they_cannot_serve(dev):
    """No packed numbering when GVQA_OPT_PACKED_GROUPS = 0, when a graph alone exceeds the aggregate-first kernel's CSR slice (1024
    in-edges), or when no order of the graphs saves a row group; mode 1 (default) builds it only when it saves a ROUND of workgroups."""
    from graphvqa_amd import _lib
    from graphvqa_amd.graph import SceneGraphBatch, HostLayout
    rng = np.random.default_rng(9)
    ei, batch, sizes = _ragged_batch(rng, 300, 20, 40, 1.0)
    N, B = int(batch.shape[0]), int(sizes.shape[0])
    mk = lambda e_, b_, n_, B_: SceneGraphBatch(t(e_, device=dev), t(b_, device=dev), n_, B_, host_layout=HostLayout.from_numpy(e_, b_, B_))
    old = _lib.set_option(_lib.OPT_PACKED_GROUPS, 0)
    try:
        assert mk(ei, batch, N, B).c.pk_num_row_groups == 0
        _lib.set_option(_lib.OPT_PACKED_GROUPS, 2)
        g = mk(ei, batch, N, B)
        assert 0 < g.c.pk_num_row_groups < g.c.num_row_groups
        _lib.set_option(_lib.OPT_PACKED_GROUPS, 1)        # ~80 in-order groups -> ~70 packed: one round of 256 CUs either way
        assert mk(ei, batch, N, B).c.pk_num_row_groups == 0
        _lib.set_option(_lib.OPT_PACKED_GROUPS, 2)
        # one graph with 1100 in-edges among small ones
        n0 = 40
        big = np.stack([rng.integers(0, n0, size=1100), rng.integers(0, n0, size=1100)]).astype(np.int64)
        ei2 = np.concatenate([big, ei + n0], axis=1)
        batch2 = np.concatenate([np.zeros(n0, np.int64), batch + 1])
        assert mk(ei2, batch2, N + n0, B + 1).c.pk_num_row_groups == 0
        # full groups already: 64 graphs of 32 nodes
        gb = synth.config3_batch(64)
        assert mk(gb.edge_index, gb.batch, gb.num_nodes, gb.num_graphs).c.pk_num_row_groups == 0
    finally:
        _lib.set_option(_lib.OPT_PACKED_GROUPS, old)


@pytest.mark.parametrize("C,de,di,lo,hi,dens", [(64, 16, 12, 20, 40, 1.5), (300, 40, 512, 20, 40, 1.5), (512, 24, 32, 20, 40, 1.5), (128, 8, 8, 0, 50, 0.5), (64, 8, 0, 1, 128, 3.0)])
def test_packed_row_groups_forward_equals_the_in_order_groups_bit_for_bit(dev, C, de, di, lo, hi, dens):
    """The aggregate-first hops (GVQA_OPT_HOP_FUSION = 4 per-hop launches with attention weights and per-hop rows, 5 one launch) on the
    packed row groups: same graphs, same per-node edge order, same per-graph scales -- so the output must EQUAL the in-order groups'
    (GVQA_OPT_PACKED_GROUPS = 0 at forward time) bit for bit, and both sit within 1e-4 of the oracle (gat_skip.py:249-279)."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.graph import SceneGraphBatch, HostLayout
    H, K = 4, 3
    rng = np.random.default_rng(77 + C)
    ei, batch, sizes = _ragged_batch(rng, 90 if hi <= 50 else 160, lo, hi, dens)      # (lo = 0: empty graphs take ids without rows; hi = 128: graphs that fill a group alone)
    N, E, B = int(batch.shape[0]), int(ei.shape[1]), int(sizes.shape[0])
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=78)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, max(di, 1)), 3)[:, :, :di]
    ref, hops_ref, alphas = R.gat_seq(t(x), t(ei), t(ea), t(ins), t(batch), tparams(p), heads=H, return_all=True)
    m = _load_module(gat_seq(C, C, de, di, K, dropout=0.1, gat_heads=H), p, dev)
    args = [t(a, device=dev) for a in (x, ei, ea, ins, batch)]
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    old_p = _lib.set_option(_lib.OPT_PACKED_GROUPS, 2)
    try:
        g = SceneGraphBatch(args[1], args[4], N, B, host_layout=HostLayout.from_numpy(ei, batch, B))
        assert 0 < g.c.pk_num_row_groups < g.c.num_row_groups
        res = {}
        for packed in (2, 0):
            _lib.set_option(_lib.OPT_PACKED_GROUPS, packed)
            for fusion in (5, 4):
                m.hop_fusion = fusion
                _lib.prof_enable(True); _lib.prof_collect()
                out = m(*args, graph=g)
                pr = _lib.prof_collect(); _lib.prof_enable(False)
                assert pr["mp"][1] == 0 and pr["alpha"][1] == 0 and pr["proj"][1] == (1 if fusion == 5 else K), (packed, fusion, pr)
                out_a, alpha, hops = m(*args, graph=g, return_attention_weights=True, return_hops=True)
                res[(packed, fusion)] = (out, out_a, alpha, hops)
                assert maxabs(out, ref) < TOL and maxabs(out_a, ref) < TOL, (packed, fusion)
                assert maxabs(alpha, torch.stack(alphas)) < 5e-5 and maxabs(hops, torch.stack(hops_ref)) < TOL, (packed, fusion)
        for fusion in (5, 4):
            for a, b_ in zip(res[(2, fusion)], res[(0, fusion)]):
                assert torch.equal(a, b_), fusion
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
        _lib.set_option(_lib.OPT_PACKED_GROUPS, old_p)


@pytest.mark.parametrize("shape", ["shard256_d512", "config2_like_d300", "ragged_d64", "tiny", "isolated_nodes", "dense_row_groups", "one_hop", "no_instructions"])
def test_chained_hops_with_in_kernel_coefficients(dev, shape):
    """GVQA_OPT_HOP_COEFFS = 1 (default 2: by batch size): the chained 8-wave hop kernel computes its attention coefficients itself (csrc/split3.hip,
    CHN = 2: partial node logits left by the previous hop's column blocks summed per row group, edge halves gathered through the CSR
    edge ids, leaky-relu + segment softmax in LDS, gat_skip.py:180-190) -- no coefficient launch, one pack pass per forward -- against
    the oracle, against the coefficient-kernel form (GVQA_OPT_HOP_COEFFS = 0) and with the attention weights returned.  Row groups
    beyond the kernel's LDS capacity (522 edges at H = 4) must take the coefficient kernels by themselves."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_seq
    H = 4
    C, de, di, K, graphs, lo, hi, rel = {"shard256_d512": (512, 64, 64, 5, 256, 32, 32, 3.0), "config2_like_d300": (300, 40, 512, 5, 300, 20, 40, 1.0),
                                         "ragged_d64": (64, 16, 12, 4, 37, 1, 60, 1.7), "tiny": (32, 8, 8, 3, 2, 1, 5, 1.0),
                                         "isolated_nodes": (128, 8, 8, 3, 40, 3, 30, 0.0), "dense_row_groups": (64, 8, 8, 3, 6, 100, 128, 6.0),
                                         "one_hop": (256, 8, 16, 1, 20, 10, 40, 2.0), "no_instructions": (68, 8, 0, 3, 25, 5, 50, 1.5)}[shape]
    gb = synth.make_graph_batch(graphs, seed=0x1C0 + len(shape), nodes_lo=lo, nodes_hi=hi, rel_per_node=rel)
    ei = gb.edge_index
    if shape == "isolated_nodes":                     # every third node loses ALL its in-edges (self-loop included): empty softmax rows
        ei = ei[:, ei[1] % 3 != 0]
    N, E, B = gb.num_nodes, ei.shape[1], gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=77)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, max(di, 1)), 3)[:, :, :di]
    ref, _, alphas = R.gat_seq(t(x), t(ei), t(ea), t(ins), t(gb.batch), tparams(p), heads=H, return_all=True)
    m = _load_module(gat_seq(C, C, de, di, K, dropout=0.1, gat_heads=H), p, dev)
    args = [t(a, device=dev) for a in (x, ei, ea, ins, gb.batch)]
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    old_f = _lib.set_option(_lib.OPT_HOP_FUSION, 1)
    res = {}
    try:
        for coeffs in (1, 0):
            old_c = _lib.set_option(_lib.OPT_HOP_COEFFS, coeffs)
            try:
                m(*args)                                              # (weight cache for this layout)
                _lib.prof_enable(True); _lib.prof_collect()
                out = m(*args)
                pr = _lib.prof_collect()
                out_a, alpha = m(*args, return_attention_weights=True)[:2]
                pr_a = _lib.prof_collect(); _lib.prof_enable(False)
            finally:
                _lib.set_option(_lib.OPT_HOP_COEFFS, old_c)
                _lib.prof_enable(False)
            res[coeffs] = (out, out_a, alpha, pr, pr_a)
            assert maxabs(out, ref) < TOL and maxabs(out_a, ref) < TOL, (shape, coeffs)
            if E:
                assert maxabs(alpha, torch.stack(alphas)) < 2e-5, (shape, coeffs)
            assert pr["mp"][1] == 0 and pr["proj"][1] == K and pr["pack"][1] == 1, (shape, coeffs, pr)       # chained: one pack pass, K hop launches
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
        _lib.set_option(_lib.OPT_HOP_FUSION, old_f)
    in_kernel = shape != "dense_row_groups"
    assert res[1][3]["alpha"][1] == (0 if in_kernel else K) and res[1][4]["alpha"][1] == (0 if in_kernel else K), (shape, res[1][3])
    assert res[0][3]["alpha"][1] == K
    assert maxabs(res[1][0], res[0][0]) < 2e-5 and torch.equal(res[1][0], res[1][1])      # the two forms agree; asking for alpha changes nothing


@pytest.mark.parametrize("stratum", ["tiny_groups", "sparse_e_over_n_1", "hubs", "many_small", "big_graphs", "shard256"])
def test_aggregate_first_column_parts_parity(dev, stratum):
    """GVQA_OPT_HOP_FUSION = 6 (round 5): the aggregate-first hop with a row group's 512 output columns split over four workgroups
    (k_hopagg4<4, 2, 1, 2, ..., CP = 4>, per-hop launches; built for strong-scaling shards, explicit only since it measured slower
    than the 8-wave kernel there).  The next hop's node logits and the per-graph maxima travel as one set per part and are combined by the
    reader.  Widths 512 / 448 / 400 (partial last part), K = 1..5, every batch regime, plain / attention-weight / per-hop outputs,
    against the oracle at 1e-4 (alpha 5e-5), and on a 256-graph shard of the config-3 batch against the oracle and the 8-wave kernel."""
    from tests.fuzz import stratified_case, run, STRATA
    from graphvqa_amd import _lib
    if stratum == "shard256":
        from oracle import ref_torch as R
        from graphvqa_amd.gat_skip import gat_seq
        from graphvqa_amd.graph import SceneGraphBatch
        gb = synth.config3_batch(256)
        N, E, B, d = gb.num_nodes, gb.num_edges, gb.num_graphs, 512
        p = synth.gat_seq_params(d, d, d, d, 5, 4, seed=777)
        x, ea, ins = synth.normal((N, d), 1), synth.normal((E, d), 2), synth.normal((5, B, d), 3)
        m = _load_module(gat_seq(d, d, d, d, 5, dropout=0.1, gat_heads=4), p, dev)
        args = [t(a, device=dev) for a in (x, gb.edge_index, ea, ins, gb.batch)]
        g = SceneGraphBatch(args[1], args[4], N, B)
        assert m.hop_kernel(g) != "aggregate_first_parts"          # (explicit only: measured slower than the 8-wave kernel on this very shard)
        out8 = m(*args, graph=g)
        old = _lib.set_option(_lib.OPT_HOP_FUSION, 6)
        try:
            assert m.hop_kernel(g) == "aggregate_first_parts"
            _lib.prof_enable(True); _lib.prof_collect()
            out = m(*args, graph=g)
            pr = _lib.prof_collect(); _lib.prof_enable(False)
        finally:
            _lib.set_option(_lib.OPT_HOP_FUSION, old)
            _lib.prof_enable(False)
        assert pr["proj"][1] == 5 and pr["alpha"][1] == 0 and pr["mp"][1] == 0, pr
        threads = torch.get_num_threads(); torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        try:
            ref = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=4)
        finally:
            torch.set_num_threads(threads)
        assert maxabs(out, ref) < TOL and maxabs(out, out8) < 2e-5     # (the 8-wave kernel on the same shard: same rows within the kernels' tolerance)
        return
    rng = np.random.default_rng(6000 + STRATA.index(stratum))
    bad, ran = [], 0
    for K in range(1, 6):
        for rep in range(2):
            c = stratified_case(rng, 6, stratum, K)
            c["H"] = 4
            c["C"] = [512, 448, 400, 512, 480][K - 1] if rep == 0 else int(rng.choice([512, 416, 388 + 4 * int(rng.integers(0, 8))]))
            c["di"] = int(rng.choice([0, 8, 512]))
            _lib.prof_enable(True); _lib.prof_collect()
            ok, errs, sz = run(c, dev)
            pr = _lib.prof_collect(); _lib.prof_enable(False)
            assert pr["mp"][1] == 0 and pr["proj"][1] == 2 * K and pr["alpha"][1] == 0 and pr["node_logit"][1] == 0, (c, pr)      # 2 forwards x K hop launches, nothing else
            ran += 1
            if not ok:
                bad.append((c, errs, sz))
    assert ran == 10 and not bad, (ran, len(bad), bad[:3])


def test_parity_sweep_bound_census(dev):
    """tests/fuzz.run asserts north_star's 1e-4 MAX-ABS whenever the oracle's outputs stay within 32 in magnitude and a bound scaled
    by peak / 32 only above that (VERDICT r04 #3a).  This test reports how many of the sweeps' cases (run earlier in this file, same
    process) took the scaled form, and holds the scaled share under a quarter: the net is an absolute-bound net."""
    import json
    from tests.fuzz import STATS, case, run
    if STATS["cases"] == 0:               # run alone (-k): a small sample so that the census is not vacuous
        rng = np.random.default_rng(7)
        for _ in range(12):
            ok, errs, _ = run(case(rng), dev)
            assert ok, errs
    print("parity sweep bound census:", json.dumps(STATS))
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/fuzz_bound_census.json", "w") as f:
            json.dump(STATS, f)
    assert STATS["scaled_bound_cases"] <= 0.25 * STATS["cases"], STATS


@pytest.mark.parametrize("fusion", [1, 2, 4])
def test_weight_rows_spanning_2_to_the_24_inside_one_column_block(dev, fusion):
    """VERDICT r03 #6, the dynamic range of split2h's weight scales.  The 8-wave and the persistent fused kernels share ONE
    power-of-two scale per 256-row column block of the packed weights: a row 2^-k below its block's largest keeps 22 - max(0, k - 16)
    significant bits, i.e. an ABSOLUTE error <= 2^-38 of the block's largest row whatever k (the second fp16 piece goes subnormal, it
    does not vanish); the aggregate-first kernel (4) scales every output column by itself.  Here every fifth output channel's weight
    rows are scaled by 2^-24 (low pieces deep in the fp16 subnormals) and the following BatchNorm amplifies exactly those channels
    by 2^12 (weight 4096, running variance 1: a 'checkpoint' that restores the small channels' range) -- the worst case for a
    shared block scale.  All three kernels must stay within 1e-4 of the fp64 oracle on every hop's output."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_seq
    H, C, K, de, di = 4, 64, 3, 8, 12
    gb = synth.make_graph_batch(40, seed=91, nodes_lo=8, nodes_hi=30, rel_per_node=1.5)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=92)
    tiny = np.arange(0, C, 5)
    for i in range(K):
        w = p[f"convs.{i}.lin_l.weight"].copy()          # [H C, Dn + Di]
        for h in range(H):
            w[h * C + tiny] *= np.float32(2.0 ** -24)
        p[f"convs.{i}.lin_l.weight"] = w
        p[f"convs.{i}.lin_r.weight"] = w
        if i < K - 1:
            g = p[f"bns.{i}.weight"].copy()
            g[tiny] = 4096.0
            p[f"bns.{i}.weight"] = g
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    m = _load_module(gat_seq(C, C, de, di, K, dropout=0.0, gat_heads=H), p, dev)
    m.hop_fusion = fusion
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        out, _, hops = m(t(x, device=dev), t(gb.edge_index, device=dev), t(ea, device=dev), t(ins, device=dev), t(gb.batch, device=dev),
                         return_hops=True)
        out2 = m(t(x, device=dev), t(gb.edge_index, device=dev), t(ea, device=dev), t(ins, device=dev), t(gb.batch, device=dev))
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    ref, hs, _ = R.gat_seq(t(x, torch.float64), t(gb.edge_index), t(ea, torch.float64), t(ins, torch.float64), t(gb.batch),
                           tparams(p, torch.float64), heads=H, return_all=True)
    scale = max(1.0, float(ref.abs().max()))
    assert maxabs(hops, torch.stack(hs)) < TOL * scale and maxabs(out, ref) < TOL * scale and maxabs(out2, ref) < TOL * scale


@pytest.mark.parametrize("fusion", [0, 1, 2])
@pytest.mark.parametrize("H,C,di", [(4, 64, 48), (2, 300, 512), (4, 36, 20)])
def test_instruction_terms_as_one_batched_two_piece_product(dev, fusion, H, C, di):
    """The per-graph instruction terms of all hops ([K] x ([B, Di] x [Di, C + H]), gat_skip.py:133,263-264 folded per graph):
    with B a whole number of 32-row tiles they are ONE batched split product (weights packed in the cache, instruction rows
    packed per call) -- forced here with the size threshold at 0, under every hop kernel, and held to the oracle; the result
    must also agree with the f32-input product the other batch sizes take (same batch, threshold back up)."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    K, de = 3, 16
    gb = synth.make_graph_batch(64, seed=3100 + C, nodes_lo=2, nodes_hi=24, rel_per_node=1.3)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=500 + H)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), 3.0 * synth.normal((K, B, di), 3)
    ref, _, alphas = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H, return_all=True)
    old_f = _lib.set_option(_lib.OPT_HOP_FUSION, fusion)
    try:
        plain, alpha_p, _ = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch, return_attention_weights=True)
        old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
        try:
            out, alpha, _ = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch, return_attention_weights=True)
        finally:
            _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    finally:
        _lib.set_option(_lib.OPT_HOP_FUSION, old_f)
    assert maxabs(out, ref) < TOL and maxabs(alpha, torch.stack(alphas)) < 2e-5
    assert maxabs(plain, ref) < TOL and maxabs(out, plain) < 5e-5 and maxabs(alpha, alpha_p) < 2e-5


@pytest.mark.parametrize("H,C,de,di,lo,hi", [(4, 64, 24, 16, 1, 40), (4, 300, 20, 12, 20, 40), (1, 32, 8, 8, 1, 128), (2, 136, 16, 0, 60, 128),
                                             (8, 48, 12, 20, 5, 70), (4, 640, 16, 8, 20, 60), (2, 1040, 8, 0, 30, 90)])
def test_persistent_hop_kernel_chained_on_ragged_batches(dev, H, C, de, di, lo, hi):
    """GVQA_OPT_HOP_FUSION = 2 (csrc/hop2.hip, two workgroups per CU) forced onto small ragged batches, every head count, channel
    counts that do not fill the last column block.  The plain eval forward CHAINS the hops: only hop 0 has a pack pass, every
    other hop reads the packed rows, scales and per-graph maxima its predecessor's launch left, and the coefficient kernel
    computes its logits from those packed rows; with per-hop outputs requested every hop reads and writes fp32 rows instead.
    Both against the oracle (/root/reference gat_skip.py:249-279 restated)."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    K = 4
    gb = synth.make_graph_batch(23, seed=2000 + H * 7 + C, nodes_lo=lo, nodes_hi=hi, rel_per_node=1.6)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=400 + H)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    ref, hs, alphas = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H, return_all=True)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    old_f = _lib.set_option(_lib.OPT_HOP_FUSION, 2)
    try:
        _lib.prof_enable(True); _lib.prof_collect()
        out, alpha, _ = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch, return_attention_weights=True)
        prof = _lib.prof_collect()
        # chained: one pack pass (hop 0; the weight cache's own pack launches are timed under the same stage on the first call)
        out2 = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch)
        _lib.prof_collect()
        out_h, alpha_h, hops = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch,
                                            return_attention_weights=True, return_hops=True)
        prof_h = _lib.prof_collect(); _lib.prof_enable(False)
        assert prof["mp"][1] == 0 and prof_h["mp"][1] == 0 and prof["alpha"][1] == K
        assert prof_h["pack"][1] - prof["pack"][1] == K - 1       # the unchained forward packs before every hop
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
        _lib.set_option(_lib.OPT_HOP_FUSION, old_f)
        _lib.prof_enable(False)
    assert maxabs(out, ref) < TOL and maxabs(alpha, torch.stack(alphas)) < 2e-5
    assert torch.equal(out, out2)                                 # deterministic
    assert maxabs(out_h, ref) < TOL and maxabs(hops, torch.stack(hs)) < TOL and maxabs(alpha_h, torch.stack(alphas)) < 2e-5


def _checkpoint_like_params(d, di, K, H, seed):
    """gat_seq parameters with the spreads a trained checkpoint shows and glorot init does not: per-head weight scales 2^-12 .. 2^12
    (all heads of a channel sit in ONE 256-row column block of the packed weights, which shares one scale), BatchNorm
    gamma / sigma spread over 2^-6 .. 2^6 per channel, shifted running means."""
    p = synth.gat_seq_params(d, d, d, di, K, H, seed=seed)
    rng = np.random.RandomState(seed)
    for i in range(K):
        W = p[f"convs.{i}.lin_l.weight"].copy().reshape(H, d, -1)
        for h in range(H):
            W[h] *= np.float32(2.0 ** rng.randint(-12, 13))
        W[rng.randint(H)] *= np.float32(1.0)
        p[f"convs.{i}.lin_l.weight"] = W.reshape(H * d, -1)
        p[f"convs.{i}.lin_r.weight"] = p[f"convs.{i}.lin_l.weight"]
        p[f"convs.{i}.att_l"] = (p[f"convs.{i}.att_l"] * np.float32(2.0 ** -6)).astype(np.float32)     # keep the logits in softmax range
        p[f"convs.{i}.att_r"] = (p[f"convs.{i}.att_r"] * np.float32(2.0 ** -6)).astype(np.float32)
    for j in range(K - 1):
        p[f"bns.{j}.weight"] = (p[f"bns.{j}.weight"] * np.exp2(rng.uniform(-6, 6, d))).astype(np.float32)
        p[f"bns.{j}.running_var"] = (p[f"bns.{j}.running_var"] * np.exp2(rng.uniform(-6, 6, d))).astype(np.float32)
        p[f"bns.{j}.running_mean"] = (p[f"bns.{j}.running_mean"] + rng.uniform(-3, 3, d)).astype(np.float32)
    return p


def test_checkpoint_like_weights_against_fp64_oracle(dev, projection_mode):
    """Operand-scale stress of the two-piece arithmetic at the fused hop's dims (d = 512, H = 4, K = 5): per-head weight scales
    2^-12 .. 2^12 inside one column block, BatchNorm gamma / sigma 2^-6 .. 2^6, node features up to |x| = 50 with a 2^10 spread
    between graphs -- every hop mode against the fp64 oracle, relative to the output's magnitude (the outputs reach 1e3 .. 1e6
    here, so the bar is the north star's 1e-4 at unit scale: 1e-4 x max|out|, and the fp32 oracle's own distance from fp64 is
    asserted to be of the same order)."""
    from oracle import ref_torch as R
    nb, d, di, K, H = 24, 512, 64, 5, 4
    gb = synth.config3_batch(nb)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = _checkpoint_like_params(d, di, K, H, seed=4242)
    rng = np.random.RandomState(7)
    x = synth.normal((N, d), 1)
    x = (x / np.abs(x).max() * 50.0 * np.exp2(-rng.randint(0, 11, B))[gb.batch][:, None]).astype(np.float32)
    ea, ins = synth.normal((E, d), 2), synth.normal((K, B, di), 3)
    ref64 = R.gat_seq(t(x, torch.float64), t(gb.edge_index), t(ea, torch.float64), t(ins, torch.float64), t(gb.batch),
                      tparams(p, torch.float64), heads=H)
    ref32 = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H)
    out = _run_gat_seq(dev, (d, d, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch)
    scale = float(ref64.abs().max())
    assert np.isfinite(scale) and scale > 0
    err, err32 = maxabs(out, ref64) / scale, maxabs(ref32, ref64) / scale
    assert err < 1e-4, (projection_mode, err, err32, scale)
    assert err < max(50 * err32, 2e-6), (projection_mode, err, err32, scale)      # the same order as fp32 arithmetic itself


def test_two_models_with_different_settings_share_a_process(dev):
    """Options travel in the dims struct of a call (gat_seq.projection / .hop_fusion), not only in process-wide state: three
    modules with different projection arithmetics and hop kernels, called alternately, each give the result of the same module
    alone under the matching process-wide option, and all agree with the oracle."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_seq
    C, de, di, K, H = 64, 24, 16, 3, 4
    gb = synth.make_graph_batch(23, seed=4321, nodes_lo=3, nodes_hi=40, rel_per_node=1.6)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=11)
    args = [t(a, device=dev) for a in (synth.normal((N, C), 1), gb.edge_index, synth.normal((E, de), 2), synth.normal((K, B, di), 3), gb.batch)]
    ref = R.gat_seq(*[a.cpu() for a in args], tparams(p), heads=H)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        settings = [("split2h", 2), ("split3", 1), ("f32", 0)]
        mods = []
        for proj, fusion in settings:
            m = _load_module(gat_seq(C, C, de, di, K, dropout=0.1, gat_heads=H), p, dev)
            m.projection, m.hop_fusion = proj, fusion
            mods.append(m)
        outs = [None] * 3
        for rnd in range(2):                              # interleaved calls
            for i, m in enumerate(mods):
                o = m(*args)
                assert outs[i] is None or torch.equal(outs[i], o)
                outs[i] = o
        codes = {"split2h": _lib.PROJECTION_SPLIT2H, "split3": _lib.PROJECTION_SPLIT3, "f32": _lib.PROJECTION_F32}
        for (proj, fusion), o in zip(settings, outs):
            op, of = _lib.set_option(_lib.OPT_PROJECTION, codes[proj]), _lib.set_option(_lib.OPT_HOP_FUSION, fusion)
            try:
                alone = _load_module(gat_seq(C, C, de, di, K, dropout=0.1, gat_heads=H), p, dev)(*args)
            finally:
                _lib.set_option(_lib.OPT_PROJECTION, op); _lib.set_option(_lib.OPT_HOP_FUSION, of)
            assert torch.equal(o, alone), (proj, fusion)
            assert maxabs(o, ref) < TOL
        assert not torch.equal(outs[0], outs[2])          # (different arithmetics do differ in the last bits)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)


def test_weight_cache_follows_replaced_parameters_and_explicit_invalidation(dev):
    """The eval forward caches folded / packed weights keyed by storage pointer and version counter.  A Parameter OBJECT replaced
    after the first call (assignment, load_state_dict(assign=True)) and a BatchNorm buffer replaced likewise must be noticed;
    an edit through `.data` does not bump the counter and needs `invalidate_weight_cache()`."""
    from graphvqa_amd.gat_skip import gat_seq
    C, de, di, K, H = 64, 24, 16, 3, 4
    gb = synth.make_graph_batch(9, seed=77, nodes_lo=3, nodes_hi=30, rel_per_node=1.5)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=21)
    p2 = synth.gat_seq_params(C, C, de, di, K, H, seed=22)
    args = [t(a, device=dev) for a in (synth.normal((N, C), 1), gb.edge_index, synth.normal((E, de), 2), synth.normal((K, B, di), 3), gb.batch)]
    m = _load_module(gat_seq(C, C, de, di, K, dropout=0.1, gat_heads=H), p, dev)
    out0 = m(*args)

    def fresh(params):
        return _load_module(gat_seq(C, C, de, di, K, dropout=0.1, gat_heads=H), params, dev)(*args)

    # (1) a replaced Parameter object (the shared lin_l / lin_r weight of hop 1) and a replaced buffer
    q = dict(p)
    q["convs.1.lin_l.weight"] = q["convs.1.lin_r.weight"] = p2["convs.1.lin_l.weight"]
    q["bns.0.running_var"] = p2["bns.0.running_var"]
    m.convs[1].lin_l.weight = torch.nn.Parameter(t(q["convs.1.lin_l.weight"], device=dev))
    m.bns[0].running_var = t(q["bns.0.running_var"], device=dev)
    out1 = m(*args)
    assert not torch.equal(out0, out1) and torch.equal(out1, fresh(q))
    # (2) load_state_dict(assign=True) swaps every tensor object
    m.load_state_dict({k: t(v, device=dev) for k, v in p2.items()}, assign=True)
    assert torch.equal(m(*args), fresh(p2))
    # (3) an edit through .data is invisible to the version counter: explicit invalidation
    m.convs[0].bias.data.add_(0.5)
    stale = m(*args)
    m.invalidate_weight_cache()
    q2 = dict(p2); q2["convs.0.bias"] = p2["convs.0.bias"] + np.float32(0.5)
    assert torch.equal(m(*args), fresh(q2))
    del stale


def test_deferred_validation_of_a_loader_side_layout(dev):
    """gvqa_graph_finalize_host reads nothing back; gvqa_graph_check_valid is its deferred check (SceneGraphBatch.check_valid):
    passes on a batch that matches its layout, reports cross-graph edges and statistics below the batch's."""
    from graphvqa_amd import _lib
    from graphvqa_amd.graph import SceneGraphBatch, HostLayout
    gb = synth.make_graph_batch(12, seed=5, nodes_lo=4, nodes_hi=30, rel_per_node=1.5)
    N, B = gb.num_nodes, gb.num_graphs
    ei, batch = t(gb.edge_index, device=dev), t(gb.batch, device=dev)
    hl = HostLayout.from_numpy(gb.edge_index, gb.batch, B)
    SceneGraphBatch(ei, batch, N, B, host_layout=hl).check_valid()
    bad = gb.edge_index.copy()
    bad[0, 3] = N - 1                                   # an edge from the last graph into the first
    with pytest.raises(ValueError):
        HostLayout.from_numpy(bad, gb.batch, B)         # the host-side constructor refuses it ...
    g = SceneGraphBatch(t(bad, device=dev), batch, N, B, host_layout=hl)        # ... a loader that vouches wrongly is caught by the deferred check
    with pytest.raises(_lib.GvqaError):
        g.check_valid()
    low = HostLayout(hl.graph_ptr, hl.edge_ptr, 1)      # a stale in-degree bound
    with pytest.raises(_lib.GvqaError):
        SceneGraphBatch(ei, batch, N, B, host_layout=low).check_valid()


def test_fused_hop_falls_back_when_a_graph_exceeds_a_row_group(dev):
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.graph import SceneGraphBatch
    H, C, de, di, K = 4, 32, 8, 8, 2
    gb = synth.make_graph_batch(3, seed=77, nodes_lo=100, nodes_hi=160, rel_per_node=1.0)
    assert gb.batch.shape[0] > 0 and np.bincount(gb.batch).max() > 128
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    g = SceneGraphBatch(t(gb.edge_index, device=dev), t(gb.batch, device=dev), N, B)
    assert g.c.num_row_groups == 0
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=5)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        out = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    assert maxabs(out, R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H)) < TOL


def test_config3_full_batch_vs_oracle(dev):
    """VERDICT r05 missing #4: the WHOLE config-3 batch (2048 graphs x 32 nodes x 128 edges, d = 512, H = 4, K = 5) through the form bench.py
    times -- default options, the loader-side layout, the one-launch aggregate-first kernel -- against the oracle's full forward
    (gat_skip.py:249-279 restated, oracle/ref_torch.gat_seq; ~15 s on 32 host threads), every one of the 65 536 x 512 outputs within 1e-4
    ABSOLUTE (the windows test below covers 9 % of the batch, the full check used to live only in bench.py's cpu_baseline leg)."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.graph import SceneGraphBatch, HostLayout
    d, H, K = 512, 4, 5
    gb = synth.config3_batch()
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(d, d, d, d, K, H, seed=777)
    x, ea, ins = synth.normal((N, d), 1), synth.normal((E, d), 2), synth.normal((K, B, d), 3)
    assert _lib.load().gvqa_get_option(_lib.OPT_HOP_FUSION) == 3
    m = _load_module(gat_seq(d, d, d, d, K, dropout=0.1, gat_heads=H), p, dev)
    args = [t(a, device=dev) for a in (x, gb.edge_index, ea, ins, gb.batch)]
    g = SceneGraphBatch(args[1], args[4], N, B, host_layout=HostLayout.from_numpy(gb.edge_index, gb.batch, B))
    assert m.hop_kernel(g) == "aggregate_first_seq"
    out = m(*args, graph=g)
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    try:
        ref = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H)
    finally:
        torch.set_num_threads(threads)
    err = (out.cpu() - ref).abs()
    assert float(ref.abs().max()) > 1.0 and float(err.max()) < TOL, (float(err.max()), float(ref.abs().max()))


@pytest.fixture(params=[1, 2, 3])
def hop_kernel(request):
    """GVQA_OPT_HOP_FUSION: 1 = the 8-wave fused hop (csrc/split3.hip), 2 = the persistent two-workgroups-per-CU kernel with chained
    hops (csrc/hop2.hip), 3 = the default rule (at config 3: the aggregate-first kernel of csrc/hopagg.hip -- one launch for the K
    hops on plain-output forwards, one launch per hop when the attention weights are returned; sub-batches fall back to 1)."""
    from graphvqa_amd import _lib
    old = _lib.set_option(_lib.OPT_HOP_FUSION, request.param)
    yield request.param
    _lib.set_option(_lib.OPT_HOP_FUSION, old)


def test_config3_full_size_properties(dev, hop_kernel):
    """BASELINE config 3 (64k nodes / 256k edges, d=512): size-independent properties --
    (1) attention rows sum to one, (2) the result is invariant to a permutation of the COO edge list
    (features permuted alike), (3) graphs are independent: a sub-batch gives the same rows, (4) oracle windows."""
    gb = synth.config3_batch()
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    d = 512
    p = synth.gat_seq_params(d, d, d, d, 5, 4, seed=777)
    x, ea, ins = synth.normal((N, d), 1), synth.normal((E, d), 2), synth.normal((5, B, d), 3)
    out, alpha, _ = _run_gat_seq(dev, (d, d, d, 5, 4), p, x, gb.edge_index, ea, ins, gb.batch,
                                 return_attention_weights=True)
    assert torch.isfinite(out).all()
    dst = t(gb.edge_index[1], device=dev)
    s = torch.zeros(N, 4, device=dev).index_add_(0, dst, alpha[2])
    assert maxabs(s, torch.ones_like(s)) < 1e-5
    perm = np.argsort(synth.uniform01(E, 99))
    out_p = _run_gat_seq(dev, (d, d, d, 5, 4), p, x, gb.edge_index[:, perm], ea[perm], ins, gb.batch)
    assert maxabs(out, out_p) < 2e-5
    nb = 64                                    # first 64 graphs on their own
    nn, ne = nb * 32, int((gb.edge_index[0] < nb * 32).sum())
    out_s = _run_gat_seq(dev, (d, d, d, 5, 4), p, x[:nn], gb.edge_index[:, :ne], ea[:ne], ins[:, :nb], gb.batch[:nn])
    assert maxabs(out[:nn], out_s) < 2e-5
    # (4) rows of the FULL-BATCH output against the oracle on a middle and on the last 64-graph window (graphs are independent, so
    # the oracle runs on the window alone): the row blocks the workgroup -> tile maps of the hop kernels move around
    from oracle import ref_torch as R
    refs = []
    for g0 in (B // 2 - 32, B - 64):
        n0, n1 = g0 * 32, (g0 + 64) * 32
        sel = (gb.edge_index[0] >= n0) & (gb.edge_index[0] < n1)
        ei_w = gb.edge_index[:, sel] - n0
        ref_w = R.gat_seq(t(x[n0:n1]), t(ei_w), t(ea[sel]), t(ins[:, g0:g0 + 64]), t(gb.batch[n0:n1] - g0), tparams(p), heads=4)
        assert maxabs(out[n0:n1], ref_w) < TOL, g0
        refs.append((n0, n1, ref_w))
    # (5) the PLAIN-output forward -- under the default rule the K hops as ONE launch, the form bench.py times (VERDICT r04 #3b) --
    # DIRECTLY against the same oracle windows, first window included; the launch count says which form ran
    from graphvqa_amd import _lib
    _lib.prof_enable(True); _lib.prof_collect()
    out_plain = _run_gat_seq(dev, (d, d, d, 5, 4), p, x, gb.edge_index, ea, ins, gb.batch)
    torch.cuda.synchronize()
    pr = _lib.prof_collect(); _lib.prof_enable(False)
    if hop_kernel == 3:
        assert pr["proj"][1] == 1 and pr["mp"][1] == 0 and pr["alpha"][1] == 0, pr          # one hop launch for K = 5, no coefficient kernel
    n0, n1 = 0, 64 * 32
    sel = gb.edge_index[0] < n1
    refs.append((n0, n1, R.gat_seq(t(x[n0:n1]), t(gb.edge_index[:, sel]), t(ea[sel]), t(ins[:, :64]), t(gb.batch[n0:n1]), tparams(p), heads=4)))
    for n0, n1, ref_w in refs:
        assert maxabs(out_plain[n0:n1], ref_w) < TOL, (n0, n1)


def test_gat_seq_train_mode_batchnorm_golden(dev):
    """model.train(), dropout p=0: batch-statistics BN forward and running-stat update (gat_skip.py:273-276)."""
    from graphvqa_amd.gat_skip import gat_seq
    meta, g0 = load_golden("gat_seq_small")
    _, g = load_golden("gat_seq_small_trainbn")
    dn, de, di, K, H = meta["dn"], meta["de"], meta["di"], meta["K"], meta["heads"]
    p = synth.gat_seq_params(dn, dn, de, di, K, H, seed=meta["param_seed"])
    m = gat_seq(dn, dn, de, di, K, dropout=0.0, gat_heads=H)
    _load_module(m, p, dev)
    m.train()
    out = m(*[t(g0[k], device=dev) for k in ("x", "edge_index", "edge_attr", "instr", "batch")])
    assert maxabs(out, g["out"]) < TOL
    rm = torch.stack([bn.running_mean for bn in m.bns])
    rv = torch.stack([bn.running_var for bn in m.bns])
    assert maxabs(rm, g["running_mean_after"]) < 1e-5 and maxabs(rv, g["running_var_after"]) < 1e-5
    assert int(m.bns[0].num_batches_tracked) == 1


def test_errors_are_loud(dev):
    from graphvqa_amd.gat_skip import gat_seq
    m = gat_seq(8, 8, 8, 8, 2, dropout=0.1, gat_heads=4).to(dev)
    x = torch.zeros(3, 8)
    m.eval()
    with pytest.raises(RuntimeError):   # CPU tensor: no fallback
        m(x, torch.zeros(2, 0, dtype=torch.int64, device=dev), torch.zeros(0, 8, device=dev),
          torch.zeros(2, 1, 8, device=dev), torch.zeros(3, dtype=torch.int64, device=dev))
    out = m(x.to(dev), torch.zeros(2, 0, dtype=torch.int64, device=dev), torch.zeros(0, 8, device=dev),
            torch.zeros(2, 1, 8, device=dev), torch.zeros(3, dtype=torch.int64, device=dev))     # E = 0 works
    assert out.shape == (3, 8) and torch.isfinite(out).all()


def test_return_flags_are_honoured_on_every_path(dev):
    """return_attention_weights / return_hops give the same (out, alpha, hops) triple on the fused eval path, in train
    mode without gradients (batch-statistics BatchNorm) and on cross-graph batches (unfolded fallback)."""
    from oracle import ref_torch as R
    from graphvqa_amd.gat_skip import gat_seq
    H, C, de, di, K = 4, 16, 8, 12, 3
    gb = synth.make_graph_batch(5, seed=19, nodes_lo=3, nodes_hi=7, rel_per_node=1.2)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=47)
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    args = lambda ei: [t(a, device=dev) for a in (x, ei, ea, ins, gb.batch)]
    m = _load_module(gat_seq(C, C, de, di, K, dropout=0.0, gat_heads=H), p, dev)
    ref, hs, alphas = R.gat_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), heads=H, return_all=True)
    out, alpha, hops = m(*args(gb.edge_index), return_attention_weights=True, return_hops=True)
    assert maxabs(out, ref) < TOL and maxabs(alpha, torch.stack(alphas)) < 2e-5 and maxabs(hops, torch.stack(hs)) < TOL
    m.train()                                               # dropout 0, no gradients: fused batch-statistics path
    res = m(*args(gb.edge_index), return_attention_weights=True, return_hops=True)
    assert isinstance(res, tuple) and len(res) == 3 and res[1].shape == (K, E, H) and res[2].shape == (K, N, C)
    plain = m(*args(gb.edge_index))
    assert maxabs(res[0], plain) < 2e-5
    m = _load_module(gat_seq(C, C, de, di, K, dropout=0.0, gat_heads=H), p, dev)     # fresh running statistics
    ei_x = gb.edge_index.copy()
    ei_x[0, 0] = N - 1                                      # an edge from the last graph into the first: cross-graph batch
    res = m(*args(ei_x), return_attention_weights=True, return_hops=True)
    refx, hsx, alx = R.gat_seq(t(x), t(ei_x), t(ea), t(ins), t(gb.batch), tparams(p), heads=H, return_all=True)
    assert maxabs(res[0], refx) < TOL and maxabs(res[1], torch.stack(alx)) < 2e-5 and maxabs(res[2], torch.stack(hsx)) < TOL


def test_malformed_batch_ids_are_reported_not_followed(dev):
    """A sentinel / out-of-range graph id in `batch` must come back as an error from the CSR build (never an
    out-of-bounds fill of graph_ptr[] or a device hang)."""
    from graphvqa_amd import _lib
    from graphvqa_amd.graph import SceneGraphBatch
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]], dtype=torch.int64, device=dev)
    for bad in ([0, -(2 ** 62), 1, 1], [0, 0, 2 ** 40, 1], [1, 0, 0, 1]):
        with pytest.raises(_lib.GvqaError):
            SceneGraphBatch(ei, torch.tensor(bad, dtype=torch.int64, device=dev), 4, 2)
    g = SceneGraphBatch(ei, torch.tensor([0, 0, 1, 1], dtype=torch.int64, device=dev), 4, 2)     # still usable afterwards
    assert g.graph_ptr.tolist() == [0, 2, 4]


# ------------------------------------------------------------------------------------------------
# GINE / GCN variants
# ------------------------------------------------------------------------------------------------
def test_gine_eps_buffer_value_is_used(dev):
    """`convs.i.eps` is a state_dict key (PyG GINEConv, train_eps=False buffer): its VALUE scales the root term
    (1 + eps) x_i -- a loaded or edited buffer must change the result exactly like in the shim."""
    from oracle import ref_torch as R
    from graphvqa_amd.baseline_models import GINEConv
    D, Cc = 12, 8
    gb = synth.make_graph_batch(3, seed=5, nodes_lo=2, nodes_hi=6, rel_per_node=1.0)
    N, E = gb.num_nodes, gb.num_edges
    nn = torch.nn.Sequential(torch.nn.Linear(D, Cc), torch.nn.ReLU(), torch.nn.Linear(Cc, Cc))
    conv = GINEConv(nn, eps=0.0).to(dev)
    x, ea = t(synth.normal((N, D), 1), device=dev), t(synth.normal((E, D), 2), device=dev)
    ei = t(gb.edge_index, device=dev)
    out0 = conv(x, ei, ea)
    conv.eps.fill_(0.75)                                    # in place, as load_state_dict does
    out1 = conv(x, ei, ea)
    w = {k: v.detach().cpu() for k, v in nn.state_dict().items()}
    def ref(eps):
        agg = torch.zeros(N, D).index_add_(0, t(gb.edge_index[1]), torch.relu(x.cpu()[gb.edge_index[0]] + ea.cpu()))
        hcat = (1 + eps) * x.cpu() + agg
        return torch.relu(hcat @ w["0.weight"].T + w["0.bias"]) @ w["2.weight"].T + w["2.bias"]
    assert maxabs(out0, ref(0.0)) < 1e-4 and maxabs(out1, ref(0.75)) < 1e-4 and maxabs(out0, out1) > 1e-3


def test_gine_seq_module_and_convs_golden(dev):
    from graphvqa_amd.baseline_models import gine_seq
    meta, g = load_golden("gine_seq_small")
    p = synth.gine_seq_params(meta["dn"], meta["dn"], meta["di"], meta["param_seed"])
    m = gine_seq(meta["dn"], meta["dn"], meta["di"], dropout=0.1)
    _load_module(m, p, dev)
    args = [t(g[k], device=dev) for k in ("x", "edge_index", "edge_attr", "instr", "batch")]
    out = m(*args)
    assert maxabs(out, g["out"]) < 1e-5                      # module output as written (conv discarded)
    out2, convs = m(*args, return_convs=True)
    assert torch.equal(out, out2)
    assert maxabs(torch.stack(convs), g["convs"]) < TOL      # the conv layers themselves


def test_gcn_seq_module_and_convs_golden(dev):
    from graphvqa_amd.baseline_models import gcn_seq, GCNConv
    meta, g = load_golden("gcn_seq_small")
    p = synth.gcn_seq_params(meta["dn"], meta["dn"], meta["di"], meta["param_seed"])
    m = gcn_seq(meta["dn"], meta["dn"], meta["di"], dropout=0.1)
    _load_module(m, p, dev)
    args = [t(g[k], device=dev) for k in ("x", "edge_index", "instr", "batch")]
    out, convs = m(*args, return_convs=True)
    assert maxabs(out, g["out"]) < 1e-5
    assert maxabs(torch.stack(convs), g["convs"]) < TOL
    # plain GCNConv on a multigraph with duplicate self loops and a node lacking one
    meta, g = load_golden("gcn_conv_small")
    p = synth.gcn_seq_params(20, 8, 4, meta["param_seed"], num_layers=1)
    conv = GCNConv(24, 8)
    _load_module(conv, {"weight": p["convs.0.weight"], "bias": p["convs.0.bias"]}, dev)
    o = conv(t(g["x"], device=dev), t(g["edge_index"], device=dev))
    assert maxabs(o, g["out"]) < 1e-5


def test_gine_gcn_config2_shape_vs_oracle(dev):
    """BASELINE config 4: GINEConv layers on the config-2 batch at the reference's dims (812-wide
    messages, MLP 812->300->300), and GCNConv(812->300), against the fp32 oracle."""
    from oracle import ref_torch as R
    from graphvqa_amd.baseline_models import gine_seq, gcn_seq
    gb = synth.config2_batch()
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    x, ea, ins = synth.normal((N, 300), 1), synth.normal((E, 300), 2), synth.normal((5, B, 512), 3)
    dargs = [t(a, device=dev) for a in (x, gb.edge_index, ea, ins, gb.batch)]
    p = synth.gine_seq_params(300, 300, 512, 404)
    m = _load_module(gine_seq(300, 300, 512), p, dev)
    out, convs = m(*dargs, return_convs=True)
    ref_out, ref_convs = R.gine_seq(t(x), t(gb.edge_index), t(ea), t(ins), t(gb.batch), tparams(p), return_convs=True)
    assert maxabs(out, ref_out) < 1e-5
    assert maxabs(torch.stack(convs), torch.stack(ref_convs)) < TOL
    p = synth.gcn_seq_params(300, 300, 512, 505)
    m = _load_module(gcn_seq(300, 300, 512), p, dev)
    out, convs = m(dargs[0], dargs[1], dargs[3], dargs[4], return_convs=True)
    ref_out, ref_convs = R.gcn_seq(t(x), t(gb.edge_index), t(ins), t(gb.batch), tparams(p), return_convs=True)
    assert maxabs(out, ref_out) < 1e-5
    assert maxabs(torch.stack(convs), torch.stack(ref_convs)) < TOL


@pytest.mark.parametrize("dn,C,di,graphs", [(300, 300, 512, 120), (64, 320, 16, 200), (812, 300, 0, 90), (40, 8, 8, 400), (128, 132, 24, 64), (300, 300, 512, 37)])
def test_gine_conv_fused_mlp_vs_oracle(dev, dn, C, di, graphs):
    """GINEConv with nn = Lin -> ReLU -> Lin as ONE kernel (csrc/gine_mlp.hip, round 6: layer 1's A operand converted from the fp32 rows
    in registers, the hidden rows handed from layer 1's accumulators to layer 2's A fragments in registers, both weights through one
    LDS ring) against the oracle's GINEConv (pipeline_model_gine.py:628,651-665) at 1e-4: the reference's dims, the widest tile count
    (C = 320), a 812-wide layer 1 without instruction halves (K tail 812 = 50 x 16 + 12), a one-quad output, widths with partial tiles,
    row counts that are no multiple of 128; against the unfused path (GVQA_GINE_FUSED is read once: compared through the oracle); large
    magnitudes in single rows (per-row scales)."""
    from oracle import ref_torch as R
    from graphvqa_amd.baseline_models import GINEConv
    from graphvqa_amd.graph import SceneGraphBatch
    from graphvqa_amd import _lib
    from torch.nn import Linear, ReLU, Sequential
    gb = synth.make_graph_batch(graphs, seed=0x61 + C + dn, nodes_lo=5, nodes_hi=40, rel_per_node=1.5)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    rng = np.random.default_rng(dn * 7 + C)
    x, ea = synth.normal((N, dn), 1), synth.normal((E, dn), 2)
    x[rng.integers(0, N, size=5)] *= 300.0                                    # a few rows far above the rest: the row scales are per row
    x[rng.integers(0, N, size=5)] *= 1e-3
    ins = synth.normal((B, max(di, 1)), 3)[:, :di]
    w = lambda *shape: (rng.standard_normal(shape) / np.sqrt(shape[-1])).astype(np.float32)
    p = {"nn.0.weight": w(C, dn + di), "nn.0.bias": w(C), "nn.2.weight": w(C, C), "nn.2.bias": w(C)}
    p["nn.2.weight"][C // 2] *= 1e-4                                           # an output channel far below its neighbours: per-column weight scales
    conv = GINEConv(Sequential(Linear(dn + di, C), ReLU(), Linear(C, C)), eps=0.25).to(dev).eval()
    conv.load_state_dict({**{k: t(v) for k, v in p.items()}, "eps": torch.tensor([0.25])})
    g = SceneGraphBatch(t(gb.edge_index, device=dev), t(gb.batch, device=dev), N, B)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        _lib.prof_enable(True); _lib.prof_collect()
        out = conv(t(x, device=dev), t(gb.edge_index, device=dev), t(ea, device=dev), graph=g, ins=t(ins, device=dev) if di else None)
        pr = _lib.prof_collect(); _lib.prof_enable(False)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    xc = np.concatenate([x, ins[gb.batch]], 1) if di else x
    ec = np.concatenate([ea, ins[gb.batch[gb.edge_index[0]]]], 1) if di else ea
    ref = R.gine_conv(t(xc).double(), t(gb.edge_index), t(ec).double(), {k: t(v).double() for k, v in p.items()}, eps=0.25)
    scale = max(1.0, float(ref.abs().max()) / 32.0)                            # (the sweeps' bound: absolute 1e-4 up to |ref| = 32, relative to the peak above)
    assert maxabs(out, ref.float()) < TOL * scale, (maxabs(out, ref.float()), scale)


def test_gine_unaligned_width_and_cross_graph_edges(dev):
    """C % 4 != 0 takes the scalar kernels; cross-graph edges take the concatenated formulation."""
    from oracle import ref_torch as R
    from graphvqa_amd.baseline_models import gine_seq
    dn, di = 10, 6
    gb = synth.make_graph_batch(5, seed=8, nodes_lo=2, nodes_hi=7, rel_per_node=1.3)
    ei = np.concatenate([gb.edge_index, np.array([[1], [gb.num_nodes - 1]])], axis=1)
    N, E, B = gb.num_nodes, ei.shape[1], gb.num_graphs
    p = synth.gine_seq_params(dn, dn, di, 909)
    x, ea, ins = synth.normal((N, dn), 1), synth.normal((E, dn), 2), synth.normal((5, B, di), 3)
    m = _load_module(gine_seq(dn, dn, di), p, dev)
    out, convs = m(t(x, device=dev), t(ei, device=dev), t(ea, device=dev), t(ins, device=dev),
                   t(gb.batch, device=dev), return_convs=True)
    _, ref_convs = R.gine_seq(t(x), t(ei), t(ea), t(ins), t(gb.batch), tparams(p), return_convs=True)
    assert maxabs(torch.stack(convs), torch.stack(ref_convs)) < 1e-5


# ------------------------------------------------------------------------------------------------
# LCGN variant
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["lcgn_seq_small", "lcgn_seq_debug4_d512"])
def test_lcgn_seq_golden(dev, name):
    from graphvqa_amd.lcgn import lcgn_seq
    meta, g = load_golden(name)
    s = meta["input_seeds"]
    O, in_c, L = meta["out_channels"], meta["in_channels"], meta["L"]
    N, B = g["batch"].shape[0], int(g["batch"].max()) + 1
    p = synth.lcgn_seq_params(in_c, O, seed=meta["param_seed"], cmd_dim=O, question_dim=O)
    m = lcgn_seq(in_channels=in_c, out_channels=O, edge_attr_dim=in_c, num_ins=5, gat_cmd_dim=O, question_dim=O,
                 MAX_ITER_NUM=4, dropout=0.1, gat_heads=1)
    _load_module(m, p, dev)
    x, q, lstm = synth.normal((N, in_c), s["x"]), synth.normal((B, O), s["q"]), synth.normal((L, B, O), s["lstm"])
    torch.manual_seed(meta["torch_seed"])        # the module draws x_ctx like the reference (lcgn.py:306)
    out = m(t(x, device=dev), t(g["edge_index"], device=dev), t(g["batch"], device=dev), t(q, device=dev),
            t(lstm, device=dev))
    assert maxabs(out, g["out"]) < TOL
    out2 = m(t(x, device=dev), t(g["edge_index"], device=dev), t(g["batch"], device=dev), t(q, device=dev),
             t(lstm, device=dev), x_ctx_init=t(g["x_ctx_init"], device=dev))
    assert torch.equal(out, out2)


@pytest.mark.parametrize("arith", ["split2h", "f32"])
def test_lcgn_config2_shape_vs_oracle(dev, arith):
    """BASELINE config 5 shape (fp32): config-2 batch, lcgn_seq(in=300, out=512, cmd=512, H=1, 4 iterations), L=10, with the
    node GEMMs on the two-piece fp16 arithmetic (default) and on the f32-input MFMA kernels."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.lcgn import lcgn_seq
    old = _lib.set_option(_lib.OPT_PROJECTION, _lib.PROJECTION_SPLIT2H if arith == "split2h" else _lib.PROJECTION_F32)
    try:
        _lcgn_config2_case(dev, R, lcgn_seq)
    finally:
        _lib.set_option(_lib.OPT_PROJECTION, old)


def _lcgn_config2_case(dev, R, lcgn_seq):
    gb = synth.config2_batch()
    N, B, O, L = gb.num_nodes, gb.num_graphs, 512, 10
    p = synth.lcgn_seq_params(300, O, seed=808)
    m = _load_module(lcgn_seq(300, O, 300, 5), p, dev)
    x, q, lstm = synth.normal((N, 300), 1), synth.normal((B, O), 2), synth.normal((L, B, O), 3)
    x_ctx = synth.normal((N, O), 4)
    out = m(t(x, device=dev), t(gb.edge_index, device=dev), t(gb.batch, device=dev), t(q, device=dev),
            t(lstm, device=dev), x_ctx_init=t(x_ctx, device=dev))
    ref = R.lcgn_seq(t(x), t(gb.edge_index), t(gb.batch), t(q), t(lstm), tparams(p), t(x_ctx))
    assert maxabs(out, ref) < TOL


# ------------------------------------------------------------------------------------------------
# "next" row 8f-2: global attention pooling + short-answer classifier
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["pool_head_small", "pool_head_debug4"])
def test_pool_and_classifier_golden(dev, name):
    from graphvqa_amd.pipeline_head import MyConditionalGlobalAttention, ShortAnswerClassifier
    meta, g = load_golden(name)
    in_c, ch, nans = meta["in_channels"], meta["channels"], meta["num_answers"]
    N, B = g["batch"].shape[0], int(g["batch"].max()) + 1
    pool = _load_module(MyConditionalGlobalAttention(in_c, ch), synth.attention_pool_params(in_c, ch, seed=meta["pool_seed"]), dev)
    head = _load_module(ShortAnswerClassifier(ch, ch, nans), synth.classifier_params(ch, ch, nans, seed=meta["fc_seed"]), dev)
    x, u = synth.normal((N, in_c), meta["input_seeds"]["x"]), synth.normal((B, ch), meta["input_seeds"]["u"])
    pooled = pool(t(x, device=dev), t(u, device=dev), t(g["batch"], device=dev))
    assert maxabs(pooled, g["pooled"]) < 2e-5
    logits = head(pooled, t(u, device=dev))
    assert maxabs(logits, g["logits"]) < TOL


@pytest.mark.parametrize("in_c,ch", [(300, 512), (64, 128), (40, 72)])
def test_pooling_head_on_the_fused_products_vs_oracle(dev, in_c, ch):
    """The pooling head's large-batch form (pipeline_model_gat.py:149-181): node_nn / gate_nn on the two-piece kernels, the
    per-graph row scaling inside gate_nn's operand pack, gate_nn's 512 -> 1 Linear as 16 partial dot products in the first
    product's epilogue, summed in order by the pooling kernel.  Golden and small tests run below the size threshold; here it
    is 0 so that this form is the one exercised -- ragged graphs, an empty graph, one-node graphs, a graph past the pooling
    kernel's 256-node LDS path -- against the fp64 oracle, and against the small-batch form on the same inputs."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.pipeline_head import MyConditionalGlobalAttention
    counts = np.array([33, 1, 0, 300, 7, 128, 1, 64] + [int(v) for v in synth.randint(40, 5, 1, 60)], dtype=np.int64)
    batch = np.repeat(np.arange(len(counts)), counts)
    N, B = int(counts.sum()), len(counts)
    pp = synth.attention_pool_params(in_c, ch, seed=17)
    rng = np.random.default_rng(7)
    for k in list(pp):
        if k.endswith("bias"):
            pp[k] = (pp[k] + 0.2 * rng.standard_normal(pp[k].shape)).astype(np.float32)
    pool = _load_module(MyConditionalGlobalAttention(in_c, ch), pp, dev)
    x, u = synth.normal((N, in_c), 1), synth.normal((B, ch), 2)
    ref = R.global_attention_pool(t(x).double(), t(u).double(), t(batch), {k: v.double() for k, v in tparams(pp).items()}, B)
    small = pool(t(x, device=dev), t(u, device=dev), t(batch, device=dev))
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        fused = pool(t(x, device=dev), t(u, device=dev), t(batch, device=dev))
        again = pool(t(x, device=dev), t(u, device=dev), t(batch, device=dev))
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    scale = max(1.0, float(ref.abs().max()))
    assert maxabs(small, ref) < TOL * scale
    assert maxabs(fused, ref) < TOL * scale
    assert torch.equal(fused, again)                      # partial sums are combined in a fixed order
    assert float(fused[2].abs().max()) == 0.0             # the empty graph pools to zeros (scatter_add of nothing)


# ------------------------------------------------------------------------------------------------
# "next" row 8f-1: scene-graph encoder; and config 1: encoder -> gat_seq -> pooling -> logits
# ------------------------------------------------------------------------------------------------
def test_scene_graph_encoder_golden(dev):
    import types
    from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
    meta, g = load_golden("sg_encoder_debug4")
    enc = GroundTruth_SceneGraph_Encoder(meta["vocab"], meta["pad_idx"], meta["dim"])
    _load_module(enc, synth.encoder_params(meta["vocab"], meta["dim"], seed=meta["param_seed"], pad_idx=meta["pad_idx"]), dev)
    data = types.SimpleNamespace(x=t(g["x_tokens"], device=dev), edge_attr=t(g["edge_tokens"], device=dev),
                                 edge_index=t(g["edge_index"], device=dev), batch=t(g["batch"], device=dev),
                                 added_sym_edge=t(g["added_sym_edge"], device=dev))
    xe, ee, _ = enc(data)
    assert maxabs(ee, g["edge_attr_encoded"]) < 5e-5
    assert maxabs(xe, g["x_encoded"]) < TOL


def test_encoder_rejects_token_ids_outside_the_table(dev):
    import types
    from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
    meta, g = load_golden("sg_encoder_debug4")
    enc = _load_module(GroundTruth_SceneGraph_Encoder(meta["vocab"], meta["pad_idx"], meta["dim"]),
                       synth.encoder_params(meta["vocab"], meta["dim"], seed=meta["param_seed"], pad_idx=meta["pad_idx"]), dev)
    def data(x_tok=g["x_tokens"], e_tok=g["edge_tokens"], added=g["added_sym_edge"]):
        return types.SimpleNamespace(x=t(x_tok, device=dev), edge_attr=t(e_tok, device=dev), edge_index=t(g["edge_index"], device=dev),
                                     batch=t(g["batch"], device=dev), added_sym_edge=t(added, device=dev))
    bx = g["x_tokens"].copy(); bx[3, 1] = meta["vocab"]
    be = g["edge_tokens"].copy(); be[7, 0] = -1
    ba = g["added_sym_edge"].copy(); ba[0] = g["edge_index"].shape[1]
    for d in (data(x_tok=bx), data(e_tok=be), data(added=ba)):
        with pytest.raises(IndexError):
            enc(d)
    enc(data())                                            # the valid batch still runs


def test_config1_pipeline_forward_golden(dev):
    """BASELINE config 1 on the HIP path against the REFERENCE'S OWN `PipelineModel.forward` (pipeline_model_gat.py:743-821,
    captured in tests/golden/pipeline_debug2.npz): debug graphs 2354786 + 2375429, batch 2, through scene-graph encoder ->
    gat_seq (K = 5) -> global attention pooling -> 1842-way answer logits; the recorded instruction vectors and question
    feature (outputs of the out-of-scope transformers) are fed at the seams (:764, :803)."""
    import types
    from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.pipeline_head import MyConditionalGlobalAttention, ShortAnswerClassifier
    meta, g = load_golden("pipeline_debug2")
    V = meta["vocab"]
    enc = _load_module(GroundTruth_SceneGraph_Encoder(V, meta["pad_idx"], 300),
                       synth.encoder_params(V, 300, seed=meta["encoder_seed"], pad_idx=meta["pad_idx"]), dev)
    gs = _load_module(gat_seq(300, 300, 300, 512, 5, dropout=0.1, gat_heads=4),
                      synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=meta["gat_seq_seed"]), dev)
    pool = _load_module(MyConditionalGlobalAttention(300, 512), synth.attention_pool_params(300, 512, seed=meta["pool_seed"]), dev)
    clf = _load_module(ShortAnswerClassifier(512, 512, 1842), synth.classifier_params(512, 512, 1842, seed=meta["fc_seed"]), dev)
    data = types.SimpleNamespace(x=t(g["x_tokens"], device=dev), edge_attr=t(g["edge_tokens"], device=dev),
                                 edge_index=t(g["edge_index"], device=dev), batch=t(g["batch"], device=dev),
                                 added_sym_edge=t(g["added_sym_edge"], device=dev))
    xe, ee, _ = enc(data)
    assert maxabs(xe, g["x_encoded"]) < TOL and maxabs(ee, g["edge_attr_encoded"]) < 5e-5
    h = gs(xe, data.edge_index, ee, t(g["instr_vectors"], device=dev), data.batch)
    assert maxabs(h, g["x_executed"]) < TOL
    q = t(g["question_feature"], device=dev)
    pooled = pool(h, q, data.batch)
    logits = clf(pooled, q)
    assert logits.shape == (2, 1842)
    assert maxabs(pooled, g["pooled"]) < TOL and maxabs(logits, g["short_answer_logits"]) < TOL
    # the path alone from the reference's own encoder outputs (what pipeline_model_gat.py:791 hands over)
    h2 = gs(t(g["x_encoded"], device=dev), data.edge_index, t(g["edge_attr_encoded"], device=dev),
            t(g["instr_vectors"], device=dev), data.batch)
    assert maxabs(h2, g["x_executed"]) < TOL


def test_config1_debug_pipeline_chain(dev):
    """BASELINE config 1 shape: the reference's two debug scene graphs (batch = 2) through
    encoder -> gat_seq (K = 5) -> global attention pooling -> answer logits on the HIP path, against the
    same chain on the CPU oracle (transformer question encoder / program decoder are out of scope:
    instruction vectors and the question feature are seeded inputs)."""
    import json, types
    from oracle import ref_torch as R
    from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.pipeline_head import MyConditionalGlobalAttention, ShortAnswerClassifier
    meta, gg = load_golden("gat_seq_debug2_d300")
    ei, batch = gg["edge_index"], gg["batch"]
    N, E, B, V = batch.shape[0], ei.shape[1], 2, 50
    assert (N, E) == (33, 125)
    x_tok = synth.randint(N * 12, 71, 0, V, stream=9).reshape(N, 12)
    e_tok = synth.randint(E, 72, 1, V, stream=9).reshape(E, 1)
    added = np.array([3, 17, 60], dtype=np.int64)
    ins, q = synth.normal((5, B, 512), 73), synth.normal((B, 512), 74)
    pe, pg = synth.encoder_params(V, 300, seed=1), synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=2)
    pp, pc = synth.attention_pool_params(300, 512, seed=3), synth.classifier_params(512, 512, 1842, seed=4)
    enc = _load_module(GroundTruth_SceneGraph_Encoder(V, 0, 300), pe, dev)
    gs = _load_module(gat_seq(300, 300, 300, 512, 5, dropout=0.1, gat_heads=4), pg, dev)
    pool = _load_module(MyConditionalGlobalAttention(300, 512), pp, dev)
    clf = _load_module(ShortAnswerClassifier(512, 512, 1842), pc, dev)
    data = types.SimpleNamespace(x=t(x_tok, device=dev), edge_attr=t(e_tok, device=dev), edge_index=t(ei, device=dev),
                                 batch=t(batch, device=dev), added_sym_edge=t(added, device=dev))
    xe, ee, _ = enc(data)
    h = gs(xe, data.edge_index, ee, t(ins, device=dev), data.batch)
    logits = clf(pool(h, t(q, device=dev), data.batch), t(q, device=dev))
    rxe, ree = R.scene_graph_encoder(t(x_tok), t(ei), t(e_tok), t(added), t(batch), B, tparams(pe))
    rh = R.gat_seq(rxe, t(ei), ree, t(ins), t(batch), tparams(pg))
    rlogits = R.short_answer_logits(R.global_attention_pool(rh, t(q), t(batch), tparams(pp), B), t(q), tparams(pc))
    assert logits.shape == (2, 1842)
    assert maxabs(h, rh) < TOL and maxabs(logits, rlogits) < TOL


def test_split3_projection_is_fp32_accurate_in_situ(dev):
    """GVQA_PROJ (read once per process -> subprocess): the split projections (two fp16 pieces = the default, three bf16
    pieces; forced onto this small batch with GVQA_SPLIT3_MIN_MFLOP=0) must be as close to the fp64 oracle as the f32-input
    MFMA kernels on the real-dims golden case recorded from the reference's own gat_seq."""
    import os, subprocess, sys, json
    code = r'''
import json, sys, numpy as np, torch
torch.set_grad_enabled(False)
sys.path.insert(0, %r)
from graphvqa_amd import synth, _lib
from graphvqa_amd.gat_skip import gat_seq
from oracle import ref_torch as R
from tests.util import load_golden, t, tparams
meta, g = load_golden("gat_seq_debug4_d300")
s = meta["input_seeds"]; N, E, B = g["batch"].shape[0], g["edge_index"].shape[1], int(g["batch"].max()) + 1
x, ea, ins = synth.normal((N, 300), s["x"]), synth.normal((E, 300), s["edge_attr"]), synth.normal((5, B, 512), s["instr"])
p = synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=meta["param_seed"])
m = gat_seq(300, 300, 300, 512, 5, dropout=0.1, gat_heads=4); m.load_state_dict({k: t(v) for k, v in p.items()}); m = m.cuda().eval()
out = m(t(x).cuda(), t(g["edge_index"]).cuda(), t(ea).cuda(), t(ins).cuda(), t(g["batch"]).cuda()).cpu().double()
ref = R.gat_seq(t(x, torch.float64), t(g["edge_index"]), t(ea, torch.float64), t(ins, torch.float64), t(g["batch"]), tparams(p, torch.float64))
print(json.dumps({"err64": float((out - ref).abs().max()), "err_golden": float((out - torch.from_numpy(g["out"]).double()).abs().max()),
                  "backend": _lib.load().gvqa_gemm_backend().decode()}))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    res = {}
    for mode in ("split2h", "split3", "f32"):
        env = dict(os.environ, GVQA_PROJ=mode, GVQA_SPLIT3_MIN_MFLOP="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    assert "three exact bf16" in res["split3"]["backend"] and "two scaled fp16" in res["split2h"]["backend"]
    assert "k_linear_split3" not in res["f32"]["backend"]
    assert all(r["err_golden"] < TOL for r in res.values()), res
    assert res["split3"]["err64"] < max(2e-5, 3 * res["f32"]["err64"]), res
    assert res["split2h"]["err64"] < max(2e-5, 3 * res["f32"]["err64"]), res


def test_empty_batch_and_empty_graphs_in_the_middle(dev):
    """N = 0 returns an empty tensor; graph ids that own no node (batch skips them), nodes without
    in-edges and single-node graphs follow the reference (softmax over nothing = 0, no instruction term)."""
    from oracle import ref_torch as R
    from graphvqa_amd.gat_skip import gat_seq
    H, C, de, di, K = 4, 16, 8, 12, 3
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=77)
    m = _load_module(gat_seq(C, C, de, di, K, gat_heads=H), p, dev)
    z = m(torch.zeros(0, C, device=dev), torch.zeros(2, 0, dtype=torch.int64, device=dev), torch.zeros(0, de, device=dev),
          torch.zeros(K, 3, di, device=dev), torch.zeros(0, dtype=torch.int64, device=dev))
    assert z.shape == (0, C)
    # 5 graph ids, graphs 1 and 3 own no node; node 4 has no in-edge; graph 4 is a single node with a self loop
    batch = np.array([0, 0, 0, 2, 2, 4], dtype=np.int64)
    ei = np.array([[0, 1, 2, 0, 3, 5, 1], [0, 1, 2, 1, 3, 5, 2]], dtype=np.int64)
    N, E, B = 6, ei.shape[1], 5
    x, ea, ins = synth.normal((N, C), 1), synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    out = m(t(x, device=dev), t(ei, device=dev), t(ea, device=dev), t(ins, device=dev), t(batch, device=dev))
    ref = R.gat_seq(t(x), t(ei), t(ea), t(ins), t(batch), tparams(p), heads=H)
    assert maxabs(out, ref) < 2e-5


@pytest.mark.parametrize("case", range(12))
def test_randomized_gat_seq_vs_oracle(dev, case):
    """Randomised shapes: heads, widths (incl. C % 4 != 0 -> scalar kernels), ragged graph sizes (incl.
    graphs too large for an LDS tile -> general kernel), multi-edges, missing self loops, hop counts."""
    from oracle import ref_torch as R
    r = lambda lo, hi, s: int(synth.randint(1, 9000 + 17 * case + s, lo, hi + 1)[0])
    H = [1, 2, 4, 8][r(0, 3, 1)]
    C = [8, 12, 20, 30, 64, 100, 132][r(0, 6, 2)]
    de, di, K = r(4, 40, 3), r(4, 40, 4), r(1, 6, 5)
    B = r(1, 40, 6)
    big = case % 4 == 3
    gb = synth.make_graph_batch(B, seed=500 + case, nodes_lo=1, nodes_hi=(700 if big else 45), rel_per_node=r(0, 30, 7) / 10.0)
    ei = gb.edge_index
    if case % 3 == 0:                                     # drop the self loops of every third node
        keep = ~((ei[0] == ei[1]) & (ei[0] % 3 == 0))
        ei = ei[:, keep]
    N, E = gb.num_nodes, ei.shape[1]
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=600 + case)
    x, ea, ins = synth.normal((N, C), 1 + case), synth.normal((E, de), 2 + case), synth.normal((K, B, di), 3 + case)
    out = _run_gat_seq(dev, (C, de, di, K, H), p, x, ei, ea, ins, gb.batch)
    ref = R.gat_seq(t(x), t(ei), t(ea), t(ins), t(gb.batch), tparams(p), heads=H)
    assert maxabs(out, ref) < TOL, (H, C, de, di, K, B, N, E)


@pytest.mark.parametrize("scheme", ["split2h", "split3"])
@pytest.mark.parametrize("case", range(10))
def test_randomized_fused_hop_vs_oracle(dev, scheme, case):
    """The fused hop forced onto randomised small batches (GVQA_OPT_SPLIT3_MIN_MFLOP = 0): head counts, channel counts that
    are multiples of 4 but of nothing else (partial last column block, K not a multiple of 16), graphs of 1 .. 128 nodes, empty
    graphs, nodes without in-edges, multi-edges, hop counts, with and without instruction vectors -- against the oracle, and bit
    for bit against itself when the batch is run a second time."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    r = lambda lo, hi, s: int(synth.randint(1, 7000 + 31 * case + s, lo, hi + 1)[0])
    H = [1, 2, 4, 8][r(0, 3, 1)]
    C = [4, 12, 20, 36, 68, 100, 132, 260][r(0, 7, 2)]
    de, di, K = r(1, 40, 3), (0 if case % 5 == 4 else r(1, 40, 4)), r(1, 5, 5)
    B = r(1, 30, 6)
    gb = synth.make_graph_batch(B, seed=800 + case, nodes_lo=1, nodes_hi=[12, 40, 128][case % 3], rel_per_node=r(0, 25, 7) / 10.0)
    ei = gb.edge_index
    if case % 2 == 0:                                     # nodes without any in-edge: drop every edge into every fourth node
        ei = ei[:, ei[1] % 4 != 0]
    N, E = gb.num_nodes, ei.shape[1]
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=900 + case)
    x, ea, ins = synth.normal((N, C), 1 + case), synth.normal((E, de), 2 + case), synth.normal((K, B, di), 3 + case)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    old_p = _lib.set_option(_lib.OPT_PROJECTION, _lib.PROJECTION_SPLIT2H if scheme == "split2h" else _lib.PROJECTION_SPLIT3)
    try:
        _lib.prof_enable(True); _lib.prof_collect()
        out = _run_gat_seq(dev, (C, de, di, K, H), p, x, ei, ea, ins, gb.batch)
        prof = _lib.prof_collect(); _lib.prof_enable(False)
        out2 = _run_gat_seq(dev, (C, de, di, K, H), p, x, ei, ea, ins, gb.batch)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
        _lib.set_option(_lib.OPT_PROJECTION, old_p)
        _lib.prof_enable(False)
    # (the fused path: no message-passing launches; a coefficient launch per hop, or -- small batches since round 6, GVQA_OPT_HOP_COEFFS = 2 -- none: in-kernel)
    assert prof["mp"][1] == 0 and prof["proj"][1] == K and prof["alpha"][1] in (0, K), "the fused path must have run"
    ref = R.gat_seq(t(x), t(ei), t(ea), t(ins), t(gb.batch), tparams(p), heads=H)
    assert maxabs(out, ref) < TOL, (H, C, de, di, K, B, N, E)
    assert torch.equal(out, out2)


def test_fused_hop_split2h_scales_on_uneven_operands(dev):
    """Two-piece projection inside the fused hop when the operand scales are stressed: node rows whose magnitudes differ by
    10^2 (per-row activation scales), one head's projection weights 10^-4 of the others' (the column block's single weight scale
    then leaves that head's rows with fewer significant bits) -- the forward must stay within fp32-class distance of the oracle,
    relative to the output's own scale, and as close as the three-piece (exact split) form."""
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    H, C, de, di, K = 4, 64, 16, 8, 2
    gb = synth.make_graph_batch(12, seed=321, nodes_lo=8, nodes_hi=60, rel_per_node=1.5)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=17)
    for i in range(K):
        w = p[f"convs.{i}.lin_l.weight"].copy()
        w[C:2 * C] *= 1e-4                                             # head 1
        p[f"convs.{i}.lin_l.weight"] = w
        if f"convs.{i}.lin_r.weight" in p:                            # the reference's lin_r IS lin_l (one Parameter, two keys)
            p[f"convs.{i}.lin_r.weight"] = w
    # (a decade either way: wider row scales make the attention logits so large that ANY fp32 evaluation of the softmax departs
    #  from the fp64 oracle by tens of per cent, whatever the projection arithmetic; the pack tests cover 10^+-30)
    x = synth.normal((N, C), 1) * (10.0 ** (synth.uniform01(N, 5)[:, None] * 2 - 1)).astype(np.float32)
    ea, ins = synth.normal((E, de), 2), synth.normal((K, B, di), 3)
    ref = R.gat_seq(t(x, torch.float64), t(gb.edge_index), t(ea, torch.float64), t(ins, torch.float64), t(gb.batch),
                    tparams(p, torch.float64), heads=H)
    scale = float(ref.abs().max())
    errs = {}
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        for name, mode in (("split2h", _lib.PROJECTION_SPLIT2H), ("split3", _lib.PROJECTION_SPLIT3), ("f32", _lib.PROJECTION_F32)):
            old_p = _lib.set_option(_lib.OPT_PROJECTION, mode)
            try:
                out = _run_gat_seq(dev, (C, de, di, K, H), p, x, gb.edge_index, ea, ins, gb.batch)
            finally:
                _lib.set_option(_lib.OPT_PROJECTION, old_p)
            errs[name] = float((out.double().cpu() - ref).abs().max()) / scale
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    assert errs["split2h"] < 1e-5 and errs["split2h"] < 4 * max(errs["split3"], errs["f32"], 1e-7), errs


def test_eval_forward_is_hip_graph_capturable(dev):
    """Serving with static shapes: the eval forward on a prebuilt batch handle makes no synchronising call, allocation outside
    torch's allocator, or host read, so torch.cuda.graph can capture it; the replay is bit-identical to the eager call."""
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.graph import SceneGraphBatch
    from graphvqa_amd import _lib
    H, C, de, di, K = 4, 64, 24, 16, 3
    gb = synth.make_graph_batch(40, seed=4242, nodes_lo=5, nodes_hi=60, rel_per_node=1.5)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    m = _load_module(gat_seq(C, C, de, di, K, gat_heads=H), synth.gat_seq_params(C, C, de, di, K, H, seed=12), dev)
    args = [t(a, device=dev) for a in (synth.normal((N, C), 1), gb.edge_index, synth.normal((E, de), 2), synth.normal((K, B, di), 3), gb.batch)]
    g = SceneGraphBatch(args[1], args[4], N, B)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)         # the fused hop, as at benchmark size
    try:
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ref = m(*args, graph=g)                              # warm-up: weight cache, LDS attributes
            torch.cuda.current_stream().wait_stream(side)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                out = m(*args, graph=g)
            args[0].mul_(0.5)                                       # new input values in the captured buffers
            ref2 = None
            cg.replay()
            torch.cuda.synchronize()
            got = out.clone()
            ref2 = m(*args, graph=g)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
    assert torch.equal(got, ref2) and not torch.equal(got, ref)


def test_sharded_execution_equals_full_batch(dev):
    """Multi-GPU correctness by construction: the per-rank shards of `parallel.shard_batch` (graphs
    partitioned by edge count), each run through the HIP path on its own, reproduce the rows of the
    full-batch run (graphs are independent; eval BatchNorm is a per-channel affine)."""
    from graphvqa_amd.parallel import shard_batch
    gb = synth.make_graph_batch(37, seed=91, nodes_lo=3, nodes_hi=50, rel_per_node=1.4)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    dims = (64, 40, 48, 5, 4)
    p = synth.gat_seq_params(64, 64, 40, 48, 5, 4, seed=92)
    x, ea, ins = synth.normal((N, 64), 1), synth.normal((E, 40), 2), synth.normal((5, B, 48), 3)
    full = _run_gat_seq(dev, dims, p, x, gb.edge_index, ea, ins, gb.batch).cpu()
    world = 4
    parts = []
    for rank in range(world):
        nsl, emask, ei, b, (g0, g1) = shard_batch(gb.edge_index, gb.batch, B, rank, world)
        parts.append(_run_gat_seq(dev, dims, p, x[nsl], ei, ea[emask], ins[:, g0:g1], b).cpu())
    assert maxabs(torch.cat(parts), full) < 1e-6


def test_per_graph_mean_rows_and_pipelined_steps(dev):
    """The rows a sharded step all-gathers: gvqa_graph_segment_mean (one launch; scatter_mean's max(count, 1) divisor,
    ragged graphs incl. single-node ones) against torch's index_add formulation in fp64; PipelinedSteps on one device
    (no process group) hands every batch's rows back one call later."""
    from graphvqa_amd.graph import SceneGraphBatch
    from graphvqa_amd.parallel import BatchShard, PipelinedSteps, graph_mean_pool
    gb = synth.make_graph_batch(53, seed=0xA11, nodes_lo=1, nodes_hi=40, rel_per_node=1.2)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    for F in (300, 512, 7):
        h = t(synth.normal((N, F), 5), device=dev)
        g = SceneGraphBatch(t(gb.edge_index, device=dev), t(gb.batch, device=dev), N, B)
        got = graph_mean_pool(h, t(gb.batch, device=dev), B, graph=g)
        want = graph_mean_pool(h.double().cpu(), t(gb.batch), B)
        assert got.shape == (B, F) and maxabs(got, want) < 1e-5
    x, ea, ins = synth.normal((N, 16), 1), synth.normal((E, 8), 2), synth.normal((2, B, 8), 3)
    shards = [BatchShard(gb.edge_index, gb.batch, B, x * (i + 1.0), ea, ins, 0, 1, dev) for i in range(3)]
    pipe = PipelinedSteps()
    state = {}

    def fwd(s):
        state["g"] = SceneGraphBatch(s.edge_index, s.batch, s.num_nodes, s.num_graphs)
        return s.x

    pool = lambda h, s: graph_mean_pool(h, s.batch, s.num_graphs, graph=state["g"])
    got = [pipe.step(s, fwd, pool) for s in shards] + [pipe.drain()]
    assert got[0] is None and pipe.drain() is None
    base = graph_mean_pool(t(x).double(), t(gb.batch), B)
    for i in range(3):
        assert maxabs(got[i + 1], base * (i + 1.0)) < 1e-5


def _lcgn_bf16_storage_emulation(x, edge_index, batch, q, lstm, p, x_ctx_init, T=4, slope=0.2, pieces=2):
    """CPU restatement of lcgn_seq.forward with the per-node tensors rounded to bf16 at exactly the points
    where the bf16-node-feature mode stores them, and the node-GEMM weights replaced by the bf16 pieces the
    matrix cores multiply by (fp32 arithmetic otherwise) -- a tight check of the bf16 path; the loose check
    is against the plain fp32 oracle."""
    import torch.nn.functional as F
    from oracle import ref_torch as R
    rb = lambda v: v.bfloat16().float()
    p = dict(p)
    x = rb(x)                                              # the input features are node tensors too
    for k in ("init_sg_emb_input.0.weight", "proj_x_loc.1.weight", "lcgn.lin_l.weight", "lcgn.lin_r.weight", "lcgn.cal_x.weight", "proj_x_ctx.1.weight",
              "output_layer.weight", "fin_layer.weight"):
        p[k] = _bf16_pieces(p[k], pieces)
    O = p["fin_layer.weight"].shape[0]
    Wcat = torch.cat([p["lcgn.lin_l.weight"], p["lcgn.lin_r.weight"], p["lcgn.cal_x.weight"]], 0)     # [3O, 3O]
    x_loc = rb(F.linear(x, p["init_sg_emb_input.0.weight"], p["init_sg_emb_input.0.bias"]))
    q_emb = F.relu(F.linear(q, p["qInput1.weight"], p["qInput1.bias"]))
    proj_x_loc = rb(F.linear(x_loc, p["proj_x_loc.1.weight"], p["proj_x_loc.1.bias"]))
    XL = rb(x_loc @ Wcat[:, :O].T)
    x_ctx = rb(x_ctx_init)
    src, dst = edge_index[0], edge_index[1]
    N = x.shape[0]
    for t_ in range(T):
        cmd = R.lcgn_extract_command(q_emb, lstm, t_, p)
        pc = torch.cat([F.linear(cmd, p["lcgn.proj_cmd.weight"]), F.linear(cmd, p["lcgn.cal_cmd.weight"])], 1)
        prod = rb(F.linear(x_ctx, p["proj_x_ctx.1.weight"], p["proj_x_ctx.1.bias"]) * proj_x_loc)
        J = rb(x_ctx @ Wcat[:, O:2 * O].T + prod @ Wcat[:, 2 * O:].T + XL)
        x_l, x_r, x_val = J[:, :O], J[:, O:2 * O], J[:, 2 * O:]
        logit = (x_l[src] * (pc[batch[dst], :O] * x_r[dst])).sum(-1, keepdim=True)
        alpha = R.segment_softmax(F.leaky_relu(logit, slope), dst, N)
        msg = rb(R.scatter_add_rows(alpha * x_val[src], dst, N) * pc[batch, O:] + p["lcgn.bias"])
        x_ctx = rb(F.linear(torch.cat([x_ctx, msg], 1), p["output_layer.weight"], p["output_layer.bias"]))
    out = F.linear(x_loc, p["fin_layer.weight"][:, :O], p["fin_layer.bias"])
    return x_ctx @ p["fin_layer.weight"][:, O:].T + out


@pytest.mark.parametrize("pieces", [2, 1])
def test_lcgn_bf16_node_features(dev, pieces):
    """BASELINE config 5: LCGN with the per-node tensors stored as bf16 and the node GEMMs on the bf16 matrix
    cores (fp32 accumulation; weights as 2 bf16 pieces, or 1).  Stated bounds: <= 3e-3 max-abs (of the output
    scale) against a CPU emulation that rounds at the same storage points and uses the same weight pieces;
    against the plain fp32 oracle the deviation is that of bf16 storage (<= 3 % of the output scale here, 5 %
    with single-piece weights); the fp32 mode keeps 1e-4."""
    from oracle import ref_torch as R
    from graphvqa_amd.lcgn import lcgn_seq
    gb = synth.config2_batch()                 # the batch config 5 is defined on: 1000 graphs, ~30k nodes
    N, B, O, L = gb.num_nodes, gb.num_graphs, 512, 10
    p = synth.lcgn_seq_params(300, O, seed=808)
    x, q, lstm = synth.normal((N, 300), 1), synth.normal((B, O), 2), synth.normal((L, B, O), 3)
    x_ctx = synth.normal((N, O), 4)
    m = _load_module(lcgn_seq(300, O, 300, 5, node_feature_dtype=torch.bfloat16, bf16_weight_pieces=pieces), p, dev)
    args = [t(a, device=dev) for a in (x, gb.edge_index, gb.batch, q, lstm)]
    out = m(*args, x_ctx_init=t(x_ctx, device=dev))
    emu = _lcgn_bf16_storage_emulation(t(x), t(gb.edge_index), t(gb.batch), t(q), t(lstm), tparams(p), t(x_ctx),
                                       pieces=pieces)
    ref = R.lcgn_seq(t(x), t(gb.edge_index), t(gb.batch), t(q), t(lstm), tparams(p), t(x_ctx))
    scale = float(ref.abs().max())
    # the ORACLE's model of the storage choice (VERDICT r04 #3c): the reference's op sequence in fp64 with every per-node tensor
    # rounded to bf16 where the reference materialises it (oracle/ref_torch.lcgn_seq(node_store=bf16_storage)); weights exact.
    # What is left between it and the device is the weight pieces (two bf16 pieces: 2^-17 relative), fp32 accumulation and the
    # roundings that flip between the two -- bounded at 2.5e-3 of the output scale for two-piece weights, 4 x tighter than the
    # storage effect itself (~1e-2 here); single-piece weights (bf16 weights: another 2^-9 on every product) are not held to it
    d64 = lambda v: t(v).double()
    ref_st = R.lcgn_seq(d64(x), t(gb.edge_index), t(gb.batch), d64(q), d64(lstm), {k: v.double() for k, v in tparams(p).items()}, d64(x_ctx),
                        node_store=R.bf16_storage).float()
    print("lcgn bf16 pieces=%d: vs emulation %.3e, vs bf16-storage fp64 oracle %.3e, vs fp32 oracle %.3e, scale %.3e" %
          (pieces, maxabs(out, emu), maxabs(out, ref_st), maxabs(out, ref), scale))
    assert maxabs(out, emu) < 3e-3 * max(scale, 1.0)
    if pieces == 2:
        assert maxabs(out, ref_st) < 2.5e-3 * max(scale, 1.0)
    assert maxabs(out, ref) < (3e-2 if pieces == 2 else 5e-2) * scale
    m32 = _load_module(lcgn_seq(300, O, 300, 5), p, dev)
    out32 = m32(*args, x_ctx_init=t(x_ctx, device=dev))
    assert maxabs(out32, ref) < TOL
    assert maxabs(out32, out) > 1e-5          # the two modes really differ (bf16 rounding is visible)


def test_forward_is_hip_graph_capturable(dev):
    """The library only enqueues work on the caller's stream (no hidden synchronisation or allocation): with a prebuilt
    batch handle the whole gat_seq forward can be captured in a HIP graph (torch.cuda.CUDAGraph) and replayed on new
    input values, bit-identically to the eager launches."""
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.graph import SceneGraphBatch
    gb = synth.make_graph_batch(16, seed=0xCA9, nodes_lo=8, nodes_hi=30, rel_per_node=1.5)
    N, E, B, D = gb.num_nodes, gb.num_edges, gb.num_graphs, 64
    m = _load_module(gat_seq(D, D, D, 96, 3, dropout=0.1, gat_heads=4), synth.gat_seq_params(D, D, D, 96, 3, 4, seed=3), dev)
    x, ea, ins = [t(a, device=dev) for a in (synth.normal((N, D), 1), synth.normal((E, D), 2), synth.normal((3, B, 96), 3))]
    ei, b = t(gb.edge_index, device=dev), t(gb.batch, device=dev)
    g = SceneGraphBatch(ei, b, N, B)                     # reads statistics back: outside the capture
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            m(x, ei, ea, ins, b, graph=g)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, stream=s):
        out = m(x, ei, ea, ins, b, graph=g)
    x.copy_(t(synth.normal((N, D), 11), device=dev))     # new values in the captured input buffer
    cg.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, m(x, ei, ea, ins, b, graph=g))


def test_linear_more_row_tiles_than_grid_y(dev):
    """M beyond 65535 row tiles (8.39 M rows: the edge-logit product of a 65536-graph batch) runs as row chunks."""
    from graphvqa_amd import _lib
    lib = _lib.load()
    M, N, K = 65535 * 128 + 300, 8, 8
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    bias = torch.randn(N, device=dev)
    add = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.gvqa_linear_f32_ex(M, N, K, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), add.data_ptr(), N, None, 0, 1,
                                      out.data_ptr(), N, st))
    ref = torch.relu(A @ B.T + bias + add)
    assert float((out - ref).abs().max()) < 1e-4
    assert float((out[-300:] - ref[-300:]).abs().max()) < 1e-4          # the rows of the second chunk


def test_randomized_variants_head_encoder_vs_oracle(dev):
    """25 random batches (empty graphs, widths 8 / 12 / 30 / 64, ragged sizes): tapped GINE / GCN conv results, lcgn_seq,
    pooling + classifier and the scene-graph encoder on the fused HIP paths against the oracle."""
    import types
    from graphvqa_amd.baseline_models import gine_seq, gcn_seq
    from graphvqa_amd.lcgn import lcgn_seq
    from graphvqa_amd.pipeline_head import MyConditionalGlobalAttention, ShortAnswerClassifier
    from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
    from oracle import ref_torch as R

    def t(a, d=None):
        x = torch.from_numpy(np.ascontiguousarray(a))
        return x.to(d) if d else x
    tp = lambda p: {k: t(v) for k, v in p.items()}
    worst = {}
    rng = np.random.default_rng(77)
    def upd(name, a, b):
        e = float((a.detach().cpu().double() - b.detach().double()).abs().max()) if a.numel() else 0.0
        worst[name] = max(worst.get(name, 0.0), e)
    def rand_batch():
        B = int(rng.integers(1, 10)); sizes = rng.integers(0, 40, size=B)
        if sizes.sum() == 0: sizes[0] = 2
        batch = np.repeat(np.arange(B), sizes).astype(np.int64); offs = np.concatenate([[0], np.cumsum(sizes)])
        src, dst = [np.zeros(0, np.int64)], [np.zeros(0, np.int64)]
        for g in range(B):
            n = int(sizes[g])
            if n == 0: continue
            e = int(rng.integers(0, 3 * n + 1))
            src.append(np.concatenate([np.arange(n), rng.integers(0, n, size=e)]) + offs[g]); dst.append(np.concatenate([np.arange(n), rng.integers(0, n, size=e)]) + offs[g])
        return B, batch, np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)
    for case in range(25):
        B, batch, ei = rand_batch(); N, E = batch.shape[0], ei.shape[1]
        D = int(rng.choice([8, 12, 30, 64])); Di = int(rng.choice([8, 16]))
        x, ea, ins = rng.standard_normal((N, D)).astype(np.float32), rng.standard_normal((E, D)).astype(np.float32), rng.standard_normal((5, B, Di)).astype(np.float32)
        # GINE / GCN tapped convs
        p = synth.gine_seq_params(D, D, Di, seed=case); m = gine_seq(D, D, Di); m.load_state_dict(tp(p)); m = m.to(dev).eval()
        out, convs = m(t(x, dev), t(ei, dev), t(ea, dev), t(ins, dev), t(batch, dev), return_convs=True)
        ro, rc = R.gine_seq(t(x), t(ei), t(ea), t(ins), t(batch), tp(p), return_convs=True)
        upd("gine.out", out, ro); [upd("gine.conv", a, b) for a, b in zip(convs, rc)]
        p = synth.gcn_seq_params(D, D, Di, seed=case); m = gcn_seq(D, D, Di); m.load_state_dict(tp(p)); m = m.to(dev).eval()
        out, convs = m(t(x, dev), t(ei, dev), t(ins, dev), t(batch, dev), return_convs=True)
        ro, rc = R.gcn_seq(t(x), t(ei), t(ins), t(batch), tp(p), return_convs=True)
        upd("gcn.out", out, ro); [upd("gcn.conv", a, b) for a, b in zip(convs, rc)]
        # LCGN
        O = int(rng.choice([8, 16, 40])); L = int(rng.integers(1, 7))
        p = synth.lcgn_seq_params(D, O, seed=case, cmd_dim=O, question_dim=O); m = lcgn_seq(D, O, D, 5, gat_cmd_dim=O, question_dim=O)
        m.load_state_dict(tp(p), strict=False); m = m.to(dev).eval()
        q, lstm, xc = rng.standard_normal((B, O)).astype(np.float32), rng.standard_normal((L, B, O)).astype(np.float32), rng.standard_normal((N, O)).astype(np.float32)
        out = m(t(x, dev), t(ei, dev), t(batch, dev), t(q, dev), t(lstm, dev), x_ctx_init=t(xc, dev))
        upd("lcgn", out, R.lcgn_seq(t(x), t(ei), t(batch), t(q), t(lstm), tp(p), t(xc)))
        # head
        Q, A = int(rng.choice([8, 24])), int(rng.choice([5, 33]))
        pp, pc = synth.attention_pool_params(D, Q, seed=case), synth.classifier_params(Q, 16, A, seed=case)
        pool, clf = MyConditionalGlobalAttention(D, Q), ShortAnswerClassifier(Q, 16, A)
        pool.load_state_dict(tp(pp)); clf.load_state_dict(tp(pc)); pool, clf = pool.to(dev).eval(), clf.to(dev).eval()
        u = rng.standard_normal((B, Q)).astype(np.float32)
        upd("head", clf(pool(t(x, dev), t(u, dev), t(batch, dev)), t(u, dev)), R.short_answer_logits(R.global_attention_pool(t(x), t(u), t(batch), tp(pp), B), t(u), tp(pc)))
        # encoder
        V = 40; pe = synth.encoder_params(V, D, seed=case); enc = GroundTruth_SceneGraph_Encoder(V, 0, D); enc.load_state_dict(tp(pe)); enc = enc.to(dev).eval()
        xt, et = rng.integers(0, V, size=(N, 12)), rng.integers(1, V, size=(E, 1)); added = rng.choice(E, size=min(E, 5), replace=False).astype(np.int64) if E else np.zeros(0, np.int64)
        data = types.SimpleNamespace(x=t(xt, dev), edge_attr=t(et, dev), edge_index=t(ei, dev), batch=t(batch, dev), added_sym_edge=t(added, dev))
        xe, ee, _ = enc(data); rxe, ree = R.scene_graph_encoder(t(xt), t(ei), t(et), t(added), t(batch), B, tp(pe))
        upd("enc.x", xe, rxe); upd("enc.e", ee, ree)
    assert all(v < 1e-4 for v in worst.values()), worst


def test_host_layout_handle_and_weight_cache_give_identical_results(dev):
    """The sync-free batch handle (loader-side per-graph layout, gvqa_graph_finalize_host) must equal the read-back one
    field by field, and forwards through the weight cache must be bit-identical to uncached ones -- also after the
    parameters change in place (the cache key follows the version counters)."""
    from graphvqa_amd import _lib
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.graph import SceneGraphBatch, HostLayout
    H, C, de, di, K = 4, 64, 24, 16, 3
    gb = synth.make_graph_batch(37, seed=4242, nodes_lo=1, nodes_hi=50, rel_per_node=1.4)
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    ei, batch = t(gb.edge_index, device=dev), t(gb.batch, device=dev)
    g1 = SceneGraphBatch(ei, batch, N, B)
    g2 = SceneGraphBatch(ei, batch, N, B, host_layout=HostLayout.from_numpy(gb.edge_index, gb.batch, B))
    for f in ("max_graph_nodes", "max_graph_edges", "max_in_degree", "intra_graph", "valid", "finalized", "num_row_groups",
              "max_row_group_edges"):
        assert getattr(g1.c, f) == getattr(g2.c, f), f
    torch.cuda.synchronize()
    rg = lambda g: g._ws[g.c.row_group_ptr - g._ws.data_ptr():][:4 * (g.c.num_row_groups + 1)].view(torch.int32)
    assert torch.equal(rg(g1), rg(g2)) and torch.equal(g1.rowptr, g2.rowptr) and torch.equal(g1.csr_src, g2.csr_src)
    with pytest.raises(_lib.GvqaError):
        SceneGraphBatch(ei, batch, N, B, host_layout=HostLayout(np.arange(B + 1), np.arange(B + 1)))      # does not span the batch
    p = synth.gat_seq_params(C, C, de, di, K, H, seed=31)
    x, ea, ins = t(synth.normal((N, C), 1), device=dev), t(synth.normal((E, de), 2), device=dev), t(synth.normal((K, B, di), 3), device=dev)
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        m = _load_module(gat_seq(C, C, de, di, K, dropout=0.0, gat_heads=H), p, dev)
        a = m(x, ei, ea, ins, batch, graph=g1)
        key1 = m._wc_key
        b = m(x, ei, ea, ins, batch, graph=g2)                 # second call: served from the cache
        assert m._wc_key == key1 and torch.equal(a, b)
        with torch.no_grad():
            m.convs[1].lin_l.weight.mul_(1.5)                  # in place: version bump -> re-prepared
        c = m(x, ei, ea, ins, batch, graph=g2)
        assert m._wc_key != key1 and maxabs(c, a) > 1e-3
        fresh = _load_module(gat_seq(C, C, de, di, K, dropout=0.0, gat_heads=H), {k: v.cpu().numpy() for k, v in m.state_dict().items()}, dev)
        assert torch.equal(fresh(x, ei, ea, ins, batch, graph=g1), c)
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)


def test_scene_graph_builder_to_device_feeds_the_path(dev):
    """SURVEY 8f-3 end to end: scene-graph dicts -> one tokenising pass (flatten_scene_graphs) -> the library's NATIVE collate
    (gvqa_scene_graph_collate, csrc/collate.hip) -> DeviceSceneGraphs (token tensors + CSR handle from the loader-side layout,
    no read-back) -> encoder -> gat_seq.  The device tensors are pinned to what the reference's own converter produced
    (tests/golden/sg_builder_debug4.npz from gqa_dataset_entry.py:190-372): topology, added_sym_edge and edge tokens exactly,
    node tokens as multisets (the reference iterates a Python set); the result equals the explicit, synchronising batch
    handle's, whose read-back statistics equal the handle's."""
    from graphvqa_amd.scene_graph import flatten_scene_graphs, collate_flat_scene_graphs
    from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
    from graphvqa_amd.gat_skip import gat_seq
    from graphvqa_amd.graph import SceneGraphBatch
    meta, g = load_golden("sg_builder_debug4")
    stoi = {w: i for i, w in enumerate(meta["itos"])}
    sgs = [meta["scene_graphs"][k] for k in meta["graphs"]]
    c = collate_flat_scene_graphs(flatten_scene_graphs(sgs, stoi))
    d = c.to(dev)
    sizes = meta["sizes"]
    n_off = np.concatenate([[0], np.cumsum([s_[0] for s_ in sizes])])
    e_off = np.concatenate([[0], np.cumsum([s_[1] for s_ in sizes])])
    assert (d.num_nodes, d.num_edges) == (int(n_off[-1]), int(e_off[-1]))
    ei_d, ea_d, x_d = d.edge_index.cpu().numpy(), d.edge_attr.cpu().numpy(), d.x.cpu().numpy()
    for idx in range(len(sgs)):
        assert np.array_equal(ei_d[:, e_off[idx]:e_off[idx + 1]], g[f"g{idx}.edge_index"] + n_off[idx])
        assert np.array_equal(ea_d[e_off[idx]:e_off[idx + 1]], g[f"g{idx}.edge_attr"])
        assert np.array_equal(x_d[n_off[idx]:n_off[idx + 1], 0], g[f"g{idx}.x"][:, 0])
        assert np.array_equal(np.sort(x_d[n_off[idx]:n_off[idx + 1]], axis=1), np.sort(g[f"g{idx}.x"], axis=1))
    assert np.array_equal(d.added_sym_edge.cpu().numpy(), np.concatenate([g[f"g{i}.added_sym_edge"] + e_off[i] for i in range(len(sgs))]))
    assert np.array_equal(d.batch.cpu().numpy(), np.repeat(np.arange(len(sgs)), [s_[0] for s_ in sizes]))
    # the device CSR built from these arrays: every COO edge sits in its destination's row, rows in COO order
    rowptr, csr_src, csr_eid = (getattr(d.graph, k).cpu().numpy() for k in ("rowptr", "csr_src", "csr_eid"))
    order = np.argsort(ei_d[1], kind="stable")
    assert np.array_equal(csr_eid, order) and np.array_equal(csr_src, ei_d[0][order])
    assert np.array_equal(rowptr, np.concatenate([[0], np.cumsum(np.bincount(ei_d[1], minlength=d.num_nodes))]))
    V = len(meta["itos"])
    enc = _load_module(GroundTruth_SceneGraph_Encoder(V, stoi["<pad>"], 300), synth.encoder_params(V, 300, seed=5, pad_idx=stoi["<pad>"]), dev)
    gs = _load_module(gat_seq(300, 300, 300, 512, 5, dropout=0.1, gat_heads=4), synth.gat_seq_params(300, 300, 300, 512, 5, 4, seed=6), dev)
    ins = t(synth.normal((5, c.num_graphs, 512), 7), device=dev)
    xe, ee, _ = enc(d, graph=d.graph)
    h = gs(xe, d.edge_index, ee, ins, d.batch, graph=d.graph)
    g2 = SceneGraphBatch(d.edge_index, d.batch, c.num_nodes, c.num_graphs)         # statistics read back from the device
    for f in ("max_graph_nodes", "max_graph_edges", "max_in_degree", "intra_graph", "num_row_groups", "max_row_group_edges"):
        assert getattr(d.graph.c, f) == getattr(g2.c, f), f
    xe2, ee2, _ = enc(d, graph=g2)
    assert torch.equal(xe, xe2) and torch.equal(h, gs(xe2, d.edge_index, ee2, ins, d.batch, graph=g2))
    assert h.shape == (c.num_nodes, 300) and torch.isfinite(h).all()


def test_scene_graph_encoder_on_the_two_piece_products(dev):
    """The encoder's large-batch form (csrc/encoder.hip: edge block of the first Linear applied to the embedding table, the
    edge_attr' product folded into the next Linear, all node- / edge-sized products on the two-piece kernels) against (i) the
    reference's own encoder outputs (golden, with the size threshold at 0 so that the form is taken), (ii) the oracle on 300
    ragged graphs at the real width d = 300 with negated symmetric edges, (iii) the small-batch form (f32-input products) on the
    same inputs."""
    import types
    from oracle import ref_torch as R
    from graphvqa_amd import _lib
    from graphvqa_amd.sg_encoder import GroundTruth_SceneGraph_Encoder
    old = _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, 0)
    try:
        meta, g = load_golden("sg_encoder_debug4")
        E0 = g["edge_index"].shape[1]
        if True:                                        # (the module caches the projected table with the weights: any vocabulary size)
            enc = GroundTruth_SceneGraph_Encoder(meta["vocab"], meta["pad_idx"], meta["dim"])
            _load_module(enc, synth.encoder_params(meta["vocab"], meta["dim"], seed=meta["param_seed"], pad_idx=meta["pad_idx"]), dev)
            data = types.SimpleNamespace(x=t(g["x_tokens"], device=dev), edge_attr=t(g["edge_tokens"], device=dev),
                                         edge_index=t(g["edge_index"], device=dev), batch=t(g["batch"], device=dev),
                                         added_sym_edge=t(g["added_sym_edge"], device=dev))
            xe, ee, _ = enc(data)
            assert maxabs(ee, g["edge_attr_encoded"]) < 5e-5 and maxabs(xe, g["x_encoded"]) < TOL
        gb = synth.make_graph_batch(300, seed=0xE1C, nodes_lo=3, nodes_hi=40, rel_per_node=2.0)
        N, E, B, V, D = gb.num_nodes, gb.num_edges, gb.num_graphs, 1500, 300
        assert V <= E
        pe = synth.encoder_params(V, D, seed=21)
        xt = synth.randint(N * 12, 5, 0, V, stream=3).reshape(N, 12); et = synth.randint(E, 6, 1, V, stream=3).reshape(E, 1)
        added = np.arange(0, E, 5, dtype=np.int64)
        enc = _load_module(GroundTruth_SceneGraph_Encoder(V, 0, D), pe, dev)
        data = types.SimpleNamespace(x=t(xt, device=dev), edge_attr=t(et, device=dev), edge_index=t(gb.edge_index, device=dev),
                                     batch=t(gb.batch, device=dev), added_sym_edge=t(added, device=dev))
        xe, ee, _ = enc(data)
        rxe, ree = R.scene_graph_encoder(t(xt), t(gb.edge_index), t(et), t(added), t(gb.batch), B, tparams(pe))
        assert maxabs(ee, ree) < 1e-4 * (1.0 + float(ree.abs().max())) and maxabs(xe, rxe) < TOL
        prev = _lib.set_option(_lib.OPT_PROJECTION, _lib.PROJECTION_F32)        # the f32-input form on the same inputs
        try:
            xe32, ee32, _ = enc(data)
        finally:
            _lib.set_option(_lib.OPT_PROJECTION, prev)
        assert maxabs(ee, ee32) < 1e-4 * (1.0 + float(ree.abs().max())) and maxabs(xe, xe32) < TOL
        # (iv) the weight-only forms are cached with the module: a weight modified in place (version counter) or replaced makes the
        # next call rebuild them -- same result as a module loaded with the new weights
        assert enc._packed is None                      # (the f32-input call just made has no packed forms)
        pe2 = {k: v.copy() for k, v in pe.items()}
        k2 = "scene_graph_encoding_layer.node_model.node_mlp_1.2.weight"
        pe2[k2] = (pe2[k2] * 1.5).astype(np.float32)
        enc(data)
        with torch.no_grad():
            dict(enc.named_parameters())[k2].mul_(1.5)
        enc(data)
        key = enc._packed_key
        xe_b, ee_b, _ = enc(data)
        assert enc._packed is not None and enc._packed_key == key
        xe_c, ee_c, _ = _load_module(GroundTruth_SceneGraph_Encoder(V, 0, D), pe2, dev)(data)
        assert torch.equal(xe_b, xe_c) and torch.equal(ee_b, ee_c) and maxabs(xe_b, xe) > 1e-4
        # (iv-b) ADVICE r04: an edit through .data does not bump the version counter -- the documented hook drops the cached forms;
        # a forward on ANOTHER stream than the one that packed waits for the pack (event), and reuses the cache
        dict(enc.named_parameters())[k2].data.mul_(2.0)
        xe_stale, _, _ = enc(data)
        assert torch.equal(xe_stale, xe_b)                # (the stale result: what the hook is for)
        enc.invalidate_weight_cache()
        pe3 = dict(pe2); pe3[k2] = (pe2[k2] * 2.0).astype(np.float32)
        side = torch.cuda.Stream()
        xe_d, ee_d, _ = enc(data)
        key = enc._packed_key
        with torch.cuda.stream(side):
            xe_s, ee_s, _ = enc(data)
        side.synchronize()
        assert enc._packed_key == key
        xe_e, ee_e, _ = _load_module(GroundTruth_SceneGraph_Encoder(V, 0, D), pe3, dev)(data)
        assert torch.equal(xe_d, xe_e) and torch.equal(ee_d, ee_e) and torch.equal(xe_s, xe_e) and torch.equal(ee_s, ee_e)
        dict(enc.named_parameters())[k2].data.mul_(0.5)
        enc.invalidate_weight_cache()
        # (v) several tokens per edge: the token sums are formed first (the one-token gather inside the pack pass does not apply)
        et2 = synth.randint(E * 2, 8, 1, V, stream=3).reshape(E, 2)
        data2 = types.SimpleNamespace(x=data.x, edge_attr=t(et2, device=dev), edge_index=data.edge_index, batch=data.batch, added_sym_edge=data.added_sym_edge)
        xe2, ee2, _ = enc(data2)
        rxe2, ree2 = R.scene_graph_encoder(t(xt), t(gb.edge_index), t(et2), t(added), t(gb.batch), B, tparams(pe2))
        assert maxabs(ee2, ree2) < 1e-4 * (1.0 + float(ree2.abs().max())) and maxabs(xe2, rxe2) < TOL
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_MIN_MFLOP, old)
