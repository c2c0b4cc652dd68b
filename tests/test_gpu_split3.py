"""The fp32-accurate projections on the 16-bit matrix cores (csrc/split3.hip) -- "split3" (three exact bf16 pieces, six
products) and "split2h" (two scaled fp16 pieces, three products; the default) -- against fp64 and against the f32-input MFMA
kernel, through the C ABI.  The hop projection they serve is /root/reference gat_skip.py:133."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# kernel variants (tile geometry x schedule x epilogue, see launch_linear_split); 0 = the library's own choice
VARIANTS = [0, 10, 11, 14, 21, 28, 29, 30, 34]
SCHEME_VARIANTS = [("split3", v) for v in VARIANTS] + [("split2h", v) for v in (0, 114, 118, 124, 134)]   # (112: even k-block counts only, own test below)


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from graphvqa_amd import _lib
    return _lib, _lib.load(), torch.device("cuda:0")


def _pack(env, X, scheme="split3"):
    _lib, lib, dev = env
    rows, K = X.shape
    nbytes, pack = ((lib.gvqa_split3_packed_bytes, lib.gvqa_split3_pack) if scheme == "split3"
                    else (lib.gvqa_split2h_packed_bytes, lib.gvqa_split2h_pack))
    buf = torch.empty(nbytes(rows, K), dtype=torch.uint8, device=dev)
    _lib.check(pack(rows, K, X.data_ptr(), X.stride(0), buf.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return buf


def _unpack(buf, rows, K):
    """numpy inverse of the fragment-major layout -> three fp32 arrays [rows, K] (pieces widened to fp32)."""
    RT, KB = -(-rows // 32), -(-K // 16)
    raw = buf.cpu().numpy().view(np.uint16).reshape(RT, KB, 3, 64, 8)
    f = (raw.astype(np.uint32) << 16).view(np.float32)                    # [RT, KB, 3, lane, e]
    f = f.reshape(RT, KB, 3, 2, 32, 8)                                    # lane = khalf * 32 + r
    f = f.transpose(2, 0, 4, 1, 3, 5).reshape(3, RT * 32, KB * 16)        # [piece, row, k]
    return f


@pytest.mark.parametrize("rows,K", [(32, 16), (70, 40), (129, 300), (5, 7)])
def test_pack_pieces_are_an_exact_split(env, rows, K):
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(rows * 1000 + K)
    X = (torch.randn(rows, K, generator=g) * torch.exp(4 * torch.randn(rows, K, generator=g))).to(dev)
    X[0, 0] = 0.0
    pieces = _unpack(_pack(env, X), rows, K).astype(np.float64)
    x = X.cpu().numpy().astype(np.float64)
    total = pieces[0] + pieces[1] + pieces[2]
    assert np.array_equal(total[:rows, :K], x)                            # exact three-piece decomposition
    assert not total[rows:].any() and not total[:, K:].any()               # zero padding
    with np.errstate(divide="ignore", invalid="ignore"):
        assert np.all(np.abs(pieces[1][:rows, :K]) <= np.abs(x) * 2.0 ** -8 + 1e-300)
        assert np.all(np.abs(pieces[2][:rows, :K]) <= np.abs(x) * 2.0 ** -16 + 1e-300)


def _unpack2h(buf, rows, K):
    """split2h: (pieces [2, rows_padded, K_padded] as fp64 in the SCALED domain, inverse scales [rows_padded])."""
    RT, KB = -(-rows // 32), -(-K // 16)
    raw = buf.cpu().numpy()
    f = raw[:RT * KB * 2048].view(np.float16).reshape(RT, KB, 2, 2, 32, 8)        # [RT, KB, piece, khalf, r, e]
    f = f.transpose(2, 0, 4, 1, 3, 5).reshape(2, RT * 32, KB * 16).astype(np.float64)
    inv = raw[RT * KB * 2048:].view(np.float32)
    assert inv.shape == (RT * 32,)
    return f, inv.astype(np.float64)


@pytest.mark.parametrize("rows,K", [(32, 16), (70, 40), (129, 300), (5, 7), (64, 512), (40, 700)])
def test_pack_split2h_pieces(env, rows, K):
    """Row scale: the largest magnitude lands in [2^13, 2^14); pieces are RN16(x s) and RN16(x s - p1), bit for bit; their sum
    misses x s by at most 2^-22 |x s| (or half an fp16 subnormal step); the stored inverse scale undoes s exactly."""
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(rows * 1000 + K)
    X = (torch.randn(rows, K, generator=g) * torch.exp(4 * torch.randn(rows, K, generator=g)))
    X[0, 0] = 0.0
    if rows > 3:
        X[3] = 0.0                                                        # an all-zero row keeps scale 1
        X[2] *= 1e-30                                                     # tiny and huge rows get their own scales
        X[1] *= 1e30
    X = X.to(dev)
    pieces, inv = _unpack2h(_pack(env, X, "split2h"), rows, K)
    x = X.cpu().numpy().astype(np.float64)
    scale = 1.0 / inv
    assert np.all(np.log2(scale) == np.round(np.log2(scale)))              # powers of two
    xs = np.zeros_like(pieces[0]); xs[:rows, :K] = x * scale[:rows, None]
    mx = np.abs(xs).max(axis=1)
    nz = mx > 0
    assert np.all((mx[nz] >= 2.0 ** 13) & (mx[nz] < 2.0 ** 14)) and np.all(scale[~nz] == 1.0)
    p1 = xs.astype(np.float32).astype(np.float16).astype(np.float64)       # x s is exact in fp32 (power-of-two scale)
    p2 = (xs - p1).astype(np.float32).astype(np.float16).astype(np.float64)
    assert np.array_equal(pieces[0], p1) and np.array_equal(pieces[1], p2)
    assert np.all(np.abs(xs - pieces[0] - pieces[1]) <= np.maximum(np.abs(xs) * 2.0 ** -22, 2.0 ** -25))
    assert not pieces[:, rows:].any() and not pieces[:, :, K:].any()       # zero padding


def _run(env, A, W, tile, bias=None, addend=None, mul=None, relu=0, scheme="split3"):
    _lib, lib, dev = env
    M, K = A.shape
    N = W.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    a, w = _pack(env, A, scheme), _pack(env, W, scheme)
    C = torch.full((M, N), float("nan"), device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None
    old = _lib.set_option(_lib.OPT_SPLIT3_VARIANT, tile)
    linear = lib.gvqa_linear_split3 if scheme == "split3" else lib.gvqa_linear_split2h
    try:
        _lib.check(linear(M, N, K, a.data_ptr(), w.data_ptr(), ptr(bias), ptr(addend),
                                          addend.stride(0) if addend is not None else 0, ptr(mul),
                                          mul.stride(0) if mul is not None else 0, relu, C.data_ptr(), C.stride(0), st))
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_VARIANT, old)
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("scheme,tile", SCHEME_VARIANTS)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 132, 40), (1000, 1200, 300), (33, 4, 16), (513, 520, 512), (2048, 512, 1024)])
def test_linear_split_matches_fp64_like_fp32(env, scheme, tile, M, N, K):
    """Error against fp64 must be in the class of an exact-fp32 k-ordered fmaf chain (the f32 MFMA kernel)."""
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(M + 7 * N + 13 * K)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    C = _run(env, A, W, tile, scheme=scheme)
    ref = A.double() @ W.double().t()
    err = float((C.double() - ref).abs().max())
    C32 = torch.empty(M, N, device=dev)
    _lib.check(lib.gvqa_linear_f32(M, N, K, A.data_ptr(), K, W.data_ptr(), K, None, 0, C32.data_ptr(), N,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    err32 = float((C32.double() - ref).abs().max())
    assert torch.isfinite(C).all()
    assert err <= max(2.0 * err32, 2e-6), (err, err32)


@pytest.mark.parametrize("scheme,tile", SCHEME_VARIANTS)
def test_linear_split_epilogues(env, scheme, tile):
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 260, 136, 48
    A, W = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev)
    bias, add, mul = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    base = A.double() @ W.double().t()
    for kw, ref in (({"bias": bias}, base + bias.double()),
                    ({"addend": add}, base + add.double()),
                    ({"bias": bias, "addend": add, "mul": mul, "relu": 1}, torch.relu((base + bias.double() + add.double()) * mul.double())),
                    ({"relu": 2}, torch.where(base > 0, base, torch.expm1(base)))):
        C = _run(env, A, W, tile, scheme=scheme, **kw)
        assert float((C.double() - ref).abs().max()) < 2e-5, kw.keys()


@pytest.mark.parametrize("scheme", ["split3", "split2h"])
def test_linear_split_wide_dynamic_range(env, scheme):
    """Pieces keep fp32 accuracy when operand magnitudes span many binades inside a row and between rows (where a plain bf16 /
    fp16 product, or an unscaled fp16 split, would not)."""
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(11)
    M, N, K = 384, 256, 512
    A = (torch.randn(M, K, generator=g) * torch.exp(3 * torch.randn(M, K, generator=g))).to(dev)
    W = (torch.randn(N, K, generator=g) * torch.exp(3 * torch.randn(N, K, generator=g))).to(dev)
    A *= torch.exp(10 * torch.randn(M, 1, generator=g)).to(dev)           # rows 2^+-30 apart
    W *= torch.exp(10 * torch.randn(N, 1, generator=g)).to(dev)
    C = _run(env, A, W, 0, scheme=scheme)
    C32 = torch.empty(M, N, device=dev)
    _lib.check(lib.gvqa_linear_f32(M, N, K, A.data_ptr(), K, W.data_ptr(), K, None, 0, C32.data_ptr(), N,
                                   torch.cuda.current_stream().cuda_stream))
    ref = A.double() @ W.double().t()
    scale = (A.double().abs() @ W.double().abs().t())                       # sum |a b|: the natural error scale
    rel = float(((C.double() - ref).abs() / scale).max())
    rel32 = float(((C32.double() - ref).abs() / scale).max())
    # the f32-MFMA fmaf chain's own roundoff class; split2h's representation bound is 3 * 2^-22 of sum |a b| (both operands and
    # the dropped p2 q2), reached only when one term carries the whole sum
    assert rel < max(2 * rel32, 2e-7 if scheme == "split3" else 4e-7), (rel, rel32)


def test_linear_split3_rejects_unaligned(env):
    _lib, lib, dev = env
    A, W = torch.randn(32, 16, device=dev), torch.randn(6, 16, device=dev)
    a, w = _pack(env, A), _pack(env, W)
    C = torch.empty(32, 6, device=dev)
    rc = lib.gvqa_linear_split3(32, 6, 16, a.data_ptr(), w.data_ptr(), None, None, 0, None, 0, 0, C.data_ptr(), 6,
                                torch.cuda.current_stream().cuda_stream)
    assert rc == _lib.E_UNSUPPORTED


def test_split2h_zero_inf_rows(env):
    """All-zero rows give exact zeros; an infinite operand value poisons its own output row only."""
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(3)
    A, W = torch.randn(64, 32, generator=g).to(dev), torch.randn(32, 32, generator=g).to(dev)
    A[5] = 0.0
    A[7, 3] = float("inf")
    C = _run(env, A, W, 0, scheme="split2h")
    assert not C[5].any() and not torch.isfinite(C[7]).any()
    keep = [i for i in range(64) if i != 7]
    assert float((C[keep].double() - A[keep].double() @ W.double().t()).abs().max()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (513, 520, 512), (2048, 512, 1024), (300, 132, 32)])
def test_linear_split2h_two_k_steps_per_barrier(env, M, N, K):
    """Variant 112: two K steps per LDS stage and per barrier (even number of k blocks only)."""
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    C = _run(env, A, W, 112, scheme="split2h")
    C0 = _run(env, A, W, 114, scheme="split2h")
    ref = A.double() @ W.double().t()
    assert float((C.double() - ref).abs().max()) < max(2e-6, 2 * float((C0.double() - ref).abs().max()))


@pytest.mark.parametrize("M,K1,K2,N,packed_out,mul", [(300, 64, 48, 96, False, False), (1000, 512, 512, 512, True, False), (777, 512, 0, 512, True, True),
                                                       (2049, 32, 512, 260, False, True), (4100, 512, 512, 1536, False, False), (129, 16, 16, 64, True, True)])
def test_chained_products_two_a_segments_and_packed_output(env, M, K1, K2, N, packed_out, mul):
    """gvqa_linear_split2h_chain (round 5; LCGN's node products, /root/reference baseline_and_test_models/lcgn.py:312-319): the A operand as
    TWO separately packed K segments with their own row scales -- deliberately 2^9 apart here, so that the exact power-of-two rescale of
    the accumulators at the switch is exercised in both directions --, bias / addend / elementwise mul / ReLU, and the finished rows
    leaving as the next product's packed operand.  Against fp64: the fp32 result within the two-piece bound of a single-segment product,
    the packed result piece-for-piece what gvqa_split2h_pack makes of the fp32 result."""
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(M + K1 + N)
    rn = lambda *s: torch.randn(*s, generator=g)
    U = (rn(M, K1) * torch.exp(2 * rn(M, 1))).to(dev)
    V = (rn(M, K2) * torch.exp(2 * rn(M, 1)) * (2.0 ** 9 if M % 2 else 2.0 ** -9)).to(dev) if K2 else None
    K2p = -(-K2 // 16) * 16
    W = (rn(N, K1 + K2p) / (K1 + K2) ** 0.5).to(dev)
    if K2 and K2p != K2:
        W[:, K1 + K2:] = 0
    bias = rn(N).to(dev)
    add = rn(M, N).to(dev)
    mulv = (rn(M, N).to(dev) if mul else None)
    upk = _pack(env, U, "split2h")
    vpk = _pack(env, V, "split2h") if K2 else None
    wpk = _pack(env, W, "split2h")
    C = torch.empty(M, N, device=dev)
    pk = torch.empty(lib.gvqa_split2h_packed_bytes(M, N), dtype=torch.uint8, device=dev) if packed_out else None
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.gvqa_linear_split2h_chain(M, N, K1, upk.data_ptr(), K2, vpk.data_ptr() if K2 else None, wpk.data_ptr(), bias.data_ptr(),
                                             add.data_ptr(), N, mulv.data_ptr() if mul else None, N if mul else 0, 1, C.data_ptr(), N,
                                             pk.data_ptr() if packed_out else None, st))
    A = torch.cat([U, V], 1) if K2 else U
    ref = A.double() @ W[:, :A.shape[1]].double().t() + bias.double() + add.double()
    if mul:
        ref = ref * mulv.double()
    ref = torch.relu(ref)
    scale = float((A.double().abs() @ W[:, :A.shape[1]].double().abs().t()).max())      # the products' own magnitude (before cancellation)
    err = float((C.double() - ref).abs().max())
    assert err < 2e-6 * scale + 1e-6, (err, scale)
    if packed_out:
        again = _pack(env, C, "split2h")                   # what the pack pass would have made of the fp32 rows
        assert torch.equal(pk, again)
