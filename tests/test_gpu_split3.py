"""split3: the fp32-accurate projection on the bf16 matrix cores (csrc/split3.hip) against fp64 and against
the f32-input MFMA kernel, through the C ABI.  The hop projection it serves is /root/reference gat_skip.py:133."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# kernel variants (tile geometry x schedule x epilogue, see launch_linear_split3); 0 = the library's own choice
VARIANTS = [0, 10, 11, 14, 21, 28, 29, 30, 34]


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from graphvqa_amd import _lib
    return _lib, _lib.load(), torch.device("cuda:0")


def _pack(env, X):
    _lib, lib, dev = env
    rows, K = X.shape
    buf = torch.empty(lib.gvqa_split3_packed_bytes(rows, K), dtype=torch.uint8, device=dev)
    _lib.check(lib.gvqa_split3_pack(rows, K, X.data_ptr(), X.stride(0), buf.data_ptr(), torch.cuda.current_stream().cuda_stream))
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


def _run(env, A, W, tile, bias=None, addend=None, mul=None, relu=0):
    _lib, lib, dev = env
    M, K = A.shape
    N = W.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    a, w = _pack(env, A), _pack(env, W)
    C = torch.full((M, N), float("nan"), device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None
    old = _lib.set_option(_lib.OPT_SPLIT3_VARIANT, tile)
    try:
        _lib.check(lib.gvqa_linear_split3(M, N, K, a.data_ptr(), w.data_ptr(), ptr(bias), ptr(addend),
                                          addend.stride(0) if addend is not None else 0, ptr(mul),
                                          mul.stride(0) if mul is not None else 0, relu, C.data_ptr(), C.stride(0), st))
    finally:
        _lib.set_option(_lib.OPT_SPLIT3_VARIANT, old)
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("tile", VARIANTS)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 132, 40), (1000, 1200, 300), (33, 4, 16), (513, 520, 512)])
def test_linear_split3_matches_fp64_like_fp32(env, tile, M, N, K):
    """Error against fp64 must be in the class of an exact-fp32 k-ordered fmaf chain (the f32 MFMA kernel)."""
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(M + 7 * N + 13 * K)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    C = _run(env, A, W, tile)
    ref = A.double() @ W.double().t()
    err = float((C.double() - ref).abs().max())
    C32 = torch.empty(M, N, device=dev)
    _lib.check(lib.gvqa_linear_f32(M, N, K, A.data_ptr(), K, W.data_ptr(), K, None, 0, C32.data_ptr(), N,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    err32 = float((C32.double() - ref).abs().max())
    assert torch.isfinite(C).all()
    assert err <= max(2.0 * err32, 2e-6), (err, err32)


@pytest.mark.parametrize("tile", VARIANTS)
def test_linear_split3_epilogues(env, tile):
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
        C = _run(env, A, W, tile, **kw)
        assert float((C.double() - ref).abs().max()) < 2e-5, kw.keys()


def test_linear_split3_wide_dynamic_range(env):
    """Pieces keep fp32 accuracy when operand magnitudes span many binades (where a bf16 or 2-piece product would not)."""
    _lib, lib, dev = env
    g = torch.Generator(device="cpu").manual_seed(11)
    M, N, K = 384, 256, 512
    A = (torch.randn(M, K, generator=g) * torch.exp(3 * torch.randn(M, K, generator=g))).to(dev)
    W = (torch.randn(N, K, generator=g) * torch.exp(3 * torch.randn(N, K, generator=g))).to(dev)
    C = _run(env, A, W, 0)
    C32 = torch.empty(M, N, device=dev)
    _lib.check(lib.gvqa_linear_f32(M, N, K, A.data_ptr(), K, W.data_ptr(), K, None, 0, C32.data_ptr(), N,
                                   torch.cuda.current_stream().cuda_stream))
    ref = A.double() @ W.double().t()
    scale = (A.double().abs() @ W.double().abs().t())                       # sum |a b|: the natural error scale
    rel = float(((C.double() - ref).abs() / scale).max())
    rel32 = float(((C32.double() - ref).abs() / scale).max())
    assert rel < max(2 * rel32, 2e-7), (rel, rel32)                       # the f32-MFMA fmaf chain's own roundoff class


def test_linear_split3_rejects_unaligned(env):
    _lib, lib, dev = env
    A, W = torch.randn(32, 16, device=dev), torch.randn(6, 16, device=dev)
    a, w = _pack(env, A), _pack(env, W)
    C = torch.empty(32, 6, device=dev)
    rc = lib.gvqa_linear_split3(32, 6, 16, a.data_ptr(), w.data_ptr(), None, None, 0, None, 0, 0, C.data_ptr(), 6,
                                torch.cuda.current_stream().cuda_stream)
    assert rc == _lib.E_UNSUPPORTED
