"""MI355X-native drop-in for the reference's `gat_skip` module (gat_skip.py).

Same class names, constructor arguments, forward signatures and state_dict keys/shapes as the
reference (`gat`: gat_skip.py:60-63,111-112; `gat_seq`: gat_skip.py:224-225,249), so that
`from gat_skip import gat_seq` in pipeline_model_gat.py:16 can be pointed here unchanged
 and checkpoints load through the pipeline's tolerant loader
(pipeline_model_gat.py:823-836).  The compute runs in hand-written HIP kernels behind the
C ABI of include/gvqa.h; torch only provides device memory and the current stream.

Scope: `.eval()` under `torch.no_grad()` (running-statistics BatchNorm, dropout inactive) is the measured,
fully fused path.  When gradients are needed (or in `.train()` with dropout p > 0) `gat_seq.forward` runs
the differentiable formulation: the message passing -- gather, segment softmax, weighted scatter-add, head
mean -- and its backward are the HIP kernels (`gvqa_gat_message_passing` / `gvqa_gat_mp_backward` behind
`gat_message_passing`, a `torch.autograd.Function`), the dense projections, BatchNorm and dropout are torch
ops on the device, so autograd produces the gradients of every parameter and input.
`.train()` with p == 0 and no gradient needed keeps the fused batch-statistics forward.  There is no CPU
path: CPU tensors raise, and a missing HIP library raises at construction.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Optional

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable
from torch.nn import Parameter, Linear

from . import _lib
from .graph import SceneGraphBatch, _stream, _ptr


# A/B switch of the training path's round-5 fusions (scripts/ab_train_parts.sh; measurement only, 0 = everything on): bit 1 no |h| maxima from
# the operand pack, 2 the skip's gradient through autograd, 4 head rows / bias / skip as a second pass, 8 tiny per-graph products on the tiled kernel;
# bit 64: the node logits as a pass of their own instead of a by-product of the operand pack; bit 32: feature-dropout masks as torch bernoulli_ tensors instead of in-kernel Philox draws;
# bit 16 switches ON the (slower, kept for the record) form that adds the logit products' input gradient in the dx product's epilogue
_TRAIN_AB = int(os.environ.get("GVQA_TRAIN_AB", "0") or 0)


def _glorot(t: Tensor):
    """PyG inits.glorot (call sites gat_skip.py:101-107)."""
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


def _f32c(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the MI355X execution path has no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def _inference_only(module: torch.nn.Module, *tensors) -> None:
    """Modules whose backward is not built: refuse to run where autograd would expect a differentiable result
    (a silently detached output would train nothing)."""
    if torch.is_grad_enabled() and (any(isinstance(t, Tensor) and t.requires_grad for t in tensors) or
                                    any(p.requires_grad for p in module.parameters())):
        raise NotImplementedError(
            f"{type(module).__name__} on the HIP path is inference-only (backward not built, SURVEY 8f-4): call it under "
            "torch.no_grad() / with requires_grad_(False) parameters")


def _workspace(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _lib_abt(a: Tensor, b: Tensor, bias: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """C = a b^T (+ bias) on the library's f32-input MFMA kernel (gvqa_linear_f32): a [M, K], b [N, K], both with unit inner stride
    (row strides are passed on); `out`: an [M, N] view with unit inner stride (e.g. a column block of a wider tensor).  The
    products of the differentiable path that are too small or the wrong shape for the split kernels -- per-graph [B, Di]-sized
    ones, products under the size threshold -- run here instead of on torch's vendor GEMM."""
    lib = _lib.load()
    M, K = a.shape
    N = b.shape[0]
    if not a.is_cuda:
        raise RuntimeError("graphvqa_amd runs on an MI355X only (no CPU fallback): got a tensor on %s" % (a.device,))
    if a.dtype != torch.float32 or b.dtype != torch.float32:       # (fp64 gradcheck, autocast halves: name the dtype -- ADVICE r04)
        raise TypeError("graphvqa_amd's products take float32 operands (got %s x %s): cast the module and its inputs to float32" % (a.dtype, b.dtype))
    if a.stride(1) != 1 or (M > 1 and a.stride(0) < K):
        a = a.contiguous()
    if b.stride(1) != 1 or (N > 1 and b.stride(0) < K):
        b = b.contiguous()
    if bias is not None:
        bias = bias.contiguous()
    res = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    if M == 0 or N == 0:
        return res
    if K == 0:
        res.zero_() if bias is None else res.copy_(bias.expand(M, N))
        return res
    assert res.stride(1) == 1
    with torch.cuda.device(a.device):
        _lib.check(lib.gvqa_linear_f32(M, N, K, a.data_ptr(), max(a.stride(0), K), b.data_ptr(), max(b.stride(0), K), _ptr(bias), 0,
                                       res.data_ptr(), max(res.stride(0), N), _stream(a.device)))
    return res


class _ProjectionLinear(torch.autograd.Function):
    """y = x W^T for the hop projection of the differentiable path (gat_skip.py:133): the forward product runs on the library's
    own GEMMs -- the arithmetic GVQA_OPT_PROJECTION selects (two-piece fp16 / three-piece bf16 split on the 16-bit matrix cores,
    or the f32-input MFMA kernel), exactly as in the eval path -- and so do dx = dy W and dW = dy^T x (a [HC, Dn] result reduced
    over all N rows: the library's transposed-pack split-K product, gvqa_linear_tn_split2h) in the backward.
    `w` may be a column slice of a wider weight (its row stride is passed on).  `bias` (optional, [N]) is added in the product's
    epilogue (no pass over y for it); its gradient is the column sum of dy."""

    @staticmethod
    def forward(ctx, x, w, bias=None):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return _ProjectionLinear._product(x, w, bias)

    @staticmethod
    def _product(x, w, bias=None, absmax_out=None, logits=None):
        """absmax_out ([_lib.ABSMAX_SLOTS] fp32, optional): filled with slice maxima of |x| when the two-piece path packs x (a by-product of
        its row scales) and then tagged `_gvqa_filled`; the backward's weight-gradient product takes it instead of a pass over x."""
        lib = _lib.load()
        M, K = x.shape
        N = w.shape[0]
        mode = lib.gvqa_get_option(_lib.OPT_PROJECTION)
        ok = (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.is_contiguous() and w.stride(1) == 1 and
              N % 4 == 0 and M > 0 and 2.0 * M * N * K >= 1e6 * lib.gvqa_get_option(_lib.OPT_SPLIT3_MIN_MFLOP))
        if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.data_ptr() % 16 != 0):
            ok = False
        if not ok:        # small / odd shapes: the library's f32-input MFMA kernel (no vendor GEMM on this path)
            return _lib_abt(x, w, bias)
        dev, st = x.device, _stream(x.device)
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if mode == _lib.PROJECTION_F32:
                _lib.check(lib.gvqa_linear_f32(M, N, K, x.data_ptr(), K, w.data_ptr(), w.stride(0), _ptr(bias), 0, out.data_ptr(), N, st))
                return out
            nbytes, pack, linear = ((lib.gvqa_split2h_packed_bytes, lib.gvqa_split2h_pack, lib.gvqa_linear_split2h)
                                    if mode == _lib.PROJECTION_SPLIT2H else
                                    (lib.gvqa_split3_packed_bytes, lib.gvqa_split3_pack, lib.gvqa_linear_split3))
            apk, wpk = _workspace(nbytes(M, K), dev), _workspace(nbytes(N, K), dev)
            if logits is not None and mode == _lib.PROJECTION_SPLIT2H and logits[0].shape == (8, K) and K % 4 == 0 and K <= 1024:
                # `logits` = (Vn [8, K], out [M, 8]): x Vn^T leaves the pack pass as well (tagged `_gvqa_filled` on the output)
                _lib.check(lib.gvqa_split2h_pack_logits(M, K, x.data_ptr(), K, apk.data_ptr(), _ptr(absmax_out), logits[0].data_ptr(), 8,
                                                        logits[1].data_ptr(), st))
                logits[1]._gvqa_filled = True
                if absmax_out is not None:
                    absmax_out._gvqa_filled = True
            elif absmax_out is not None and mode == _lib.PROJECTION_SPLIT2H:
                _lib.check(lib.gvqa_split2h_pack_absmax(M, K, x.data_ptr(), K, apk.data_ptr(), absmax_out.data_ptr(), st))
                absmax_out._gvqa_filled = True
            else:
                _lib.check(pack(M, K, x.data_ptr(), K, apk.data_ptr(), st))
            _lib.check(pack(N, K, w.data_ptr(), w.stride(0), wpk.data_ptr(), st))
            _lib.check(linear(M, N, K, apk.data_ptr(), wpk.data_ptr(), _ptr(bias), None, 0, None, 0, 0, out.data_ptr(), N, st))
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gb = gy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        fused = _ProjectionLinear._backward_fused(gy, x, w, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        if fused is not None:
            return fused[0], fused[1], gb
        gx = None
        if ctx.needs_input_grad[0]:
            # dx = dy W = dy (W^T)^T: the same kernels with the (small) weight transposed
            gx = _ProjectionLinear._product(gy.contiguous(), w.t().contiguous())
        gw = _ProjectionLinear._weight_grad(gy, x) if ctx.needs_input_grad[1] else None
        return gx, gw, gb

    @staticmethod
    def _backward_fused(gy, x, w, want_x, want_w, gx_init=None, gw_out=None, x_absmax=None, lowrank=None, addend=None):
        """dx = dy W and dW = dy^T x in one library call under the two-piece arithmetic (gvqa_linear_backward_split2h: dy is read and
        packed once for both products); None when the shapes / settings are not the ones it takes."""
        lib = _lib.load()
        R, M = gy.shape
        K = x.shape[1]
        ok = (gy.is_cuda and gy.dtype == torch.float32 and x.dtype == torch.float32 and w.dtype == torch.float32 and M % 4 == 0 and
              K % 4 == 0 and R > 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and w.stride(1) == 1 and w.stride(0) % 4 == 0 and
              lib.gvqa_get_option(_lib.OPT_PROJECTION) == _lib.PROJECTION_SPLIT2H and
              2.0 * R * M * K >= 1e6 * lib.gvqa_get_option(_lib.OPT_SPLIT3_MIN_MFLOP))
        if not ok or not (want_x or want_w) or (gw_out is not None and (gw_out.stride(1) != 1 or gw_out.stride(0) % 4 != 0)):
            return None
        am = _absmax_hint(gy)
        gy = gy.contiguous()
        dev = gy.device
        gx = (gx_init if gx_init is not None else torch.empty((R, K), dtype=torch.float32, device=dev)) if want_x else None
        gw = (gw_out if gw_out is not None else torch.empty((M, K), dtype=torch.float32, device=dev)) if want_w else None      # (gw_out: a column block of a wider gradient, row stride = its width)
        with torch.cuda.device(dev):
            ws = _workspace(lib.gvqa_linear_backward_workspace_bytes(R, M, K), dev)
            ex = _lib.LinearBackwardExtras()
            ex.x_absmax, ex.x_absmax_n = _ptr(x_absmax), 0 if x_absmax is None else x_absmax.numel()
            if want_x and lowrank is not None:        # dx += g v^T in the product's epilogue (`lowrank` = (g [R, J], v [K, J]))
                ex.lowrank_g, ex.lowrank_v, ex.J = lowrank[0].data_ptr(), lowrank[1].data_ptr(), lowrank[0].shape[1]
            if want_x and addend is not None:
                ex.addend, ex.ld_addend = addend.data_ptr(), addend.stride(0)
            rc = lib.gvqa_linear_backward_split2h_ex(R, M, K, gy.data_ptr(), M, w.data_ptr(), w.stride(0), x.data_ptr(), x.stride(0),
                                                     _ptr(am), 0 if am is None else am.numel(), _ptr(gx), K,
                                                     int(gx_init is not None and want_x), _ptr(gw), K if gw is None else gw.stride(0),
                                                     C.byref(ex), ws.data_ptr(), ws.numel(), _stream(dev))
            if rc == _lib.E_UNSUPPORTED and (lowrank is not None or addend is not None):
                return None                            # (the epilogue terms need the direct product: the caller takes the separate passes)
            _lib.check(rc)
        return gx, gw

    @staticmethod
    def _weight_grad(gy, x):
        """dW = dy^T x, a reduction over all rows: the library's split-K product on the fp16 matrix cores (gvqa_linear_tn_split2h)
        under the two-piece arithmetic, torch's fp32 matmul otherwise (small products, other GVQA_OPT_PROJECTION settings)."""
        lib = _lib.load()
        R, M = gy.shape
        N = x.shape[1]
        ok = (gy.is_cuda and gy.dtype == torch.float32 and x.dtype == torch.float32 and M % 4 == 0 and N % 4 == 0 and R > 0 and
              x.stride(1) == 1 and x.stride(0) % 4 == 0 and lib.gvqa_get_option(_lib.OPT_PROJECTION) == _lib.PROJECTION_SPLIT2H and
              2.0 * R * M * N >= 1e6 * lib.gvqa_get_option(_lib.OPT_SPLIT3_MIN_MFLOP))
        if not ok:
            return _lib_abt(gy.t().contiguous(), x.t().contiguous())
        # the producer of dy may have left its largest magnitudes with the tensor (the message-passing backward does): no pass over dy
        am = _absmax_hint(gy)
        gy = gy.contiguous()
        dev = gy.device
        gw = torch.empty((M, N), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = _workspace(lib.gvqa_linear_tn_workspace_bytes(R, M, N), dev)
            _lib.check(lib.gvqa_linear_tn_split2h(R, M, N, gy.data_ptr(), M, x.data_ptr(), x.stride(0), _ptr(am), 0 if am is None else am.numel(),
                                                  None, 0, gw.data_ptr(), N, ws.data_ptr(), ws.numel(), _stream(dev)))
        return gw


def _absmax_hint(gy: Tensor):
    """The per-slice largest magnitudes a producer left with its gradient tensor (`_gvqa_absmax`: the message-passing backward
    does, so that the split products need no pass over dy) -- honoured only if the tensor is still the one, and in the state,
    the producer described: same storage address, same in-place version counter, contiguous.  gvqa_linear_tn_split2h's contract
    is `hint >= max|dy|`; a stale hint would mean fp16 overflow in dW."""
    hint = getattr(gy, "_gvqa_absmax", None)
    if hint is None or not gy.is_contiguous():
        return None
    am, ptr, version = hint
    return am if (gy.data_ptr() == ptr and gy._version == version) else None


def _attention_dropout_mask(E: int, H: int, p: float, device) -> Tensor:
    """mask / (1 - p) of F.dropout(alpha) (gat_skip.py:205) as one library launch keyed on torch's CUDA generator (counters reserved there), or
    torch's own three kernels where that does not apply."""
    n = E * H
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()] if device.type == "cuda" else None
    if gen is None or n % 4 != 0 or n == 0 or not hasattr(gen, "get_offset") or (_TRAIN_AB & 32):
        return torch.bernoulli(torch.full((E, H), 1.0 - p, device=device)) / (1.0 - p)
    off = gen.get_offset()
    gen.set_offset(off + 4 * ((n // 4 + 3) // 4))
    mask = torch.empty((E, H), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.load().gvqa_dropout_scale_mask(n, gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, off, float(p), mask.data_ptr(), _stream(device)))
    return mask


class _ColumnBlocks(torch.autograd.Function):
    """[R, K*W] -> K column blocks [R, W] as views (no copies); the backward writes the K gradients into ONE [R, K*W] tensor -- K strided copies
    instead of autograd's K zero-filled full-size tensors, K slice copies and K - 1 full-size adds."""

    @staticmethod
    def forward(ctx, a, K, W):
        ctx.dims = (a.shape, K, W)
        return tuple(a[:, i * W:(i + 1) * W] for i in range(K))

    @staticmethod
    def backward(ctx, *grads):
        shape, K, W = ctx.dims
        ref = next(g for g in grads if g is not None)
        out = torch.empty(shape, dtype=ref.dtype, device=ref.device)
        for i, g in enumerate(grads):
            if g is None:
                out[:, i * W:(i + 1) * W].zero_()
            else:
                out[:, i * W:(i + 1) * W].copy_(g)
        return out, None, None


_ONES = {}


def _ones_column(n: int, device) -> Tensor:
    key = (n, str(device))
    t = _ONES.get(key)
    if t is None:
        if len(_ONES) > 16:
            _ONES.clear()
        t = _ONES[key] = torch.ones((n, 1), dtype=torch.float32, device=device)
    return t


class _NoCtx:
    """Stand-in for an autograd context when a Function's forward is used as a plain kernel call inside another node."""
    def save_for_backward(self, *a):
        pass


class _SkinnyLinear(torch.autograd.Function):
    """y = x V for a tall x [R, D] and a narrow V [D, J] (J <= 32): the attention logits of the differentiable path,
    (x_i * att).sum(-1) of gat_skip.py:134-135,151 with att folded through the projection weights.  Forward, dV = x^T dy and
    dx = dy V^T are the library's kernels (csrc/train.hip): each streams x (or dx) through HBM once."""

    @staticmethod
    def supported(x, V):
        return (x.is_cuda and x.dtype == torch.float32 and V.dtype == torch.float32 and x.dim() == 2 and V.dim() == 2 and
                x.shape[1] % 4 == 0 and x.shape[1] <= 1024 and 1 <= V.shape[1] <= 32 and x.stride(1) == 1 and x.stride(0) % 4 == 0)

    @staticmethod
    def forward(ctx, x, V):
        lib = _lib.load()
        V = V.contiguous()
        R, D = x.shape
        J = V.shape[1]
        y = torch.empty((R, J), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.gvqa_skinny_forward(R, D, J, x.data_ptr(), x.stride(0), V.data_ptr(), y.data_ptr(), _stream(x.device)))
        ctx.save_for_backward(x, V)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, V = ctx.saved_tensors
        R, D = x.shape
        J = V.shape[1]
        gy = gy.contiguous()
        gx = gV = None
        with torch.cuda.device(x.device):
            st = _stream(x.device)
            if ctx.needs_input_grad[1]:
                gV = torch.empty_like(V)
                ws = _workspace(lib.gvqa_skinny_backward_weight_workspace_bytes(R, D, J), x.device)
                _lib.check(lib.gvqa_skinny_backward_weight(R, D, J, x.data_ptr(), x.stride(0), gy.data_ptr(), gV.data_ptr(),
                                                           ws.data_ptr(), ws.numel(), st))
            if ctx.needs_input_grad[0]:
                gx = torch.empty((R, D), dtype=torch.float32, device=x.device)
                _lib.check(lib.gvqa_skinny_backward_input(R, D, J, gy.data_ptr(), V.data_ptr(), None, 0, gx.data_ptr(), D, st))
        return gx, gV


class _LibMatmul(torch.autograd.Function):
    """y = x V on the library's f32-input MFMA kernel, differentiable (dx = dy V^T, dV = x^T dy): the shapes the tall-skinny kernels do
    not take (widths that are not a multiple of 4, ...)."""

    @staticmethod
    def forward(ctx, x, V):
        ctx.save_for_backward(x, V)
        return _lib_abt(x, V.t().contiguous())

    @staticmethod
    def backward(ctx, gy):
        x, V = ctx.saved_tensors
        gy = gy.contiguous()
        gx = _lib_abt(gy, V) if ctx.needs_input_grad[0] else None
        gV = _lib_abt(x.t().contiguous(), gy.t().contiguous()) if ctx.needs_input_grad[1] else None
        return gx, gV


def skinny_linear(x: Tensor, V: Tensor) -> Tensor:
    """x @ V through the library's tall-skinny kernels (column groups of 32 when V is wider); differentiable."""
    if not _SkinnyLinear.supported(x, V[:, :1]):
        return _LibMatmul.apply(x, V) if (x.is_cuda and x.dtype == torch.float32 and V.dtype == torch.float32) else x @ V
    J = V.shape[1]
    if J <= 32:
        return _SkinnyLinear.apply(x, V)
    return torch.cat([_SkinnyLinear.apply(x, V[:, j:j + 32]) for j in range(0, J, 32)], dim=1)


class _FoldAttention(torch.autograd.Function):
    """V [Kin, J]: the attention vectors folded through a projection weight, so that (x W^T * att).sum(-1) = x V
    (gat_skip.py:134-135,151): V[k, h] = sum_c W[h C + c, k] att_a[h, c] and, with att_b, V[k, H + h] likewise (J = 2H).
    Forward and adjoint are the library's kernels (gvqa_fold_attention_*), one launch each over W."""

    @staticmethod
    def forward(ctx, W, att_a, att_b, heads):
        lib = _lib.load()
        W = _f32c(W, "weight")
        HC, Kin = W.shape
        Cc = HC // heads
        a = _f32c(att_a.reshape(-1), "att")
        b = None if att_b is None else _f32c(att_b.reshape(-1), "att")
        V = torch.empty((Kin, heads * (1 if b is None else 2)), dtype=torch.float32, device=W.device)
        with torch.cuda.device(W.device):
            _lib.check(lib.gvqa_fold_attention_forward(heads, Cc, Kin, W.data_ptr(), Kin, a.data_ptr(), _ptr(b), V.data_ptr(), _stream(W.device)))
        ctx.save_for_backward(W, a, b)
        ctx.heads, ctx.shapes = heads, (att_a.shape, None if att_b is None else att_b.shape)
        return V

    @staticmethod
    def backward(ctx, dV):
        lib = _lib.load()
        W, a, b = ctx.saved_tensors
        heads = ctx.heads
        HC, Kin = W.shape
        dV = dV.contiguous()
        dW = torch.empty_like(W) if ctx.needs_input_grad[0] else None
        want_att = ctx.needs_input_grad[1] or (b is not None and ctx.needs_input_grad[2])
        da = torch.empty_like(a) if want_att else None
        db = torch.empty_like(b) if (want_att and b is not None) else None
        with torch.cuda.device(W.device):
            _lib.check(lib.gvqa_fold_attention_backward(heads, HC // heads, Kin, W.data_ptr(), Kin, a.data_ptr(), _ptr(b), dV.data_ptr(),
                                                        _ptr(dW), Kin, _ptr(da), _ptr(db), _stream(W.device)))
        sa, sb = ctx.shapes
        return dW, (None if da is None else da.view(sa)), (None if db is None else db.view(sb)), None


def fold_attention(W: Tensor, att_a: Tensor, att_b: Optional[Tensor], heads: int) -> Tensor:
    """[Kin, H] (or [Kin, 2H] with att_b): attention vectors folded through the projection weight W [H*C, Kin]; differentiable."""
    return _FoldAttention.apply(W, att_a, att_b, heads)


class _HopProducts(torch.autograd.Function):
    """Everything a hop multiplies by its lin_l weight, as ONE autograd node (gat_skip.py:133-135 on x_cat = [h | ins[batch]], :263-264):
        xp      = h W[:, :Dn]^T            [N, H*C]   node half of the projection (library GEMM)
        a_part  = h F[:Dn]                 [N, 2H]    node half of the folded logits (F = att_l | att_r through W)
        xp_rows = ins W[:, Dn:]^T          [B, H*C]   instruction half, one row per graph
        a_rows  = ins [F[Dn:, :H] + U_e | F[Dn:, H:]]   [B, 2H]   (U_e: the edge's instruction term, :257-260, on the source half)
    The backward writes ONE full-size dW (left columns from the split-K product, right columns from the per-graph rows) and one
    full-size dF -- no column slices of W or F in the autograd graph, hence no zero-padded slice gradients to fill and add."""

    @staticmethod
    def forward(ctx, h, ins, W, F_, U_e, Dn, skip_grad=None):
        # skip_grad: a list shared with the hop's message-passing node (gat_seq._forward_autograd): that node's backward leaves the gradient
        # of its `skip` operand -- the same h -- there instead of returning it, and the backward below adds it inside the kernel that
        # writes dh (no [N, D] add by autograd between the two nodes)
        ctx.skip_grad = skip_grad
        heads2 = F_.shape[1]
        H = heads2 // 2
        with torch.no_grad():
            # (the largest magnitudes of h leave the operand pack as a by-product: the backward's dW = dxp^T h needs h's ONE scale)
            am_h = torch.empty(_lib.ABSMAX_SLOTS, dtype=torch.float32, device=h.device) if (h.is_cuda and W.requires_grad and not (_TRAIN_AB & 1)) else None
            # ... and the node halves of the folded logits, a_part = h F[:Dn], come out of the same pass over h (2H = 8)
            lg = None
            if heads2 == 8 and h.is_cuda and h.is_contiguous() and not (_TRAIN_AB & 64):
                lg = (F_[:Dn].t().contiguous(), torch.empty((h.shape[0], 8), dtype=torch.float32, device=h.device))
            xp = _ProjectionLinear._product(h, W[:, :Dn], absmax_out=am_h, logits=lg)
            ctx.h_absmax = (am_h, h._version) if (am_h is not None and getattr(am_h, "_gvqa_filled", False)) else None
            a_part = lg[1] if (lg is not None and getattr(lg[1], "_gvqa_filled", False)) else skinny_linear(h, F_[:Dn])
            xp_rows = _lib_abt(ins, W[:, Dn:])
            U_n = F_[Dn:].clone()
            U_n[:, :H] += U_e
            # [B, Di] x [Di, 2H]: the tall-skinny kernel (a K = Di loop on 16 workgroups of the tiled f32 kernel is latency, 38 us)
            a_rows = (_SkinnyLinear.forward(_NoCtx(), ins, U_n) if _SkinnyLinear.supported(ins, U_n) and ins.is_contiguous() and not (_TRAIN_AB & 8)
                      else _lib_abt(ins, U_n.t().contiguous()))
        ctx.save_for_backward(h, ins, W, F_, U_n)
        ctx.Dn = Dn
        return xp, a_part, xp_rows, a_rows

    @staticmethod
    @once_differentiable        # (the skip gradient travels in `skip_grad`, outside autograd: a double backward would silently lose it -- ADVICE r05)
    def backward(ctx, gxp, ga, g_rows, g_arows):
        h, ins, W, F_, U_n = ctx.saved_tensors
        Dn = ctx.Dn
        lib = _lib.load()
        want_h, want_ins, want_w, want_f, want_ue = ctx.needs_input_grad[:5]
        R, D = h.shape
        H2 = F_.shape[1]
        H = H2 // 2
        dev = h.device
        V = F_[:Dn]                                           # contiguous row block
        ga = ga.contiguous()
        gh = gF = gW = gins = gUe = None
        ok = _SkinnyLinear.supported(h, V) and H2 in (2, 4, 8, 16)
        if want_f or want_ue:
            gF = torch.empty_like(F_)
            if ok:
                with torch.cuda.device(dev):
                    ws = _workspace(lib.gvqa_skinny_backward_weight_workspace_bytes(R, D, H2), dev)
                    _lib.check(lib.gvqa_skinny_backward_weight(R, D, H2, h.data_ptr(), h.stride(0), ga.data_ptr(), gF.data_ptr(), ws.data_ptr(),
                                                               ws.numel(), _stream(dev)))
            else:
                _lib_abt(h.t().contiguous(), ga.t().contiguous(), out=gF[:Dn])
            g_arows = g_arows.contiguous()
            if _SkinnyLinear.supported(ins, U_n) and ins.is_contiguous() and H2 in (2, 4, 8, 16) and not (_TRAIN_AB & 8):
                with torch.cuda.device(dev):                  # d a_rows / d U_n = ins^T g_arows: rows [Dn:] of dF ... (tall-skinny dV kernel)
                    ws = _workspace(lib.gvqa_skinny_backward_weight_workspace_bytes(ins.shape[0], ins.shape[1], H2), dev)
                    _lib.check(lib.gvqa_skinny_backward_weight(ins.shape[0], ins.shape[1], H2, ins.data_ptr(), ins.stride(0), g_arows.data_ptr(),
                                                               gF[Dn:].data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
            else:
                ins_t = ins.t().contiguous()                  # [Di, B]: the contraction over graphs as the unit-stride dimension
                _lib_abt(ins_t, g_arows.t().contiguous(), out=gF[Dn:])
            if want_ue:
                gUe = gF[Dn:, :H].clone()                     # ... whose source half is dU_e as well
        extra = ctx.skip_grad.pop() if ctx.skip_grad else None          # d loss / d (the hop's skip operand), left by the message-passing node
        if want_w:
            gW = torch.empty_like(W)
            # instruction half: dW_i = d xp_rows^T ins, a reduction over the graphs: the direct transposing product (no [B, H*C] / [B, Di]
            # transposes through torch), or the tiled f32 kernel on transposed copies where that does not apply
            g_rows = g_rows.contiguous()
            gWi = gW[:, Dn:]
            Bg, HC, Di_ = g_rows.shape[0], g_rows.shape[1], ins.shape[1]
            if (lib.gvqa_get_option(_lib.OPT_PROJECTION) == _lib.PROJECTION_SPLIT2H and lib.gvqa_get_option(_lib.OPT_TN_DIRECT) and ins.is_contiguous()
                    and HC % 4 == 0 and Di_ % 4 == 0 and gWi.stride(0) % 4 == 0 and gWi.data_ptr() % 16 == 0 and Bg >= 256 and not (_TRAIN_AB & 8)):
                with torch.cuda.device(dev):
                    wst = _workspace(lib.gvqa_linear_tn_workspace_bytes(Bg, HC, Di_), dev)
                    _lib.check(lib.gvqa_linear_tn_split2h(Bg, HC, Di_, g_rows.data_ptr(), HC, ins.data_ptr(), Di_, None, 0, None, 0, gWi.data_ptr(),
                                                          gWi.stride(0), wst.data_ptr(), wst.numel(), _stream(dev)))
            else:
                _lib_abt(g_rows.t().contiguous(), ins.t().contiguous(), out=gWi)
        Wh = W[:, :Dn]
        # (the hint holds while h is what the forward packed: an in-place change since then bumps its version counter)
        hint = ctx.h_absmax[0] if (ctx.h_absmax is not None and ctx.h_absmax[1] == h._version) else None
        extra_ok = extra is None or (extra.is_contiguous() and extra.shape == (R, D) and extra.dtype == torch.float32)
        fused = None
        if want_h and ok and extra_ok and H2 % 4 == 0 and H2 <= 16 and (_TRAIN_AB & 16):
            # dh = d xp W_h + d a V^T + d skip, written ONCE: the logit products' input gradient (a rank-2H term) and the skip's gradient ride in
            # the epilogue of the product that reads d xp directly (gvqa_linear_backward_split2h_ex).  OFF by default: measured 0.27 ms per
            # config-3 step SLOWER than the separate tall-skinny pass (13.16 vs 12.89 ms, same box) -- the epilogue's 8 FMAs per element run with
            # the matrix cores idle, the separate pass streams at the copy rate
            fused = _ProjectionLinear._backward_fused(gxp, h, Wh, True, want_w, gw_out=None if gW is None else gW[:, :Dn], x_absmax=hint,
                                                      lowrank=(ga, V), addend=extra)
            if fused is not None:
                gh = fused[0]
        if fused is None:
            if want_h:
                if ok and extra_ok:
                    gh = torch.empty((R, D), dtype=torch.float32, device=dev)
                    with torch.cuda.device(dev):
                        _lib.check(lib.gvqa_skinny_backward_input(R, D, H2, ga.data_ptr(), V.data_ptr(), _ptr(extra), D, gh.data_ptr(), D, _stream(dev)))
                else:
                    gh = _lib_abt(ga, V)
                    if extra is not None:
                        gh = gh + extra
            fused = _ProjectionLinear._backward_fused(gxp, h, Wh, want_h, want_w, gx_init=gh, gw_out=None if gW is None else gW[:, :Dn], x_absmax=hint)
            if fused is not None:
                gh = fused[0] if want_h else None
            else:
                if want_h:
                    gh = _ProjectionLinear._product(gxp.contiguous(), Wh.t().contiguous()) + gh
                if want_w:
                    gW[:, :Dn] = _ProjectionLinear._weight_grad(gxp, h)
        if want_ins:
            gins = _lib_abt(g_rows, W[:, Dn:].t().contiguous()) + _lib_abt(g_arows, U_n)
        return gh, gins, gW, (gF if want_f else None), gUe, None, None


class _GatMessagePassing(torch.autograd.Function):
    """out[i] = (1/H) sum_h sum_{e -> i} alpha[e,h] mask[e,h] xp[src_e, h, :],  alpha = softmax over the in-edges
    of leaky_relu(a_node[src,h] + a_node[dst,H+h] + a_edge[e,h])   (gat_skip.py:155,183-208,162-165).
    Forward and backward are HIP kernels; returns (out [N, C], alpha [E, H]).
    `graph_rows` [B, H*C] (optional): rows added to xp per GRAPH -- the instruction half of lin_l([h | ins[batch]]),
    gat_skip.py:133,263-264 -- i.e. the op computes MP(xp + graph_rows[batch]) without forming that [N, H*C] sum or its adjoint:
    out = MP(xp) + (1/H) sum_h s[i,h] graph_rows[g,h,:] with s[i,h] = sum_{e->i} alpha mask (gvqa_graph_head_rows_*).
    `bias` [C] and `skip` [N, C] (optional) are added in the same pass (gat_skip.py:167-168, :270)."""

    @staticmethod
    def forward(ctx, xp, a_node, a_edge, mask, graph, heads, channels, slope, graph_rows=None, bias=None, skip=None, skip_grad=None):
        lib = _lib.load()
        ctx.skip_grad = skip_grad          # (a list: the gradient of `skip` is left there for _HopProducts.backward instead of being returned)
        xp, a_node = _f32c(xp, "xp"), _f32c(a_node, "a_node")
        # a_edge may be a column block of a wider [E, K*H] tensor (the K hops' edge logits side by side): the kernels take its row stride
        if not (a_edge.is_cuda and a_edge.dtype == torch.float32 and a_edge.dim() == 2 and a_edge.stride(1) == 1 and a_edge.stride(0) >= a_edge.shape[1]):
            a_edge = _f32c(a_edge, "a_edge")
        if mask is not None:
            mask = _f32c(mask, "alpha_mask")
        if graph_rows is not None:
            graph_rows = _f32c(graph_rows, "graph_rows")
            if graph_rows.shape != (graph.num_graphs, heads * channels):
                raise ValueError("gat_message_passing: graph_rows must be [num_graphs, heads * channels]")
            if not graph.intra_graph:
                raise ValueError("gat_message_passing: graph_rows needs a batch without cross-graph edges (a message would carry the "
                                 "source graph's row); add the rows to xp instead")
        N, E, dev = graph.num_nodes, graph.num_edges, xp.device
        if xp.shape != (N, heads * channels) or a_node.shape != (N, 2 * heads) or a_edge.shape != (E, heads):
            raise ValueError("gat_message_passing: operand shapes do not match the graph")
        out = torch.empty((N, channels), dtype=torch.float32, device=dev)
        alpha = torch.empty((E, heads), dtype=torch.float32, device=dev)
        m = _lib.GatMpDesc()
        m.C, m.H, m.negative_slope, m.bn_eps = channels, heads, slope, 1e-5
        m.xp, m.a_node, m.a_edge = xp.data_ptr(), a_node.data_ptr(), a_edge.data_ptr()
        m.a_edge_stride = a_edge.stride(0)
        m.out, m.alpha_out, m.alpha_mask = out.data_ptr(), alpha.data_ptr(), _ptr(mask)
        s = None
        if bias is not None:
            bias = _f32c(bias, "bias")
        if skip is not None:
            skip = _f32c(skip, "skip")
            if skip.shape != (N, channels):
                raise ValueError("gat_message_passing: skip must be [N, channels]")
        with torch.cuda.device(dev):
            ws = _workspace(4 * E * heads, dev)
            fused = False
            if (graph_rows is not None or bias is not None or skip is not None) and not (_TRAIN_AB & 4):
                # one pass: the per-graph rows (weighted by the nodes' coefficient sums, which the kernel has in hand), bias and skip in the
                # message-passing kernel's own epilogue -- the LDS-tiled kernel's form; other batches take the two-pass form below
                m.bias, m.skip = _ptr(bias), _ptr(skip)
                if graph_rows is not None:
                    m.head_rows = graph_rows.data_ptr()
                    if mask is not None:
                        s = torch.empty((N, heads), dtype=torch.float32, device=dev)
                        m.head_weight_out = s.data_ptr()
                rc = lib.gvqa_gat_message_passing(C.byref(graph.c), C.byref(m), ws.data_ptr(), ws.numel(), _stream(dev))
                if rc == _lib.E_UNSUPPORTED:
                    m.bias = m.skip = m.head_rows = m.head_weight_out = None
                    s = None
                else:
                    _lib.check(rc)
                    fused = True
            if not fused:
                _lib.check(lib.gvqa_gat_message_passing(C.byref(graph.c), C.byref(m), ws.data_ptr(), ws.numel(), _stream(dev)))
                if graph_rows is not None or bias is not None or skip is not None:
                    if graph_rows is not None and mask is not None:   # s[i,h] = sum over the in-edges of alpha * mask (1 when nothing is dropped)
                        s = _edge_rows_sum_raw(alpha * mask, graph)
                    _lib.check(lib.gvqa_graph_head_rows_add(C.byref(graph.c), channels, heads, _ptr(graph_rows), _ptr(s), _ptr(bias), _ptr(skip),
                                                            channels, out.data_ptr(), channels, _stream(dev)))
        ctx.save_for_backward(xp, a_node, a_edge, alpha, mask, graph_rows, s)
        ctx.graph, ctx.dims = graph, (heads, channels, slope)
        ctx.has_bias, ctx.has_skip = bias is not None, skip is not None
        ctx.mark_non_differentiable(alpha)
        return out, alpha

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, _dalpha):
        lib = _lib.load()
        xp, a_node, a_edge, alpha, mask, graph_rows, s = ctx.saved_tensors
        heads, channels, slope = ctx.dims
        graph, dev = ctx.graph, xp.device
        dout = dout.contiguous()
        dxp, da_node = torch.empty_like(xp), torch.empty_like(a_node)
        da_edge = torch.empty(a_edge.shape, dtype=torch.float32, device=dev)
        d_rows = ds = dcol = None
        if graph_rows is not None or ctx.has_bias:
            if graph_rows is not None:
                d_rows = torch.empty_like(graph_rows)
                if mask is not None:          # without a mask s == 1 and the term is constant under the softmax
                    ds = torch.empty((graph.num_nodes, heads), dtype=torch.float32, device=dev)
            if ctx.has_bias:                  # per-graph column sums of dout; their sum is the bias gradient
                dcol = torch.empty((graph.num_graphs, channels), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.gvqa_graph_head_rows_backward(C.byref(graph.c), channels, heads, dout.data_ptr(), channels, _ptr(graph_rows),
                                                             _ptr(s), _ptr(d_rows), _ptr(ds), _ptr(dcol), _stream(dev)))
        d = _lib.GatMpBwdDesc()
        d.C, d.H, d.negative_slope = channels, heads, slope
        d.xp, d.a_node, d.a_edge = xp.data_ptr(), a_node.data_ptr(), a_edge.data_ptr()
        d.a_edge_stride = a_edge.stride(0)
        d.alpha, d.alpha_mask, d.dout = alpha.data_ptr(), _ptr(mask), dout.data_ptr()
        d.dxp, d.da_node, d.da_edge = dxp.data_ptr(), da_node.data_ptr(), da_edge.data_ptr()
        d.dalpha_node = _ptr(ds)
        am = torch.empty(_lib.ABSMAX_SLOTS, dtype=torch.float32, device=dev)      # largest |dxp| in slices, for the consumer of dxp
        d.dxp_absmax = am.data_ptr()
        with torch.cuda.device(dev):
            gt = graph.transposed()
            _lib.check(lib.gvqa_gat_mp_backward(C.byref(graph.c), C.byref(gt.c), C.byref(d), _stream(dev)))
        # the hint is valid for THIS tensor in THIS state only: autograd may accumulate a second consumer's gradient into dxp in
        # place, a hook may rescale it -- either bumps the version counter and the consumer then measures the maxima itself
        dxp._gvqa_absmax = (am, dxp.data_ptr(), dxp._version)
        d_bias = None
        if dcol is not None:                 # bias gradient = the sum of the per-graph column sums: a tall-skinny product with a column of ones
            Bg = dcol.shape[0]
            if channels % 4 == 0 and channels <= 1024:
                ones = _ones_column(Bg, dev)
                d_bias = torch.empty((channels, 1), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    wsb = _workspace(lib.gvqa_skinny_backward_weight_workspace_bytes(Bg, channels, 1), dev)
                    _lib.check(lib.gvqa_skinny_backward_weight(Bg, channels, 1, dcol.data_ptr(), channels, ones.data_ptr(), d_bias.data_ptr(),
                                                               wsb.data_ptr(), wsb.numel(), _stream(dev)))
                d_bias = d_bias.view(channels)
            else:
                d_bias = dcol.sum(0)
        if ctx.skip_grad is not None and ctx.has_skip:
            # internal hand-over to the same hop's _HopProducts.backward, which runs next and pops it.  A backward that stops short of that
            # node (torch.autograd.grad on a subset of inputs, retain_graph re-runs) must not pile tensors up here: at most ONE entry
            del ctx.skip_grad[:]
            ctx.skip_grad.append(dout)
            return dxp, da_node, da_edge, None, None, None, None, None, d_rows, d_bias, None, None
        return dxp, da_node, da_edge, None, None, None, None, None, d_rows, d_bias, (dout if ctx.has_skip else None), None


class _BatchNormReluTrain(torch.autograd.Function):
    """relu(BatchNorm1d(x)) with batch statistics (gat_skip.py:273-275 under model.train()), HIP forward and backward.
    Returns (y, batch mean, biased batch variance); the module updates the running statistics.
    `keep` [N, C] uint8 (optional) with `keep_scale` = 1 / (1 - p): feature dropout (gat_skip.py:276) applied in the same passes,
    y = relu(bn(x)) * (keep ? keep_scale : 0) -- the mask is drawn by the caller with torch's generator."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, keep=None, keep_scale=1.0, rng=None):
        """rng = (seed, offset, p): the keep decisions are drawn inside the kernels (gvqa_bn_relu_dropout_train_*_rng) -- no mask tensor."""
        lib = _lib.load()
        x, weight, bias = _f32c(x, "x"), _f32c(weight, "bn.weight"), _f32c(bias, "bn.bias")
        N, Cc = x.shape
        ctx.rng = rng
        if rng is not None:
            y, mean, var = torch.empty_like(x), torch.empty_like(weight), torch.empty_like(weight)
            with torch.cuda.device(x.device):
                ws = _workspace(lib.gvqa_bn_train_workspace_bytes(N, Cc), x.device)
                _lib.check(lib.gvqa_bn_relu_dropout_train_forward_rng(N, Cc, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), eps, rng[0], rng[1],
                                                                      float(rng[2]), y.data_ptr(), mean.data_ptr(), var.data_ptr(), ws.data_ptr(),
                                                                      ws.numel(), _stream(x.device)))
            ctx.save_for_backward(x, weight, bias, mean, var, None)
            ctx.eps, ctx.keep_scale = eps, 1.0
            ctx.mark_non_differentiable(mean, var)
            return y, mean, var
        if keep is not None and (keep.dtype != torch.uint8 or keep.shape != x.shape or not keep.is_contiguous() or keep.device != x.device):
            raise ValueError("bn_relu_train: keep must be a contiguous uint8 [N, C] tensor on x's device")
        y, mean, var = torch.empty_like(x), torch.empty_like(weight), torch.empty_like(weight)
        with torch.cuda.device(x.device):
            ws = _workspace(lib.gvqa_bn_train_workspace_bytes(N, Cc), x.device)
            _lib.check(lib.gvqa_bn_relu_dropout_train_forward(N, Cc, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), eps, _ptr(keep),
                                                              float(keep_scale), y.data_ptr(), mean.data_ptr(), var.data_ptr(), ws.data_ptr(),
                                                              ws.numel(), _stream(x.device)))
        ctx.save_for_backward(x, weight, bias, mean, var, keep)
        ctx.eps, ctx.keep_scale = eps, float(keep_scale)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        lib = _lib.load()
        x, weight, bias, mean, var, keep = ctx.saved_tensors
        N, Cc = x.shape
        dy = dy.contiguous()
        if ctx.rng is not None and dy.data_ptr() % 16:
            dy = dy.clone(memory_format=torch.contiguous_format)      # (a view at an unaligned offset: the rng kernels read 16-byte quads; a fresh allocation is aligned -- ADVICE r05: never raise from backward for this)
        dx, dw, db = torch.empty_like(x), torch.empty_like(weight), torch.empty_like(bias)
        if ctx.rng is not None:
            with torch.cuda.device(x.device):
                ws = _workspace(lib.gvqa_bn_train_workspace_bytes(N, Cc), x.device)
                _lib.check(lib.gvqa_bn_relu_dropout_train_backward_rng(N, Cc, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), mean.data_ptr(),
                                                                       var.data_ptr(), ctx.eps, ctx.rng[0], ctx.rng[1], float(ctx.rng[2]), dy.data_ptr(),
                                                                       dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                                                       _stream(x.device)))
            return dx, dw, db, None, None, None, None
        with torch.cuda.device(x.device):
            ws = _workspace(lib.gvqa_bn_train_workspace_bytes(N, Cc), x.device)
            _lib.check(lib.gvqa_bn_relu_dropout_train_backward(N, Cc, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), mean.data_ptr(),
                                                               var.data_ptr(), ctx.eps, _ptr(keep), ctx.keep_scale, dy.data_ptr(), dx.data_ptr(),
                                                               dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), _stream(x.device)))
        return dx, dw, db, None, None, None, None


def _bn_relu_train(bn: torch.nn.BatchNorm1d, x: Tensor, p: float = 0.0) -> Tensor:
    """dropout_p(relu(bn(x))) in training mode on the HIP kernels, with torch's running-statistics update (momentum, unbiased
    variance, num_batches_tracked).  p > 0: the keep mask is drawn here with torch's generator (one byte per element) and applied
    inside the BatchNorm passes, forward and backward."""
    if bn.weight is None or bn.bias is None or x.shape[0] < 2 or p >= 1.0:
        # (p = 1: F.dropout returns zeros -- gat_skip.py:276 -- and BatchNorm still updates its running statistics)
        return torch.nn.functional.dropout(torch.relu(bn(x)), p=p, training=p > 0)
    rng = keep = None
    if p > 0 and x.is_cuda and x.shape[1] % 4 == 0 and x.data_ptr() % 16 == 0 and x.is_contiguous() and not (_TRAIN_AB & 32):
        # the decisions are drawn in the kernels (Philox, keyed on torch's CUDA generator: reproducible from torch.manual_seed) -- the counters this
        # call uses are reserved on the generator, as torch's own dropout does
        gen = torch.cuda.default_generators[x.device.index if x.device.index is not None else torch.cuda.current_device()]
        if hasattr(gen, "get_offset") and hasattr(gen, "set_offset"):
            off = gen.get_offset()
            gen.set_offset(off + 4 * ((x.numel() // 4 + 3) // 4))
            rng = (gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, off, p)
    if p > 0 and rng is None:
        keep = torch.empty(x.shape, dtype=torch.uint8, device=x.device).bernoulli_(1.0 - p)
    y, mean, var = _BatchNormReluTrain.apply(x, bn.weight, bn.bias, bn.eps, keep, 1.0 / (1.0 - p) if p > 0 else 1.0, rng)
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            n = x.shape[0]
            stats = [bn.running_mean, bn.running_var]             # (same arithmetic as mul_().add_() per tensor, two launches instead of four)
            torch._foreach_mul_(stats, 1 - mom)
            torch._foreach_add_(stats, [mean, var * (n / (n - 1))], alpha=mom)
    return y


class _AddGraphRows(torch.autograd.Function):
    """x[i, :] += rows[graph(i), :] in place (x must be a fresh intermediate).  Backward: dx = dout,
    drows[b] = sum of dout over the nodes of graph b -- both HIP kernels, deterministic (the gather's native
    backward is a sort-based index_put)."""

    @staticmethod
    def forward(ctx, x, rows, graph):
        lib = _lib.load()
        rows = _f32c(rows, "rows")
        if not x.is_contiguous() or x.dtype != torch.float32 or x.shape[0] != graph.num_nodes or \
                rows.shape != (graph.num_graphs, x.shape[1]):
            raise ValueError("add_graph_rows: x must be contiguous fp32 [N, F], rows [B, F]")
        with torch.cuda.device(x.device):
            _lib.check(lib.gvqa_graph_rows_to_nodes(C.byref(graph.c), x.shape[1], rows.data_ptr(), rows.shape[1], x.data_ptr(),
                                                    x.shape[1], 1, _stream(x.device)))
        ctx.mark_dirty(x)
        ctx.graph = graph
        return x

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        graph = ctx.graph
        dout = dout.contiguous()
        drows = torch.empty((graph.num_graphs, dout.shape[1]), dtype=torch.float32, device=dout.device)
        with torch.cuda.device(dout.device):
            _lib.check(lib.gvqa_graph_segment_sum(C.byref(graph.c), dout.shape[1], dout.data_ptr(), dout.shape[1],
                                                  drows.data_ptr(), drows.shape[1], _stream(dout.device)))
        return dout, drows, None


def add_graph_rows(x: Tensor, rows: Tensor, graph: SceneGraphBatch) -> Tensor:
    """x [N, F] (a fresh intermediate, updated in place) + rows[graph of node] ([B, F]); differentiable in both."""
    return _AddGraphRows.apply(x, rows, graph)


class _GraphRows(torch.autograd.Function):
    """rows[graph(i), :] for every node i -> [N, F] (HIP broadcast); backward: per-graph sum of dout."""

    @staticmethod
    def forward(ctx, rows, graph):
        lib = _lib.load()
        rows = _f32c(rows, "rows")
        if rows.dim() != 2 or rows.shape[0] != graph.num_graphs:
            raise ValueError("graph_rows: rows must be [B, F]")
        out = torch.empty((graph.num_nodes, rows.shape[1]), dtype=torch.float32, device=rows.device)
        with torch.cuda.device(rows.device):
            _lib.check(lib.gvqa_graph_rows_to_nodes(C.byref(graph.c), rows.shape[1], rows.data_ptr(), rows.shape[1], out.data_ptr(),
                                                    rows.shape[1], 0, _stream(rows.device)))
        ctx.graph = graph
        return out

    @staticmethod
    def backward(ctx, dout):
        return _segment_sum_raw(dout.contiguous(), ctx.graph), None


class _GraphSegmentSum(torch.autograd.Function):
    """out[b, :] = sum of x[i, :] over the nodes of graph b (HIP, deterministic); backward: broadcast of dout."""

    @staticmethod
    def forward(ctx, x, graph):
        x = _f32c(x, "x")
        if x.dim() != 2 or x.shape[0] != graph.num_nodes:
            raise ValueError("graph_segment_sum: x must be [N, F]")
        ctx.graph = graph
        return _segment_sum_raw(x, graph)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        graph, dout = ctx.graph, dout.contiguous()
        dx = torch.empty((graph.num_nodes, dout.shape[1]), dtype=torch.float32, device=dout.device)
        with torch.cuda.device(dout.device):
            _lib.check(lib.gvqa_graph_rows_to_nodes(C.byref(graph.c), dout.shape[1], dout.data_ptr(), dout.shape[1], dx.data_ptr(),
                                                    dout.shape[1], 0, _stream(dout.device)))
        return dx, None


def _segment_sum_raw(x: Tensor, graph: SceneGraphBatch, mean: bool = False) -> Tensor:
    lib = _lib.load()
    out = torch.empty((graph.num_graphs, x.shape[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check((lib.gvqa_graph_segment_mean if mean else lib.gvqa_graph_segment_sum)(
            C.byref(graph.c), x.shape[1], x.data_ptr(), x.shape[1], out.data_ptr(), x.shape[1], _stream(x.device)))
    return out


def graph_rows(rows: Tensor, graph: SceneGraphBatch) -> Tensor:
    """Differentiable rows[batch] ([B, F] -> [N, F]) on the HIP kernels."""
    return _GraphRows.apply(rows, graph)


def graph_segment_sum(x: Tensor, graph: SceneGraphBatch) -> Tensor:
    """Differentiable per-graph sum of node rows ([N, F] -> [B, F]) on the HIP kernels."""
    return _GraphSegmentSum.apply(x, graph)


def _edge_rows_sum_raw(x: Tensor, g: SceneGraphBatch) -> Tensor:
    lib = _lib.load()
    out = torch.empty((g.num_nodes, x.shape[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.gvqa_graph_edge_rows_sum(C.byref(g.c), x.shape[1], x.data_ptr(), x.shape[1], out.data_ptr(), x.shape[1],
                                                _stream(x.device)))
    return out


class _EdgeGather(torch.autograd.Function):
    """x[src] or x[dst] per COO edge ([N, F] -> [E, F]); backward: HIP CSR row sums over the forward graph (dst) or the
    transposed one (src) instead of torch's sort-based index_put."""

    @staticmethod
    def forward(ctx, x, graph, side):
        edge_index = graph._keep[0]
        ctx.graph, ctx.side = graph, side
        return x.index_select(0, edge_index[side])

    @staticmethod
    def backward(ctx, dout):
        g = ctx.graph if ctx.side == 1 else ctx.graph.transposed()
        return _edge_rows_sum_raw(_f32c(dout, "dout"), g), None, None


class _EdgeScatterAdd(torch.autograd.Function):
    """out[i] = sum of m[e] over the in-edges e of node i (torch_scatter.scatter_add by destination, HIP, deterministic);
    backward: dm[e] = dout[dst_e]."""

    @staticmethod
    def forward(ctx, m, graph):
        ctx.graph = graph
        return _edge_rows_sum_raw(_f32c(m, "m"), graph)

    @staticmethod
    def backward(ctx, dout):
        return dout.index_select(0, ctx.graph._keep[0][1]), None


def edge_gather(x: Tensor, graph: SceneGraphBatch, side: str) -> Tensor:
    """Differentiable x[src] (side='src') / x[dst] (side='dst') per edge with a HIP adjoint."""
    return _EdgeGather.apply(x, graph, 0 if side == "src" else 1)


def edge_scatter_add(m: Tensor, graph: SceneGraphBatch) -> Tensor:
    """Differentiable per-destination sum of per-edge rows ([E, F] -> [N, F]) on the HIP kernel."""
    return _EdgeScatterAdd.apply(m, graph)


def graph_softmax(score: Tensor, graph: SceneGraphBatch) -> Tensor:
    """Softmax of score [N, F] over the nodes of each graph (torch_geometric.utils.softmax semantics: max-shifted,
    denominator + 1e-16), differentiable; the per-graph reductions / broadcasts are the HIP ops above."""
    B = graph.num_graphs
    idx = graph.node_graph.long().unsqueeze(1).expand_as(score)
    gmax = torch.full((B, score.shape[1]), float("-inf"), device=score.device).scatter_reduce(
        0, idx, score.detach(), reduce="amax", include_self=True)
    gmax = torch.where(torch.isinf(gmax), torch.zeros_like(gmax), gmax)
    ex = (score - graph_rows(gmax, graph)).exp()
    return ex / (graph_rows(graph_segment_sum(ex, graph), graph) + 1e-16)


def gat_message_passing(xp: Tensor, a_node: Tensor, a_edge: Tensor, graph: SceneGraphBatch, heads: int, channels: int,
                        negative_slope: float = 0.2, alpha_mask: Optional[Tensor] = None, graph_rows: Optional[Tensor] = None,
                        bias: Optional[Tensor] = None, skip: Optional[Tensor] = None, _skip_grad: Optional[list] = None):
    """Differentiable GAT message passing on the HIP kernels: (out [N, C], alpha [E, H]).
    xp [N, H*C] projected features, a_node [N, 2H] = (a_l | a_r), a_edge [E, H]; alpha_mask [E, H] multiplies alpha
    after the softmax (attention dropout: mask / (1 - p)); graph_rows [B, H*C]: rows added to xp per graph (kept out of xp);
    bias [C], skip [N, C]: added to the result in the same pass."""
    if (channels % 4 != 0 or heads > 8) and (graph_rows is not None or bias is not None or skip is not None):
        # the one-pass kernels move 16 bytes at a time and hold at most 8 heads: other shapes take the explicit form (same math,
        # torch adds)
        if graph_rows is not None:
            xp = add_graph_rows(xp, graph_rows, graph)
        out, alpha = _GatMessagePassing.apply(xp, a_node, a_edge, alpha_mask, graph, heads, channels, negative_slope, None, None, None)
        if bias is not None:
            out = out + bias
        if skip is not None:
            out = out + skip
        return out, alpha
    return _GatMessagePassing.apply(xp, a_node, a_edge, alpha_mask, graph, heads, channels, negative_slope, graph_rows, bias, skip, _skip_grad)


class gat(torch.nn.Module):
    """Edge- and instruction-conditioned GAT layer (reference class `gat`, gat_skip.py:16-213).

    forward(x, edge_index, edge_attr, size=None, return_attention_weights=None) -> out [N, C]
    (concat=False) / [N, H C] (concat=True) or (out, (edge_index, alpha [E, H])).  No self-loops are added
    (gat_skip.py:111-177 never uses `add_self_loops`).  `in_channels` may be a pair (gat_skip.py:78-80): separate `lin_l` /
    `lin_r`, `x` then a pair (x_l, x_r) of node tensors over the SAME node set (x_r may be None); graphs with different
    source / destination node sets are not taken (GraphVQA has none).  The eval fast path (one C call, gvqa_gat_conv_forward)
    serves the form gat_seq uses -- int in_channels, concat=False; everything else runs the HIP message passing per call
    (concat=True: once per head) around torch-side projections.
    """

    def __init__(self, in_channels, out_channels: int, edge_in_channels: int, heads: int = 1,
                 concat: bool = True, negative_slope: float = 0.2, dropout: float = 0.0,
                 add_self_loops: bool = True, bias: bool = True, **kwargs):
        super().__init__()
        _lib.load()   # fail loudly at construction when the HIP library is missing
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.add_self_loops = add_self_loops
        if isinstance(in_channels, int):
            self.lin_l = Linear(in_channels, heads * out_channels, bias=False)
            self.lin_r = self.lin_l                  # shared (gat_skip.py:76-77); separate state_dict key
        else:
            self.lin_l = Linear(in_channels[0], heads * out_channels, bias=False)     # gat_skip.py:79-80
            self.lin_r = Linear(in_channels[1], heads * out_channels, bias=False)
        self.lin_e = Linear(edge_in_channels, heads * out_channels, bias=False)
        self.att_e = Parameter(torch.empty(1, heads, out_channels))
        self.att_l = Parameter(torch.empty(1, heads, out_channels))
        self.att_r = Parameter(torch.empty(1, heads, out_channels))
        if bias and concat:
            self.bias = Parameter(torch.empty(heads * out_channels))
        elif bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        for t in (self.lin_l.weight, self.lin_e.weight, self.att_l, self.att_r, self.att_e):
            _glorot(t)
        if self.lin_r is not self.lin_l:
            _glorot(self.lin_r.weight)
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()

    def _params(self, bn: Optional[torch.nn.BatchNorm1d] = None, keep: Optional[list] = None) -> "_lib.GatConvParams":
        """Device pointers of this layer's parameters.  `keep` collects the (possibly re-laid-out)
        tensors so that they outlive the C call."""
        keep = [] if keep is None else keep

        def ptr(t, name):
            t = _f32c(t, name)
            keep.append(t)
            return t.data_ptr()

        p = _lib.GatConvParams()
        p._keep = keep
        p.lin_l_weight = ptr(self.lin_l.weight, "lin_l.weight")
        p.lin_e_weight = ptr(self.lin_e.weight, "lin_e.weight")
        p.att_l, p.att_r, p.att_e = ptr(self.att_l, "att_l"), ptr(self.att_r, "att_r"), ptr(self.att_e, "att_e")
        p.bias = None if self.bias is None else ptr(self.bias, "bias")
        if bn is not None:
            p.bn_weight, p.bn_bias = ptr(bn.weight, "bn.weight"), ptr(bn.bias, "bn.bias")
            p.bn_mean, p.bn_var = ptr(bn.running_mean, "bn.running_mean"), ptr(bn.running_var, "bn.running_var")
        return p

    def _forward_autograd(self, x, edge_index, edge_attr, graph, want_alpha):
        """Differentiable single layer (gat_skip.py:125-177): projections and logits as torch ops, message passing and
        its backward on the HIP kernels; attention dropout (:205) as a drawn mask in training.  Also the general form: a pair
        (x_l, x_r) with separate lin_l / lin_r (:136-143), and concat=True (:162-163) -- the message passing then runs once per
        head (H = 1 calls on that head's columns), the heads' results side by side."""
        import torch.nn.functional as F
        pair = isinstance(x, (tuple, list))
        x_l, x_r = (x[0], x[1]) if pair else (x, x)
        H, Cc, N, E = self.heads, self.out_channels, x_l.shape[0], edge_index.shape[1]
        xp = _ProjectionLinear.apply(x_l, self.lin_l.weight)
        if not pair:
            a_node = skinny_linear(x_l, fold_attention(self.lin_l.weight, self.att_l, self.att_r, H))
        else:
            a_l = skinny_linear(x_l, fold_attention(self.lin_l.weight, self.att_l, None, H))
            a_r = skinny_linear(x_r, fold_attention(self.lin_r.weight, self.att_r, None, H)) if x_r is not None else torch.zeros_like(a_l)
            a_node = torch.cat((a_l, a_r), dim=1)
        a_edge = skinny_linear(edge_attr, fold_attention(self.lin_e.weight, self.att_e, None, H))
        p = self.dropout if self.training else 0.0
        mask = torch.bernoulli(torch.full((E, H), 1.0 - p, device=xp.device)) / (1.0 - p) if p > 0 else None
        if not self.concat:
            out, alpha = gat_message_passing(xp, a_node, a_edge, graph, H, Cc, self.negative_slope, mask)
        else:
            outs, alphas = [], []
            for h in range(H):
                o_h, al_h = gat_message_passing(xp[:, h * Cc:(h + 1) * Cc].contiguous(),
                                                torch.stack((a_node[:, h], a_node[:, H + h]), dim=1).contiguous(),
                                                a_edge[:, h:h + 1].contiguous(), graph, 1, Cc, self.negative_slope,
                                                None if mask is None else mask[:, h:h + 1].contiguous())
                outs.append(o_h); alphas.append(al_h)
            out, alpha = torch.cat(outs, dim=1), torch.cat(alphas, dim=1)
        if self.bias is not None:
            out = out + self.bias
        return (out, (edge_index, alpha)) if want_alpha else out

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor, size=None,
                return_attention_weights=None, graph: Optional[SceneGraphBatch] = None):
        pair = isinstance(x, (tuple, list))
        assert (x[0] if pair else x).dim() == 2, "Static graphs not supported in `GATConv`."      # gat_skip.py:132,137
        lib = _lib.load()
        edge_attr = _f32c(edge_attr, "edge_attr")
        if pair or self.concat or self.lin_r is not self.lin_l:
            # the forms gat_seq never uses (gat_skip.py:78-80,136-143,162-163): the general formulation, with or without gradients
            if not pair and self.lin_r is not self.lin_l:
                x = (x, x)
            if pair:
                x = (_f32c(x[0], "x[0]"), None if x[1] is None else _f32c(x[1], "x[1]"))
                if x[1] is not None and x[1].shape[0] != x[0].shape[0]:
                    raise NotImplementedError("gat: source and destination node sets of different sizes (bipartite graphs) are not supported")
            else:
                x = _f32c(x, "x")
            n_nodes = (x[0] if pair else x).shape[0]
            if graph is None:
                graph = SceneGraphBatch(edge_index, None, n_nodes, 1)
            return self._forward_autograd(x, edge_index, edge_attr, graph, isinstance(return_attention_weights, bool))
        x = _f32c(x, "x")
        N, E = x.shape[0], edge_index.shape[1]
        if graph is None:
            graph = SceneGraphBatch(edge_index, None, N, 1)
        H, Cc = self.heads, self.out_channels
        d = _lib.GatDims(self.in_channels, edge_attr.shape[1], 0, Cc, H, 1, self.negative_slope, 1e-5)
        if x.shape[1] != self.in_channels or edge_attr.shape[1] != self.lin_e.weight.shape[1]:
            raise ValueError("feature width does not match the layer")
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or edge_attr.requires_grad or
                                                  any(q.requires_grad for q in self.parameters()))
        if needs_grad or (self.training and self.dropout > 0):
            return self._forward_autograd(x, edge_index, edge_attr, graph, isinstance(return_attention_weights, bool))
        p = self._params()
        out = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
        want_alpha = isinstance(return_attention_weights, bool)
        alpha = torch.empty((E, H), dtype=torch.float32, device=x.device) if want_alpha else None
        with torch.cuda.device(x.device):
            ws = _workspace(lib.gvqa_gat_conv_workspace_bytes(C.byref(graph.c), C.byref(d)), x.device)
            _lib.check(lib.gvqa_gat_conv_forward(C.byref(graph.c), C.byref(d), C.byref(p), x.data_ptr(),
                                                 edge_attr.data_ptr(), out.data_ptr(), _ptr(alpha),
                                                 ws.data_ptr(), ws.numel(), _stream(x.device)))
        if want_alpha:
            return out, (edge_index, alpha)
        return out

    def __repr__(self):
        return "{}({}, {}, heads={})".format(self.__class__.__name__, self.in_channels, self.out_channels,
                                             self.heads)


class gat_seq(torch.nn.Module):
    """K hops of instruction-conditioned GAT with skip, BN, ReLU (reference `gat_seq`,
    gat_skip.py:220-279).  forward(x, edge_index, edge_attr, instr_vectors, batch) -> h [N, out]."""

    def __init__(self, in_channels, out_channels, edge_attr_dim, ins_dim, num_ins, dropout=0.0, gat_heads=4,
                 gat_negative_slope=0.2, gat_bias=True):
        super().__init__()
        self.convs = torch.nn.ModuleList([
            gat(in_channels=in_channels + ins_dim, out_channels=out_channels,
                edge_in_channels=edge_attr_dim + ins_dim, heads=gat_heads, concat=False,
                negative_slope=gat_negative_slope, dropout=dropout, bias=gat_bias) for _ in range(num_ins)])
        self.bns = torch.nn.ModuleList([torch.nn.BatchNorm1d(out_channels) for _ in range(num_ins - 1)])
        self.dropout = dropout
        self.in_channels, self.out_channels = in_channels, out_channels
        self.edge_attr_dim, self.ins_dim, self.heads = edge_attr_dim, ins_dim, gat_heads
        self.negative_slope = gat_negative_slope
        self.last_stats = None
        # Per-module overrides of the library's process-wide options (None = follow gvqa_set_option / the environment), carried in the
        # dims struct of every call -- two models with different settings can live in one process:
        #   projection: "split2h" | "split3" | "f32"      hop_fusion: 0 .. 6 (include/gvqa.h, GVQA_OPT_HOP_FUSION: 3 = the default rule, 4 / 5 aggregate-first per hop / one launch, 6 column parts)
        self.projection = None
        self.hop_fusion = None

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()

    def forward(self, x, edge_index, edge_attr, instr_vectors, batch, graph: Optional[SceneGraphBatch] = None,
                return_attention_weights: bool = False, return_hops: bool = False):
        lib = _lib.load()
        assert x.dim() == 2, "Static graphs not supported in `GATConv`."
        x = _f32c(x, "x")
        edge_attr = _f32c(edge_attr, "edge_attr")
        instr = _f32c(instr_vectors, "instr_vectors")
        K = len(self.convs)
        N, E = x.shape[0], edge_index.shape[1]
        if instr.dim() != 3 or instr.shape[0] < K or instr.shape[2] != self.ins_dim:
            raise ValueError(f"instr_vectors must be [>= {K}, B, {self.ins_dim}]")
        B = instr.shape[1]
        if x.shape[1] != self.in_channels or edge_attr.shape[1] != self.edge_attr_dim:
            raise ValueError("feature width does not match the module")
        if graph is None:
            graph = SceneGraphBatch(edge_index, batch, N, B)
        elif graph.num_nodes != N or graph.num_edges != E or graph.num_graphs != B:
            raise ValueError("prebuilt graph does not match the inputs")
        plist = self._param_list()
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or edge_attr.requires_grad or instr.requires_grad or
                                                  any(p.requires_grad for p in plist))
        if needs_grad or (self.training and self.dropout > 0):
            return self._forward_autograd(x, edge_index, edge_attr, instr, batch, graph, return_attention_weights,
                                          return_hops)
        if (return_attention_weights or return_hops) and (not graph.intra_graph or self.training):
            # the cross-graph fallback and the fused batch-statistics path return the final tensor only: the same
            # (out, alpha, hops) triple comes from the differentiable formulation run without gradients
            with torch.no_grad():
                return self._forward_autograd(x, edge_index, edge_attr, instr, batch, graph, return_attention_weights,
                                              return_hops)
        if not graph.intra_graph:
            return self._forward_unfolded(x, edge_index, edge_attr, instr, batch, graph)
        H, Cc = self.heads, self.out_channels
        d = self._dims()
        hops, keep = self._hop_params()
        dev = x.device
        out = torch.empty((N, Cc), dtype=torch.float32, device=dev)
        if self.training:
            return self._forward_train_bn(lib, graph, d, hops, x, edge_attr, instr, out)
        alpha = torch.empty((K, E, H), dtype=torch.float32, device=dev) if return_attention_weights else None
        hop_out = torch.empty((K, N, Cc), dtype=torch.float32, device=dev) if return_hops else None
        with torch.cuda.device(dev):
            ws = _workspace(lib.gvqa_gat_seq_workspace_bytes(C.byref(graph.c), C.byref(d)), dev)
            layout = lib.gvqa_gat_seq_weight_layout(C.byref(graph.c), C.byref(d))
            cache = self._weight_cache(lib, d, hops, layout, dev)
            _lib.check(lib.gvqa_gat_seq_forward_cached(C.byref(graph.c), C.byref(d), hops, x.data_ptr(),
                                                       edge_attr.data_ptr(), instr.data_ptr(), out.data_ptr(),
                                                       _ptr(alpha), _ptr(hop_out), cache.data_ptr(), cache.numel(), layout,
                                                       ws.data_ptr(), ws.numel(), _stream(dev)))
        if return_attention_weights or return_hops:
            return out, alpha, hop_out
        return out

    def _param_list(self):
        """The module's Parameters and BatchNorm buffers in a fixed order.  Walking the module tree costs 50 us per call, so the walk
        is done once and records WHERE each tensor lives (the owning module's `_parameters` / `_buffers` dict and key); every call
        then re-reads those ~10 K slots (a few us) and notices a replaced Parameter or buffer object (`conv.lin_l.weight =
        nn.Parameter(...)`, `load_state_dict(assign=True)`, weight tying) -- which `id(self.convs)` alone would not."""
        sig = (len(self.convs), len(self.bns), id(self.convs), id(self.bns))
        if getattr(self, "_plist_sig", None) != sig:
            slots = []
            for mod in self.modules():
                slots += [(mod._parameters, k) for k, v in mod._parameters.items() if v is not None]
            for bn in self.bns:
                slots += [(bn._buffers, "running_mean"), (bn._buffers, "running_var")]
            self._pslots, self._plist_sig = slots, sig
            self._plist = [d[k] for d, k in slots]
        else:
            cur = [d.get(k) for d, k in self._pslots]
            if any(a is not b for a, b in zip(cur, self._plist)):
                if any(a is None for a in cur):          # a slot disappeared: walk again
                    self._plist_sig = None
                    return self._param_list()
                self._plist = cur
        return self._plist

    def invalidate_weight_cache(self):
        """Drop everything derived from the parameters (folded attention vectors, packed projection weights, hop structs).  The
        cache keys hold each tensor's storage pointer and in-place version counter, which optimizers, `load_state_dict`, `.to()` and
        ordinary in-place ops bump; edits through `.data` (`p.data.mul_()`, EMA swaps, manual checkpoint loading) do NOT bump the
        counter -- call this after them."""
        self._wc_key = self._wc_buf = None
        self._hops_key = None
        self._plist_sig = None

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)              # .to() / .cuda() / .float(): storages may move or be replaced
        self.invalidate_weight_cache()
        return out

    def _dims(self) -> "_lib.GatDims":
        K, H, Cc = len(self.convs), self.heads, self.out_channels
        d = _lib.GatDims(self.in_channels, self.edge_attr_dim, self.ins_dim, Cc, H, K, self.negative_slope,
                         self.bns[0].eps if len(self.bns) else 1e-5)
        if self.projection is not None:
            d.projection = 1 + {"split3": _lib.PROJECTION_SPLIT3, "f32": _lib.PROJECTION_F32, "split2h": _lib.PROJECTION_SPLIT2H}[self.projection]
        if self.hop_fusion is not None:
            d.hop_fusion = 1 + int(self.hop_fusion)
        return d

    def hop_kernel(self, graph: SceneGraphBatch) -> str:
        """Which hop kernel an eval forward of `graph` runs under this module's settings (one of _lib.HOP_KERNELS)."""
        d = self._dims()
        rc = _lib.load().gvqa_gat_seq_hop_kernel(C.byref(graph.c), C.byref(d))
        if rc < 0:
            _lib.check(rc)
        return _lib.HOP_KERNELS[rc]

    def _hop_params(self):
        """The K `gvqa_gat_conv_params` structs of the eval forward, rebuilt only when a tensor's storage moved."""
        plist = self._param_list()
        ptrs = tuple(p.data_ptr() for p in plist)
        if not all(p.dtype == torch.float32 and p.is_contiguous() for p in plist):
            ptrs = None         # a parameter needs a converted copy per call: nothing to cache
        if ptrs is None or getattr(self, "_hops_key", None) != ptrs:
            K = len(self.convs)
            hops = (_lib.GatConvParams * K)()
            keep = []   # parameter tensors referenced by raw pointer stay alive as long as the structs are cached
            for i, conv in enumerate(self.convs):
                hops[i] = conv._params(self.bns[i] if i != K - 1 else None, keep)
            self._hops, self._hops_keep, self._hops_key = hops, keep, ptrs
        return self._hops, self._hops_keep

    def _weight_cache(self, lib, d, hops, layout, dev) -> Tensor:
        """Parameter-only products of the forward (folded attention vectors, per-graph term weights, packed
        projection weights) prepared once per (parameter state, layout, device): the key holds every parameter's storage
        pointer and in-place version counter, so an optimizer step, load_state_dict or .to() invalidates it."""
        key = (layout, dev, tuple((p.data_ptr(), p._version) for p in self._param_list()))
        if getattr(self, "_wc_key", None) != key:
            buf = torch.empty(max(int(lib.gvqa_gat_seq_weight_cache_bytes(C.byref(d), layout)), 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.gvqa_gat_seq_prepare_weights(C.byref(d), hops, layout, buf.data_ptr(), buf.numel(), _stream(dev)))
            self._wc_buf, self._wc_key = buf, key
        return self._wc_buf

    def _forward_train_bn(self, lib, graph, d, hops, x, edge_attr, instr, out):
        """model.train() with dropout p = 0: BatchNorm uses batch statistics over all N rows
        (gat_skip.py:274) and its running statistics are updated like torch does (momentum, unbiased
        variance, num_batches_tracked)."""
        K, N, Cc, dev = len(self.convs), x.shape[0], self.out_channels, x.device
        stats = torch.empty((max(K - 1, 1), 2, Cc), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = _workspace(lib.gvqa_gat_seq_workspace_bytes(C.byref(graph.c), C.byref(d)), dev)
            _lib.check(lib.gvqa_gat_seq_forward_trainbn(C.byref(graph.c), C.byref(d), hops, x.data_ptr(),
                                                        edge_attr.data_ptr(), instr.data_ptr(), out.data_ptr(),
                                                        stats.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
        with torch.no_grad():
            for j, bn in enumerate(self.bns):
                if not bn.track_running_stats:
                    continue
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                unbiased = stats[j, 1] * (N / max(N - 1, 1))
                bn.running_mean.mul_(1 - mom).add_(stats[j, 0], alpha=mom)
                bn.running_var.mul_(1 - mom).add_(unbiased, alpha=mom)
        return out

    def _forward_autograd(self, x, edge_index, edge_attr, instr, batch, graph, return_attention_weights=False,
                          return_hops=False, alpha_masks=None, feature_masks=None):
        """Differentiable forward (training): per hop, gat_skip.py:254-276 with the instruction halves of the two
        concatenations applied per graph instead of per row --
            xp      = h W_h^T + (ins W_i^T)[batch]                  (lin_l on [h | ins[batch]], :133,263-264)
            a_l|a_r = h V_n + (ins U_n)[batch],   a_e = edge_attr V_e   (att_* folded into lin_l / lin_e, :134-135,150-151;
                      the edge's own instruction term ins[batch[src]] U_e, :257-260, is carried by a_l)
        -- the per-graph rows are added and back-propagated by HIP kernels (gvqa_graph_rows_to_nodes / _segment_sum) --
        then the HIP message passing (and its HIP backward), + bias, skip, BatchNorm, ReLU, dropout as torch
        ops.  Attention dropout (:205) is a mask on alpha drawn with torch's generator (not the reference's
        stream: dropout masks are not reproducible across implementations); `alpha_masks` / `feature_masks`
        (lists, one per hop) override the drawn masks (tests)."""
        import torch.nn.functional as F
        K, H, Cc = len(self.convs), self.heads, self.out_channels
        N, E = x.shape[0], edge_index.shape[1]
        Dn, De = self.in_channels, self.edge_attr_dim
        p = self.dropout if self.training else 0.0
        h = x
        alphas, hops = [], []
        # edge logits of ALL hops in one pass over edge_attr (and one pass in the backward): the H columns of every hop side by side
        # (att_e through lin_e, one launch per hop: rows [:De] act on edge_attr, rows [De:] on the instruction half, :257-260)
        folds_e = [fold_attention(c.lin_e.weight, c.att_e, None, H) for c in self.convs]
        a_edge_all = skinny_linear(edge_attr, torch.cat([f[:De] for f in folds_e], dim=1))
        a_edge_cols = _ColumnBlocks.apply(a_edge_all, K, H) if a_edge_all.shape[1] == K * H else [a_edge_all[:, i * H:(i + 1) * H] for i in range(K)]
        for i, conv in enumerate(self.convs):
            ins = instr[i]
            W, We = conv.lin_l.weight, conv.lin_e.weight
            # projected features: node half per row, instruction half per graph
            # (the per-graph rows ride through the message passing as `graph_rows`: the [N, H*C] sum is never formed)
            # attention logits through the attention vectors folded into the weights ([D, H] matrices): a_l | a_r per
            # node; the edge's instruction term ins[batch[src]] . U_e (:257-260) rides on the source half a_l
            fold_n = fold_attention(W, conv.att_l, conv.att_r, H)              # [Dn + Di, 2H]: att_l | att_r through lin_l
            # h feeds two nodes of the hop, the products and the message passing's skip: the skip's gradient travels between their backwards in
            # `sg` and is added inside the kernel that writes dh, not by autograd (one [N, D] pass per hop less).  Only where the one-pass
            # message-passing op applies (otherwise the skip is a torch add that needs its own gradient)
            sg = [] if (h.requires_grad and torch.is_grad_enabled() and Cc % 4 == 0 and H <= 8 and not (_TRAIN_AB & 2)) else None
            xp, a_part, xp_rows, a_rows = _HopProducts.apply(h, ins, W, fold_n, folds_e[i][De:], Dn, sg)
            a_node = add_graph_rows(a_part, a_rows, graph)
            a_edge = a_edge_cols[i]
            mask = None
            if alpha_masks is not None:
                mask = alpha_masks[i]
            elif p > 0:
                mask = _attention_dropout_mask(E, H, p, x.device)
            # aggregation + head mean (:155-165) + bias (:167-168) + skip (:270) in one op
            # (per-graph rows stay out of xp when every edge stays inside its graph -- any batch the reference's collate makes;
            # with cross-graph edges a message carries the SOURCE graph's row and the sum is formed)
            if not graph.intra_graph:
                xp, xp_rows = add_graph_rows(xp, xp_rows, graph), None
            h, alpha = gat_message_passing(xp, a_node, a_edge, graph, H, Cc, self.negative_slope, mask, graph_rows=xp_rows,
                                           bias=conv.bias, skip=h.detach() if sg is not None else h, _skip_grad=sg)
            if i != K - 1:
                if feature_masks is not None:                  # (tests: given masks)
                    h = (_bn_relu_train(self.bns[i], h) if self.training else torch.relu(self.bns[i](h))) * feature_masks[i]
                elif self.training:                            # BatchNorm + ReLU + feature dropout (:273-276) in the same passes
                    h = _bn_relu_train(self.bns[i], h, p)
                else:
                    h = torch.relu(self.bns[i](h))
            alphas.append(alpha)
            hops.append(h)
        if return_attention_weights or return_hops:
            return (h, torch.stack(alphas) if return_attention_weights else None,
                    torch.stack(hops) if return_hops else None)
        return h

    def _forward_unfolded(self, x, edge_index, edge_attr, instr, batch, graph):
        """Batches whose edges cross graphs (never produced by the reference's collate): run the
        reference's literal per-hop formulation (gat_skip.py:254-276) on the generic conv op."""
        if self.training:
            raise NotImplementedError("train-mode forward needs an intra-graph batch")
        K = len(self.convs)
        h = x
        edge_batch = batch[edge_index[0]]
        for i in range(K):
            ins = instr[i]
            edge_cat = torch.cat((edge_attr, ins[edge_batch]), dim=-1)
            x_cat = torch.cat((h, ins[batch]), dim=-1)
            h = self.convs[i](x_cat, edge_index, edge_cat, graph=graph) + h
            if i != K - 1:
                h = torch.relu(self.bns[i](h))
        return h
