"""Shim of torch_scatter (test infrastructure; see ../README.md)."""
import torch


def _bcast(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    while index.dim() < src.dim():
        index = index.unsqueeze(-1)
    return index.expand_as(src), dim


def _out(src, dim, dim_size, index):
    size = list(src.size())
    if dim_size is not None:
        size[dim] = dim_size
    elif index.numel() == 0:
        size[dim] = 0
    else:
        size[dim] = int(index.max()) + 1
    return size


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    index, dim = _bcast(index, src, dim)
    if out is None:
        out = torch.zeros(_out(src, dim, dim_size, index), dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


scatter_sum = scatter_add


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    s = scatter_add(src, index, dim, out, dim_size)
    idx, d = _bcast(index, src, dim)
    cnt = torch.zeros_like(s).scatter_add_(d, idx, torch.ones_like(src))
    return s / cnt.clamp(min=1)


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    idx, d = _bcast(index, src, dim)
    size = _out(src, d, dim_size, idx)
    res = torch.full(size, float("-inf"), dtype=src.dtype, device=src.device)
    res = res.scatter_reduce(d, idx, src, reduce="amax", include_self=True)
    res = torch.where(torch.isinf(res) & (res < 0), torch.zeros_like(res), res)
    return res, None


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if reduce in ("sum", "add"):
        return scatter_add(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == "max":
        return scatter_max(src, index, dim, out, dim_size)[0]
    raise ValueError(reduce)
