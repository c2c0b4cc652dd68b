import torch
from torch_scatter import scatter


def maybe_num_nodes(index, num_nodes=None):
    if num_nodes is not None:
        return num_nodes
    return int(index.max()) + 1 if index.numel() else 0


def softmax(src, index, ptr=None, num_nodes=None):
    N = maybe_num_nodes(index, num_nodes)
    out = src - scatter(src, index, dim=0, dim_size=N, reduce="max")[index]
    out = out.exp()
    out_sum = scatter(out, index, dim=0, dim_size=N, reduce="sum")[index]
    return out / (out_sum + 1e-16)


def degree(index, num_nodes=None, dtype=None):
    N = maybe_num_nodes(index, num_nodes)
    out = torch.zeros((N,), dtype=dtype, device=index.device)
    one = torch.ones((index.size(0),), dtype=out.dtype, device=out.device)
    return out.scatter_add_(0, index, one)


def add_remaining_self_loops(edge_index, edge_weight=None, fill_value=1., num_nodes=None):
    N = maybe_num_nodes(edge_index, num_nodes)
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop_index = torch.arange(0, N, dtype=row.dtype, device=row.device)
    loop_index = loop_index.unsqueeze(0).repeat(2, 1)
    if edge_weight is not None:
        inv_mask = ~mask
        loop_weight = torch.full((N,), fill_value, dtype=edge_weight.dtype,
                                 device=edge_index.device)
        remaining = edge_weight[inv_mask]
        if remaining.numel() > 0:
            loop_weight[row[inv_mask]] = remaining
        edge_weight = torch.cat([edge_weight[mask], loop_weight], dim=0)
    edge_index = torch.cat([edge_index[:, mask], loop_index], dim=1)
    return edge_index, edge_weight


def remove_self_loops(*a, **k):  # name-only
    raise NotImplementedError


def add_self_loops(*a, **k):  # name-only
    raise NotImplementedError
