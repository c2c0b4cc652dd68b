import inspect

import torch
from torch import Tensor
from torch.nn import Parameter
from torch_scatter import scatter, scatter_add
from torch_geometric.utils import add_remaining_self_loops
from torch_geometric.nn.inits import glorot, zeros, reset

_SPECIAL = {"index", "ptr", "size_i", "size_j", "edge_index", "edge_index_i",
            "edge_index_j", "adj_t", "dim_size"}


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr = aggr
        self.node_dim = node_dim
        self._msg_params = [p for p in inspect.signature(self.message).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        src_idx, dst_idx = edge_index[0], edge_index[1]
        size = [None, None] if size is None else list(size)

        def _set(k, t):
            if size[k] is None and isinstance(t, Tensor):
                size[k] = t.size(self.node_dim)

        coll = {}
        for name in self._msg_params:
            if name in _SPECIAL:
                continue
            if name.endswith("_i") or name.endswith("_j"):
                which = 1 if name.endswith("_i") else 0
                data = kwargs.get(name[:-2], None)
                if isinstance(data, (tuple, list)):
                    _set(0, data[0])
                    _set(1, data[1])
                    data = data[which]
                elif isinstance(data, Tensor):
                    _set(0, data)
                    _set(1, data)
                if isinstance(data, Tensor):
                    data = data.index_select(self.node_dim, dst_idx if which else src_idx)
                coll[name] = data
            else:
                coll[name] = kwargs.get(name, None)
        if size[0] is None:
            size[0] = size[1]
        if size[1] is None:
            size[1] = size[0]
        for name in self._msg_params:
            if name == "index":
                coll[name] = dst_idx
            elif name == "ptr":
                coll[name] = None
            elif name == "size_i" or name == "dim_size":
                coll[name] = size[1]
            elif name == "size_j":
                coll[name] = size[0]
        out = self.message(**coll)
        out = scatter(out, dst_idx, dim=self.node_dim, dim_size=size[1], reduce=self.aggr)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def update(self, inputs):
        return inputs


class GCNConv(MessagePassing):
    """PyG 1.6/1.7 GCNConv: weight [in, out], add_remaining_self_loops, sym norm."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False,
                 add_self_loops=True, normalize=True, bias=True, **kwargs):
        super().__init__(aggr="add", node_dim=0)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved = improved
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)

    def forward(self, x, edge_index, edge_weight=None):
        N = x.size(0)
        if edge_weight is None:
            edge_weight = torch.ones((edge_index.size(1),), dtype=x.dtype, device=x.device)
        fill = 2. if self.improved else 1.
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill, N)
        row, col = edge_index[0], edge_index[1]
        deg = scatter_add(edge_weight, col, dim=0, dim_size=N)
        dis = deg.pow(-0.5)
        dis.masked_fill_(dis == float("inf"), 0)
        norm = dis[row] * edge_weight * dis[col]
        x = torch.matmul(x, self.weight)
        out = self.propagate(edge_index, x=x, edge_weight=norm, size=None)
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j, edge_weight):
        return edge_weight.view(-1, 1) * x_j


class GINEConv(MessagePassing):
    def __init__(self, nn, eps=0., train_eps=False, **kwargs):
        super().__init__(aggr="add", node_dim=0)
        self.nn = nn
        self.initial_eps = eps
        if train_eps:
            self.eps = Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer("eps", torch.Tensor([eps]))
        self.reset_parameters()

    def reset_parameters(self):
        reset(self.nn)
        self.eps.data.fill_(self.initial_eps)

    def forward(self, x, edge_index, edge_attr=None, size=None):
        if isinstance(x, Tensor):
            x = (x, x)
        if edge_attr is not None:
            assert x[0].size(-1) == edge_attr.size(-1)
        out = self.propagate(edge_index, x=x, edge_attr=edge_attr, size=size)
        x_r = x[1]
        if x_r is not None:
            out = out + (1 + self.eps) * x_r
        return self.nn(out)

    def message(self, x_j, edge_attr):
        return torch.nn.functional.relu(x_j + edge_attr)
