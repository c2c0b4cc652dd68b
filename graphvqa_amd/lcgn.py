"""MI355X-native drop-in for the reference's `lcgn` module (baseline_and_test_models/lcgn.py).

`lcgn_seq` keeps the reference's constructor arguments (lcgn.py:255-256), forward signature
(:303) and state_dict keys (incl. the dead `bns.*` entries); the whole forward runs in one C-ABI
call (`gvqa_lcgn_seq_forward`).  `x_ctx` is initialised exactly like the reference:
`torch.randn(x_loc.size())` drawn on the CPU generator, then moved (lcgn.py:306), so the same
`torch.manual_seed` gives the same result.  Inference only (dropout layers inactive).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
from torch.nn import Linear, Parameter

from . import _lib
from .gat_skip import (_f32c, _workspace, _glorot, _ProjectionLinear, gat_message_passing, graph_rows,
                       edge_gather)
from .graph import SceneGraphBatch, _stream


class _StoreBf16(torch.autograd.Function):
    """A per-node tensor as the bf16-node-feature mode keeps it in HBM: rounded to bf16 (to nearest even) in the forward, the
    gradient passed straight through -- the rounding is a storage format, not part of the model."""

    @staticmethod
    def forward(ctx, v):
        return v.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class gat_lcgn(nn.Module):
    """Parameter container of the reference's `gat_lcgn` layer (lcgn.py:63-118).  Its compute is fused
    into `lcgn_seq.forward`; calling it on its own is not part of the GraphVQA path."""

    def __init__(self, in_channels, out_channels, edge_in_channels, heads=1, concat=True, negative_slope=0.2,
                 dropout=0.0, cmd_dim=512, add_self_loops=True, bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.lin_l = Linear(in_channels, heads * out_channels, bias=False)
        self.lin_r = Linear(in_channels, heads * out_channels, bias=False)
        self.cal_x = Linear(in_channels, heads * out_channels, bias=False)
        self.proj_cmd = Linear(cmd_dim, heads * out_channels, False)
        self.cal_cmd = Linear(cmd_dim, heads * out_channels, False)
        if bias and concat:
            self.bias = Parameter(torch.empty(heads * out_channels))
        elif bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        for lin in (self.lin_l, self.lin_r, self.proj_cmd, self.cal_cmd, self.cal_x):
            _glorot(lin.weight)
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()

    def forward(self, *a, **k):
        raise NotImplementedError("gat_lcgn is executed inside lcgn_seq.forward on the HIP path")


class lcgn_seq(nn.Module):
    def __init__(self, in_channels, out_channels, edge_attr_dim, num_ins, gat_cmd_dim=512, question_dim=512,
                 MAX_ITER_NUM=4, dropout=0.0, gat_heads=1, gat_negative_slope=0.2, gat_bias=True,
                 node_feature_dtype: torch.dtype = torch.float32, bf16_weight_pieces: int = 2):
        super().__init__()
        _lib.load()
        if node_feature_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("node_feature_dtype must be torch.float32 or torch.bfloat16")
        # build-side choice (BASELINE config 5): per-node tensors stored as bf16 in HBM, node GEMMs on the
        # bf16 matrix cores (fp32 accumulation) against the fp32 weights split into `bf16_weight_pieces`
        # bf16 pieces (2: the weights keep 16 significant bits; 1: weights rounded to bf16)
        if bf16_weight_pieces not in (1, 2):
            raise ValueError("bf16_weight_pieces must be 1 or 2")
        self.node_feature_dtype = node_feature_dtype
        self.bf16_weight_pieces = bf16_weight_pieces
        self._packed, self._packed_key = None, None
        self.init_sg_emb_input = nn.Sequential(Linear(in_channels, out_channels), nn.Dropout(dropout))
        self.MAX_ITER_NUM = MAX_ITER_NUM
        self.qInput1 = Linear(question_dim, out_channels)
        for t in range(MAX_ITER_NUM):
            setattr(self, "qInput2_%d" % t, Linear(out_channels, out_channels))
        self.cmd_inter2logits = Linear(out_channels, 1)
        self.dropout = dropout
        self.proj_x_loc = nn.Sequential(nn.Dropout(dropout), Linear(out_channels, out_channels))
        self.proj_x_ctx = nn.Sequential(nn.Dropout(dropout), Linear(out_channels, out_channels))
        self.output_layer = Linear(2 * out_channels, out_channels)
        self.fin_layer = Linear(2 * out_channels, out_channels)
        self.lcgn = gat_lcgn(in_channels=3 * out_channels, out_channels=out_channels, edge_in_channels=1,
                             heads=gat_heads, concat=False, negative_slope=gat_negative_slope, dropout=dropout,
                             bias=gat_bias, cmd_dim=gat_cmd_dim)
        self.bns = nn.ModuleList([nn.BatchNorm1d(out_channels) for _ in range(num_ins - 1)])   # dead in the reference too
        self.in_channels, self.out_channels, self.question_dim = in_channels, out_channels, question_dim
        self.gat_cmd_dim, self.gat_heads, self.negative_slope = gat_cmd_dim, gat_heads, gat_negative_slope

    def _forward_autograd(self, x, edge_index, q, lstm, graph, x_ctx_init):
        """Differentiable / training formulation of lcgn.py:303-323: the node-sized dense layers (with the module's
        dropouts) and their dx / dW run on the library's own products (_ProjectionLinear: the arithmetic
        GVQA_OPT_PROJECTION selects, as in the eval path), the per-question layers are torch ops; the per-graph command broadcasts, the softmax-weighted neighbour sum and their backward run on the HIP
        per-graph / message-passing kernels (the dot-product logit x_l[src] . (proj_cmd * x_r)[dst], lcgn.py:154,207,
        enters the message passing as its per-edge term).
        node_feature_dtype = bfloat16 (BASELINE config 5): the per-node tensors are rounded to bf16 at the points where the
        inference kernels store them (x, x_loc, proj_x_loc, the x_loc block of the joint projection, x_ctx, the gated product,
        x_l | x_r | x_val, the message), with a straight-through gradient; the fp32 master weights are used as they are (the
        inference path multiplies by their two bf16 pieces: 16 of their 24 bits) and gradients stay fp32."""
        rb = _StoreBf16.apply if self.node_feature_dtype == torch.bfloat16 else (lambda v: v)
        if self.gat_heads != 1:
            raise NotImplementedError("lcgn_seq: gat_heads != 1 is not implemented (reference default 1)")
        O, N, E = self.out_channels, x.shape[0], edge_index.shape[1]
        L = self.lcgn
        proj = _ProjectionLinear.apply              # node-sized products + their dx / dW on the library's own GEMMs
        init, p_loc, p_ctx = self.init_sg_emb_input, self.proj_x_loc, self.proj_x_ctx
        x_loc = init[1](rb(proj(rb(x), init[0].weight, init[0].bias)))                # :305
        x_ctx = rb(x_ctx_init)                                                        # :306
        plin = lambda m, v: proj(v, m.weight, m.bias)      # per-question Linears ([B, .]-sized): the library's f32-input kernel, not torch's vendor GEMM
        q_emb = torch.relu(plin(self.qInput1, q))                                     # :307
        proj_x_loc = rb(proj(p_loc[0](x_loc), p_loc[1].weight, p_loc[1].bias))        # :308
        lo = lstm.transpose(1, 0)                                                     # [B, L, O]
        zeros2 = torch.zeros((N, 2), device=x.device)
        p_att = L.dropout if self.training else 0.0
        # lin_l / lin_r / cal_x act on x_joint = [x_loc | x_ctx | proj_x_ctx(x_ctx) * proj_x_loc] (:312-313, :144-145, :230) as
        # one stacked [3O, 3O] weight; its x_loc column block multiplies an iteration-invariant operand: applied once
        w_joint = torch.cat([L.lin_l.weight, L.lin_r.weight, L.cal_x.weight], dim=0)
        z_loc = rb(proj(x_loc, w_joint[:, :O]))
        w_iter = w_joint[:, O:]
        for t in range(self.MAX_ITER_NUM):
            q_cmd = plin(getattr(self, "qInput2_%d" % t), q_emb)                      # :292-300
            # cmd_inter2logits is a [1, O] Linear: its product with q_cmd * lstm_out is a weighted sum over channels
            att = torch.softmax(((q_cmd * self.cmd_inter2logits.weight)[:, None, :] * lo).sum(dim=-1) + self.cmd_inter2logits.bias, dim=-1)
            cmd = (att[:, :, None] * lo).sum(dim=1)                                    # (a weighted sum over the L question tokens: no batched GEMM)
            x_pair = torch.cat([x_ctx, rb(proj(p_ctx[0](x_ctx), p_ctx[1].weight, p_ctx[1].bias) * proj_x_loc)], dim=-1)
            x_l, x_r, x_val = rb(proj(x_pair, w_iter) + z_loc).split(O, dim=1)
            y = graph_rows(plin(L.proj_cmd, cmd), graph) * x_r                        # :148-154
            a_edge = (edge_gather(x_l, graph, "src") * edge_gather(y, graph, "dst")).sum(dim=-1, keepdim=True)   # :207
            mask = torch.bernoulli(torch.full((E, 1), 1.0 - p_att, device=x.device)) / (1.0 - p_att) if p_att > 0 else None
            agg, _ = gat_message_passing(x_val, zeros2, a_edge, graph, 1, O, self.negative_slope, mask)   # :209-238
            msg = agg * graph_rows(plin(L.cal_cmd, cmd), graph)                       # :231 (edges are intra-graph)
            if L.bias is not None:
                msg = msg + L.bias
            x_ctx = rb(proj(torch.cat([x_ctx, rb(msg)], dim=-1), self.output_layer.weight, self.output_layer.bias))   # :316-319
        return proj(torch.cat([x_loc, x_ctx], dim=-1), self.fin_layer.weight, self.fin_layer.bias)           # :321-322

    def forward(self, x, edge_index, batch, q_encoding, lstm_outputs, edge_attr=None, instr_vectors=None,
                graph: SceneGraphBatch | None = None, x_ctx_init: torch.Tensor | None = None):
        lib = _lib.load()
        x = _f32c(x, "x")
        q = _f32c(q_encoding, "q_encoding")
        lstm = _f32c(lstm_outputs, "lstm_outputs")
        N, O = x.shape[0], self.out_channels
        B, L = q.shape[0], lstm.shape[0]
        if self.gat_cmd_dim != O or lstm.shape[2] != O:
            raise ValueError("command width must equal out_channels (cmd is a weighted sum of lstm_outputs)")
        if lstm.shape[1] != B or x.shape[1] != self.in_channels or q.shape[1] != self.question_dim:
            raise ValueError("input shapes do not match the module")
        if x_ctx_init is None:
            x_ctx_init = torch.randn((N, O)).to(x.device)          # lcgn.py:306: CPU generator, then moved
        x_ctx_init = _f32c(x_ctx_init, "x_ctx_init")
        if graph is None:
            graph = SceneGraphBatch(edge_index, batch, N, B)
        if self.training or (torch.is_grad_enabled() and (x.requires_grad or q.requires_grad or lstm.requires_grad or
                                                          any(w.requires_grad for w in self.parameters()))):
            return self._forward_autograd(x, edge_index, q, lstm, graph, x_ctx_init)
        d = _lib.LcgnDims(self.in_channels, O, self.question_dim, self.MAX_ITER_NUM, L, self.gat_heads,
                          self.negative_slope,
                          (3 - self.bf16_weight_pieces) if self.node_feature_dtype == torch.bfloat16 else 0)
        p = _lib.LcgnParams()
        keep = []

        def ptr(t, name):
            t = _f32c(t, name)
            keep.append(t)
            return t.data_ptr()

        p.init_weight, p.init_bias = ptr(self.init_sg_emb_input[0].weight, "w"), ptr(self.init_sg_emb_input[0].bias, "b")
        p.qinput1_weight, p.qinput1_bias = ptr(self.qInput1.weight, "w"), ptr(self.qInput1.bias, "b")
        for t in range(self.MAX_ITER_NUM):
            lin = getattr(self, "qInput2_%d" % t)
            p.qinput2_weight[t], p.qinput2_bias[t] = ptr(lin.weight, "w"), ptr(lin.bias, "b")
        p.cmd_logit_weight, p.cmd_logit_bias = ptr(self.cmd_inter2logits.weight, "w"), ptr(self.cmd_inter2logits.bias, "b")
        p.proj_x_loc_weight, p.proj_x_loc_bias = ptr(self.proj_x_loc[1].weight, "w"), ptr(self.proj_x_loc[1].bias, "b")
        p.proj_x_ctx_weight, p.proj_x_ctx_bias = ptr(self.proj_x_ctx[1].weight, "w"), ptr(self.proj_x_ctx[1].bias, "b")
        p.output_weight, p.output_bias = ptr(self.output_layer.weight, "w"), ptr(self.output_layer.bias, "b")
        p.fin_weight, p.fin_bias = ptr(self.fin_layer.weight, "w"), ptr(self.fin_layer.bias, "b")
        p.lin_l_weight, p.lin_r_weight = ptr(self.lcgn.lin_l.weight, "w"), ptr(self.lcgn.lin_r.weight, "w")
        p.cal_x_weight = ptr(self.lcgn.cal_x.weight, "w")
        p.proj_cmd_weight, p.cal_cmd_weight = ptr(self.lcgn.proj_cmd.weight, "w"), ptr(self.lcgn.cal_cmd.weight, "w")
        p.bias = None if self.lcgn.bias is None else ptr(self.lcgn.bias, "b")
        out = torch.empty((N, O), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            # call-invariant weight forms (stacked blocks, bf16 pieces): rebuilt only when a weight tensor was
            # replaced or modified in place (data_ptr / autograd version counter) or the mode changed
            key = (d.node_bf16, str(x.device)) + tuple((t.data_ptr(), t._version) for t in keep)
            if self._packed is None or self._packed_key != key:
                nbytes = lib.gvqa_lcgn_pack_bytes(C.byref(d))
                packed = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                _lib.check(lib.gvqa_lcgn_pack_weights(C.byref(d), C.byref(p), packed.data_ptr(), nbytes, _stream(x.device)))
                self._packed, self._packed_key = packed, key
            p.packed, p.packed_bytes = self._packed.data_ptr(), self._packed.numel()
            ws = _workspace(lib.gvqa_lcgn_seq_workspace_bytes(C.byref(graph.c), C.byref(d)), x.device)
            _lib.check(lib.gvqa_lcgn_seq_forward(C.byref(graph.c), C.byref(d), C.byref(p), x.data_ptr(), q.data_ptr(),
                                                 lstm.data_ptr(), x_ctx_init.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), _stream(x.device)))
        return out
