"""MI355X-native counterpart of the step right before the execution path:
`GroundTruth_SceneGraph_Encoder` (pipeline_model_gat.py:553-610).

Same forward signature (`forward(gt_scene_graphs) -> (x_encoded, edge_attr_encoded, None)`, reading
`.x`, `.edge_attr`, `.edge_index`, `.batch`, `.added_sym_edge`) and state_dict keys as the reference;
the constructor takes the vocabulary size / padding index explicitly instead of importing them from
`gqa_dataset_entry` (the reference reads `len(SG_ENCODING_TEXT.vocab)`, pipeline_model_gat.py:556-562).
"""
from __future__ import annotations

import ctypes as C

import torch
from torch.nn import Sequential as Seq, Linear as Lin, ReLU, Parameter

from . import _lib
from .gat_skip import _f32c, _workspace, graph_rows, graph_segment_sum, edge_scatter_add, _ProjectionLinear, _edge_rows_sum_raw
from .graph import SceneGraphBatch, _stream


class _NoGradRow(torch.autograd.Function):
    """Identity on a [V, D] table whose row `row` receives no gradient (nn.Embedding's padding_idx rule for a use of the table
    that is not an embedding lookup)."""

    @staticmethod
    def forward(ctx, w, row):
        ctx.row = row
        return w.view_as(w)

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        g[ctx.row].zero_()
        return g, None


class _EmbedSum(torch.autograd.Function):
    """out[r] = (+-) sum_t table[tokens[r, t]] (pipeline_model_gat.py:583-593): the fused gather-sum kernel of the inference path
    forward (gvqa_embed_sum); the table gradient by torch's own sort-based embedding backward on the rows' gradients repeated per
    token (deterministic, like nn.Embedding's)."""

    @staticmethod
    def forward(ctx, tokens, table, negate):
        lib = _lib.load()
        table = _f32c(table, "table")
        tokens = tokens.contiguous()
        rows, T = tokens.shape
        V, D = table.shape
        out = torch.empty((rows, D), dtype=torch.float32, device=table.device)
        with torch.cuda.device(table.device):
            _lib.check(lib.gvqa_embed_sum(rows, T, V, D, tokens.data_ptr(), table.data_ptr(), None if negate is None else negate.data_ptr(),
                                          out.data_ptr(), _stream(table.device)))
        ctx.save_for_backward(tokens, negate)
        ctx.V = V
        return out

    @staticmethod
    def backward(ctx, g):
        tokens, negate = ctx.saved_tensors
        rows, T = tokens.shape
        if negate is not None:
            g = g * (1.0 - 2.0 * negate.to(g.dtype)).view(-1, 1)
        ge = g.unsqueeze(1).expand(rows, T, g.shape[1]).reshape(rows * T, g.shape[1])
        return None, torch.ops.aten.embedding_dense_backward(ge, tokens.reshape(-1), ctx.V, -1, False), None


class _GatherAddRelu(torch.autograd.Function):
    """relu(a[src] + b[dst] + y + bias) per edge in one pass (gvqa_gather_add_relu): the first Linear of an edge-level MLP with its
    node-side column blocks projected per node (pipeline_model_gat.py:65-76,92-95).  b may be None.  Backward: the ReLU mask on the
    incoming gradient once, then the gathers' adjoints as CSR row sums (by source on the transposed graph, by destination on the
    forward one), the bias gradient as its column sum."""

    @staticmethod
    def forward(ctx, a, b, y, bias, graph):
        lib = _lib.load()
        def rows(t_, name):                           # a column block of a wider product keeps its row stride
            if t_ is None or (t_.is_cuda and t_.dtype == torch.float32 and t_.dim() == 2 and t_.stride(1) == 1):
                return t_
            return _f32c(t_, name)
        a, b, y, bias = rows(a, "a"), rows(b, "b"), _f32c(y, "y"), _f32c(bias, "bias")
        ei = graph._keep[0]
        E, D = y.shape
        out = torch.empty_like(y)
        with torch.cuda.device(y.device):
            _lib.check(lib.gvqa_gather_add_relu(E, D, a.data_ptr(), a.stride(0), ei[0].data_ptr(), None if b is None else b.data_ptr(),
                                                0 if b is None else b.stride(0), None if b is None else ei[1].data_ptr(), bias.data_ptr(),
                                                y.data_ptr(), out.data_ptr(), _stream(y.device)))
        ctx.save_for_backward(out)
        ctx.graph, ctx.has_b = graph, b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dpre = torch.ops.aten.threshold_backward(dout.contiguous(), out, 0.0)
        da = _edge_rows_sum_raw(dpre, ctx.graph.transposed()) if ctx.needs_input_grad[0] else None
        db = _edge_rows_sum_raw(dpre, ctx.graph) if (ctx.has_b and ctx.needs_input_grad[1]) else None
        dbias = dpre.sum(0) if ctx.needs_input_grad[3] else None
        return da, db, dpre, dbias, None


def _no_grad_row(w, row):
    return _NoGradRow.apply(w, int(row))


class _EdgeModel(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.edge_mlp = Seq(Lin(3 * d, d), ReLU(), Lin(d, d))


class _NodeModel(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.node_mlp_1 = Seq(Lin(2 * d, d), ReLU(), Lin(d, d))
        self.node_mlp_2 = Seq(Lin(2 * d, d), ReLU(), Lin(d, d))


class _MetaLayer(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.edge_model = _EdgeModel(d)
        self.node_model = _NodeModel(d)
        self.global_model = None


class _GraphLayerNorm(torch.nn.Module):
    """graph_utils/my_graph_layernorm.LayerNorm: weight / bias are 1-element tensors (:38-39)."""

    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = Parameter(torch.ones(1))
        self.bias = Parameter(torch.zeros(1))


class GroundTruth_SceneGraph_Encoder(torch.nn.Module):
    def __init__(self, vocab_size: int, pad_idx: int = 0, sg_emb_dim: int = 300):
        super().__init__()
        _lib.load()
        self.sg_emb_dim = sg_emb_dim
        self.sg_vocab_embedding = torch.nn.Embedding(vocab_size, sg_emb_dim, padding_idx=pad_idx)
        self.scene_graph_encoding_layer = _MetaLayer(sg_emb_dim)
        self.graph_layer_norm = _GraphLayerNorm(sg_emb_dim)

    def _forward_autograd(self, x_tok, e_tok, ei, added, graph):
        """Differentiable formulation of pipeline_model_gat.py:575-610 (training), in the split-column form of the fused path
        (csrc/encoder.hip): the concatenations [x_src | x_dst | e], [x_src | e'], [x | agg] are never built -- the first Linear of
        every MLP is applied by column block, node blocks per NODE and gathered (HIP gathers with CSR row sums as adjoints), the
        edge block of EdgeModel's first Linear to the embedding TABLE (the per-edge product becomes the token lookup; its weight
        gradient a [V, D] product instead of a reduction over all E rows).  Every node- / edge-sized product is the library's
        (`_ProjectionLinear`: two-piece split GEMMs forward, dx and dW from one pass over dy); the graph LayerNorm's per-graph
        reductions / broadcasts run on the HIP per-graph ops with their adjoints (my_graph_layernorm.py:57-78: eps OUTSIDE the
        square root)."""
        F = torch.nn.functional
        emb, m, D = self.sg_vocab_embedding, self.scene_graph_encoding_layer, self.sg_emb_dim
        N = x_tok.shape[0]
        proj = _ProjectionLinear.apply
        e0, e2l = m.edge_model.edge_mlp[0], m.edge_model.edge_mlp[2]
        n10, n12 = m.node_model.node_mlp_1[0], m.node_model.node_mlp_1[2]
        n20, n22 = m.node_model.node_mlp_2[0], m.node_model.node_mlp_2[2]
        # token sums as fused gather-sums (no [N, T, D] intermediate); nn.Embedding's rule that the padding row gets no gradient is
        # kept by masking that row's gradient of the table (the row itself takes part as stored, exactly like emb(tokens))
        table = _no_grad_row(emb.weight, emb.padding_idx) if emb.padding_idx is not None else emb.weight
        self._check_ids(x_tok, e_tok, added, ei.shape[1])                                # (the kernel clamps: out-of-table ids raise here)
        x = _EmbedSum.apply(x_tok, table, None)                                          # :583-587
        # edge tokens through the PROJECTED table
        te = F.linear(table, e0.weight[:, 2 * D:])                                       # [V, D]
        neg = None
        if added is not None and added.numel():
            neg = torch.zeros(e_tok.shape[0], dtype=torch.uint8, device=te.device)
            neg[added.to(te.device)] = 1                                                 # :590
        ye = _EmbedSum.apply(e_tok, te, neg)                                             # [E, D]
        dst = ei[1]
        # EdgeModel :65-76
        # (biases ride in the products' epilogues: where a sum of products has one bias, the first product takes it)
        # the four per-node column blocks that act on x -- EdgeModel's x_src and x_dst, node_mlp_1's x_src, node_mlp_2's x -- as ONE
        # product [N, D] x [D, 4D] (x packed once forward, its gradient one product backward)
        xs, xd, xp1, xt = proj(x, torch.cat((e0.weight[:, :D], e0.weight[:, D:2 * D], n10.weight[:, :D], n20.weight[:, :D]), 0), None).split(D, 1)
        y1 = _GatherAddRelu.apply(xs, xd, ye, e0.bias, graph)
        e2 = proj(y1, e2l.weight, e2l.bias)
        # NodeModel :78-98
        y3 = _GatherAddRelu.apply(xp1, None, proj(e2, n10.weight[:, D:], None), n10.bias, graph)
        mm = proj(y3, n12.weight, n12.bias)
        cnt = torch.bincount(dst, minlength=N).clamp(min=1).to(mm.dtype)
        agg = edge_scatter_add(mm, graph) / cnt.view(-1, 1)                              # scatter_mean :96
        x2 = proj(torch.relu(xt + proj(agg, n20.weight[:, D:], n20.bias)), n22.weight, n22.bias)
        gp = graph.graph_ptr.long()
        norm = ((gp[1:] - gp[:-1]).clamp(min=1) * D).to(x2.dtype).view(-1, 1)
        mean = graph_segment_sum(x2, graph).sum(dim=-1, keepdim=True) / norm
        xc = x2 - graph_rows(mean, graph)
        var = graph_segment_sum(xc * xc, graph).sum(dim=-1, keepdim=True) / norm
        out = xc / (graph_rows(var.sqrt(), graph) + self.graph_layer_norm.eps)
        return out * self.graph_layer_norm.weight + self.graph_layer_norm.bias, e2, None

    def invalidate_weight_cache(self):
        """Drop the packed / folded / projected weight forms of the eval forward.  The cache key holds every parameter's identity,
        storage pointer and in-place version counter, which optimizers, `load_state_dict`, `.to()` and ordinary in-place ops bump;
        edits through `.data` (`p.data.copy_()`, EMA swaps, manual checkpoint loading) do NOT -- call this after them."""
        self._packed = self._packed_key = self._packed_event = None

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)              # .to() / .cuda() / .float(): storages may move or be replaced
        self.invalidate_weight_cache()
        return out

    def _check_ids(self, x_tok, e_tok, added, E):
        """validate_ids: True (default) = every call, like the reference's nn.Embedding / index assignment, which raise on ANY call;
        "first" = opt-in for the loader path that never synchronises: the first 4 calls of this module only (a vocabulary / checkpoint mismatch
        shows at once; a blocking host read on every step would undo the loader path that never synchronises), False = never.
        nn.Embedding / index assignment in the reference raise on ids outside the table (a vocabulary / checkpoint mismatch); the
        kernels would clamp them silently, so the range is checked here (one small reduction and host read)."""
        V = self.sg_vocab_embedding.num_embeddings
        mode = getattr(self, "validate_ids", True)
        seen = getattr(self, "_validated_calls", 0)
        if mode is True or (mode == "first" and seen < 4):
            self._validated_calls = seen + 1
            bad = ((x_tok < 0) | (x_tok >= V)).any() | ((e_tok < 0) | (e_tok >= V)).any()
            if added is not None and added.numel():
                bad = bad | ((added < 0) | (added >= E)).any().to(bad.device)
            if bool(bad.item()):
                raise IndexError(f"scene-graph token id outside [0, {V}) or added_sym_edge index outside [0, {E})")

    def forward(self, gt_scene_graphs, graph: SceneGraphBatch | None = None):
        lib = _lib.load()
        d = gt_scene_graphs
        x_tok, e_tok, ei, batch = d.x, d.edge_attr, d.edge_index, d.batch
        added = getattr(d, "added_sym_edge", None)
        dev = x_tok.device
        if not x_tok.is_cuda:
            raise RuntimeError("scene-graph tensors are on %s: the MI355X path has no CPU fallback" % dev)
        x_tok, e_tok, ei = x_tok.contiguous(), e_tok.contiguous(), ei.contiguous()
        N, E = x_tok.shape[0], ei.shape[1]
        if graph is None:
            graph = SceneGraphBatch(ei, batch, N)
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            return self._forward_autograd(x_tok, e_tok, ei, added, graph)
        D, V = self.sg_emb_dim, self.sg_vocab_embedding.num_embeddings
        self._check_ids(x_tok, e_tok, added, E)
        m = self.scene_graph_encoding_layer
        p = _lib.EncoderParams()
        keep, srcs, copied = [], [], False
        for name, t in (("embedding", self.sg_vocab_embedding.weight),
                        ("edge0_weight", m.edge_model.edge_mlp[0].weight), ("edge0_bias", m.edge_model.edge_mlp[0].bias),
                        ("edge2_weight", m.edge_model.edge_mlp[2].weight), ("edge2_bias", m.edge_model.edge_mlp[2].bias),
                        ("node1_0_weight", m.node_model.node_mlp_1[0].weight), ("node1_0_bias", m.node_model.node_mlp_1[0].bias),
                        ("node1_2_weight", m.node_model.node_mlp_1[2].weight), ("node1_2_bias", m.node_model.node_mlp_1[2].bias),
                        ("node2_0_weight", m.node_model.node_mlp_2[0].weight), ("node2_0_bias", m.node_model.node_mlp_2[0].bias),
                        ("node2_2_weight", m.node_model.node_mlp_2[2].weight), ("node2_2_bias", m.node_model.node_mlp_2[2].bias),
                        ("ln_weight", self.graph_layer_norm.weight), ("ln_bias", self.graph_layer_norm.bias)):
            tc = _f32c(t, name)
            keep.append(tc)
            copied = copied or tc is not t            # a temporary (non-contiguous / non-fp32 parameter): its address says nothing next call
            srcs.append(t)
            setattr(p, name, tc.data_ptr())
        xe = torch.empty((N, D), dtype=torch.float32, device=dev)
        ee = torch.empty((E, D), dtype=torch.float32, device=dev)
        na = 0 if added is None else int(added.numel())
        added_c = None if na == 0 else added.to(dev).contiguous()
        with torch.cuda.device(dev):
            # call-invariant weight forms (projected table, stacked / folded / packed weights): rebuilt only when a PARAMETER was replaced
            # (object identity), moved (data_ptr) or modified in place (autograd version counter), or the projection arithmetic changed.
            # Edits through `.data` do not bump the counter: invalidate_weight_cache() after them (as gat_seq documents).  A parameter
            # that had to be copied to contiguous fp32 is never cached (the copy's address is recycled by the allocator).
            st = _stream(dev)
            key = None if copied else (str(dev), lib.gvqa_get_option(_lib.OPT_PROJECTION)) + tuple((id(t), t.data_ptr(), t._version) for t in srcs)
            if key is None or getattr(self, "_packed_key", None) != key:
                nbytes = lib.gvqa_sg_encoder_pack_bytes(V, D)
                packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                rc = lib.gvqa_sg_encoder_pack_weights(V, D, C.byref(p), packed.data_ptr(), nbytes, st)
                if rc == _lib.E_UNSUPPORTED:
                    packed = None                      # shapes / settings the large-batch path does not take: nothing to cache
                else:
                    _lib.check(rc)
                self._packed, self._packed_key = packed, key
                # the pack ran on THIS stream: a forward on another stream waits for it
                self._packed_event = torch.cuda.Event()
                self._packed_event.record(torch.cuda.current_stream(dev))
                self._packed_stream = st
            elif getattr(self, "_packed_stream", st) != st and getattr(self, "_packed_event", None) is not None:
                torch.cuda.current_stream(dev).wait_event(self._packed_event)
                if self._packed is not None:
                    # used on a stream other than the one it was allocated on: the caching allocator must not hand the block back to the packing
                    # stream while this stream's forward still reads it (a dropped / rebuilt cache would free it) -- ADVICE r05
                    self._packed.record_stream(torch.cuda.current_stream(dev))
            if self._packed is not None:
                p.packed, p.packed_bytes = self._packed.data_ptr(), self._packed.numel()
            ws = _workspace(lib.gvqa_sg_encoder_workspace_bytes(C.byref(graph.c), D), dev)
            _lib.check(lib.gvqa_sg_encoder_forward(C.byref(graph.c), V, D, x_tok.shape[1], e_tok.shape[1], C.byref(p),
                                                   x_tok.data_ptr(), e_tok.data_ptr(),
                                                   None if added_c is None else added_c.data_ptr(), na, ei.data_ptr(),
                                                   self.graph_layer_norm.eps, xe.data_ptr(), ee.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), _stream(dev)))
        return xe, ee, None
