"""MI355X-native counterparts of the step right after the execution path:
`MyConditionalGlobalAttention` (pipeline_model_gat.py:108-185) and the short-answer classifier
`logit_fc` fed with [g || q || g*q] (pipeline_model_gat.py:722-728, 800-816).

Same class name / constructor / forward signature / state_dict keys as the reference for the
pooling layer; the classifier wraps a Sequential laid out exactly like `logit_fc`
(keys `logit_fc.{1,4}.{weight,bias}`).  The produced logits [B, 1842] are the per-graph payload
that is all-gathered over RCCL in the data-parallel path.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch.nn import Sequential as Seq, Linear as Lin, ReLU

from . import _lib
from .gat_skip import _f32c, _workspace, graph_rows, graph_segment_sum, graph_softmax, _ProjectionLinear, skinny_linear
from .graph import SceneGraphBatch, _stream


class MyConditionalGlobalAttention(torch.nn.Module):
    def __init__(self, num_node_features, num_out_features):
        super().__init__()
        _lib.load()
        channels = num_out_features
        self.gate_nn = Seq(Lin(channels, channels), ReLU(), Lin(channels, 1))
        self.node_nn = Seq(Lin(num_node_features, channels), ReLU(), Lin(channels, channels))
        self.ques_nn = Seq(Lin(channels, channels), ReLU(), Lin(channels, channels))
        self.num_node_features, self.channels = num_node_features, channels

    def forward(self, x, u, batch, size=None, graph: SceneGraphBatch | None = None):
        lib = _lib.load()
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        x, u = _f32c(x, "x"), _f32c(u, "u")
        B = u.shape[0] if size is None else size
        N = x.shape[0]
        if graph is None:
            graph = SceneGraphBatch(torch.zeros((2, 0), dtype=torch.int64, device=x.device), batch, N, B)
        if torch.is_grad_enabled() and (x.requires_grad or u.requires_grad or any(q.requires_grad for q in self.parameters())):
            # differentiable formulation (pipeline_model_gat.py:149-181): the MLPs are torch ops, the per-graph
            # broadcast / softmax / sum run on the HIP per-graph kernels and their adjoints
            # (node-sized Linears on the library's products -- two-piece split GEMMs forward, dx and dW from one pass over dy -- and the
            # final 512 -> 1 gate product on the tall-skinny kernels; torch's own Linear below the size at which those pay)
            def lin(t, layer):
                return _ProjectionLinear.apply(t, layer.weight, layer.bias)
            xn = lin(torch.relu(lin(x, self.node_nn[0])), self.node_nn[2])
            z = torch.relu(lin(graph_rows(self.ques_nn(u), graph) * xn, self.gate_nn[0]))
            gate = graph_softmax(skinny_linear(z, self.gate_nn[2].weight.t()) + self.gate_nn[2].bias, graph)
            return graph_segment_sum(gate * xn, graph)
        p = _lib.PoolParams()
        keep = []
        for name, lin in (("node0", self.node_nn[0]), ("node2", self.node_nn[2]), ("ques0", self.ques_nn[0]),
                          ("ques2", self.ques_nn[2]), ("gate0", self.gate_nn[0]), ("gate2", self.gate_nn[2])):
            w, b = _f32c(lin.weight, name), _f32c(lin.bias, name)
            keep += [w, b]
            setattr(p, name + "_weight", w.data_ptr())
            setattr(p, name + "_bias", b.data_ptr())
        out = torch.empty((B, self.channels), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            ws = _workspace(lib.gvqa_attention_pool_workspace_bytes(C.byref(graph.c), self.num_node_features, self.channels),
                            x.device)
            _lib.check(lib.gvqa_attention_pool_forward(C.byref(graph.c), self.num_node_features, self.channels, C.byref(p),
                                                       x.data_ptr(), u.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                                       ws.numel(), _stream(x.device)))
        return out

    def __repr__(self):
        return "{}(gate_nn={}, node_nn={}, ques_nn={})".format(self.__class__.__name__, self.gate_nn, self.node_nn,
                                                                self.ques_nn)


class ShortAnswerClassifier(torch.nn.Module):
    """`logit_fc` of PipelineModel (pipeline_model_gat.py:718-728) + the feature concat of :814-816."""

    def __init__(self, question_hidden_dim=512, out_classifier_dim=512, num_short_answer_choices=1842):
        super().__init__()
        _lib.load()
        self.logit_fc = Seq(torch.nn.Dropout(p=0.2), Lin(question_hidden_dim * 3, out_classifier_dim), torch.nn.ELU(),
                            torch.nn.Dropout(p=0.2), Lin(out_classifier_dim, num_short_answer_choices))
        self.Q, self.hidden, self.A = question_hidden_dim, out_classifier_dim, num_short_answer_choices

    def forward(self, graph_final_feature, question_feature):
        lib = _lib.load()
        g, q = _f32c(graph_final_feature, "graph_final_feature"), _f32c(question_feature, "question_feature")
        if self.training or (torch.is_grad_enabled() and (g.requires_grad or q.requires_grad or
                                                          any(w.requires_grad for w in self.parameters()))):
            # training / differentiable: two dense layers on [B, 3Q] rows -- torch ops (with the Sequential's dropouts)
            return self.logit_fc(torch.cat((g, q, g * q), dim=-1))
        B = g.shape[0]
        p = _lib.ClassifierParams(_f32c(self.logit_fc[1].weight, "w").data_ptr(), _f32c(self.logit_fc[1].bias, "b").data_ptr(),
                                  _f32c(self.logit_fc[4].weight, "w").data_ptr(), _f32c(self.logit_fc[4].bias, "b").data_ptr())
        logits = torch.empty((B, self.A), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            ws = _workspace(lib.gvqa_answer_logits_workspace_bytes(B, self.Q, self.hidden), g.device)
            _lib.check(lib.gvqa_answer_logits_forward(B, self.Q, self.hidden, self.A, C.byref(p), g.data_ptr(), q.data_ptr(),
                                                      logits.data_ptr(), ws.data_ptr(), ws.numel(), _stream(g.device)))
        return logits
