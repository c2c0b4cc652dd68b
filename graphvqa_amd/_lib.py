"""ctypes binding of libgvqa_hip.so (C ABI declared in include/gvqa.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, importing
an operator raises `GvqaLibraryError` with the build command.  (The CPU oracle under oracle/ is
test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GVQA_LIB", os.path.join(HERE, "lib", "libgvqa_hip.so"))   # GVQA_LIB: A/B a second build

GVQA_OK, E_INVALID, E_WORKSPACE, E_HIP, E_GRAPH, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
STAGES = ("graph", "fold", "edge_logit", "graph_term", "proj", "node_logit", "mp", "other", "pack", "alpha")
# gvqa_set_option keys / values (include/gvqa.h)
OPT_PROJECTION, OPT_VENDOR_GEMM, OPT_SPLIT3_MIN_MFLOP, OPT_SPLIT3_VARIANT, OPT_HOP_FUSION, OPT_COEFF_KERNEL, OPT_MP_PARTS, OPT_HOP_COEFFS, OPT_HOP_HALF_TILES, OPT_TN_DIRECT, OPT_PACKED_GROUPS = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
PROJECTION_SPLIT3, PROJECTION_F32, PROJECTION_SPLIT2H = 0, 1, 2
NUM_STAGES = len(STAGES)
HOP_KERNELS = ("unfused", "fused8", "persistent", "fused8_chained", "persistent_chained", "aggregate_first", "aggregate_first_seq", "aggregate_first_parts")      # GVQA_HOP_*


class GvqaLibraryError(RuntimeError):
    pass


class GvqaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gvqa error {code}: {msg}")
        self.code = code


class GvqaUnsupported(GvqaError):
    pass


class Graph(C.Structure):
    """Mirror of `struct gvqa_graph`."""
    _fields_ = [("num_nodes", C.c_int64), ("num_edges", C.c_int64), ("num_graphs", C.c_int64),
                ("rowptr", C.c_void_p), ("csr_src", C.c_void_p), ("csr_eid", C.c_void_p),
                ("node_graph", C.c_void_p), ("graph_ptr", C.c_void_p), ("stats_dev", C.c_void_p),
                ("max_graph_nodes", C.c_int32), ("max_graph_edges", C.c_int32),
                ("max_in_degree", C.c_int32), ("intra_graph", C.c_int32), ("valid", C.c_int32),
                ("finalized", C.c_int32), ("row_group_ptr", C.c_void_p), ("num_row_groups", C.c_int32),
                ("max_row_group_edges", C.c_int32), ("row_group_order", C.c_void_p),
                ("pk_num_row_groups", C.c_int32), ("pk_max_row_group_edges", C.c_int32), ("pk_row_group_ptr", C.c_void_p),
                ("pk_rowptr", C.c_void_p), ("pk_csr_src", C.c_void_p), ("pk_csr_eid", C.c_void_p), ("pk_node_graph", C.c_void_p),
                ("pk_node_old", C.c_void_p), ("pk_graph_old", C.c_void_p)]


class GatConvParams(C.Structure):
    """Mirror of `struct gvqa_gat_conv_params` (device pointers)."""
    _fields_ = [(n, C.c_void_p) for n in
                ("lin_l_weight", "lin_e_weight", "att_l", "att_r", "att_e", "bias",
                 "bn_weight", "bn_bias", "bn_mean", "bn_var")]


class GatMpDesc(C.Structure):
    """Mirror of `struct gvqa_gat_mp_desc`."""
    _fields_ = [("C", C.c_int32), ("H", C.c_int32), ("negative_slope", C.c_float), ("bn_eps", C.c_float),
                ("xp", C.c_void_p), ("xp_ld", C.c_int64), ("a_node", C.c_void_p), ("a_edge", C.c_void_p),
                ("a_edge_stride", C.c_int64), ("graph_term", C.c_void_p), ("graph_term_ld", C.c_int64),
                ("graph_scale", C.c_void_p), ("graph_scale_ld", C.c_int64), ("skip", C.c_void_p),
                ("skip_ld", C.c_int64), ("bias", C.c_void_p), ("bn_weight", C.c_void_p), ("bn_bias", C.c_void_p),
                ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p), ("out", C.c_void_p), ("out_ld", C.c_int64),
                ("alpha_out", C.c_void_p), ("alpha_mask", C.c_void_p), ("force", C.c_int32),
                ("head_rows", C.c_void_p), ("head_rows_ld", C.c_int64), ("head_weight_out", C.c_void_p)]


ABSMAX_SLOTS = 256        # GVQA_ABSMAX_SLOTS


class LinearBackwardExtras(C.Structure):
    """Mirror of `struct gvqa_linear_backward_extras`."""
    _fields_ = [("x_absmax", C.c_void_p), ("x_absmax_n", C.c_int32), ("J", C.c_int32), ("lowrank_g", C.c_void_p), ("lowrank_v", C.c_void_p),
                ("addend", C.c_void_p), ("ld_addend", C.c_int64)]


class GatMpBwdDesc(C.Structure):
    """Mirror of `struct gvqa_gat_mp_bwd_desc`."""
    _fields_ = [("C", C.c_int32), ("H", C.c_int32), ("negative_slope", C.c_float), ("xp", C.c_void_p), ("xp_ld", C.c_int64),
                ("a_node", C.c_void_p), ("a_edge", C.c_void_p), ("a_edge_stride", C.c_int64), ("alpha", C.c_void_p),
                ("alpha_mask", C.c_void_p), ("dout", C.c_void_p), ("dout_ld", C.c_int64), ("dxp", C.c_void_p),
                ("dxp_ld", C.c_int64), ("da_node", C.c_void_p), ("da_edge", C.c_void_p), ("dalpha_node", C.c_void_p),
                ("dxp_absmax", C.c_void_p)]


class BnParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("weight", "bias", "mean", "var")]


class GineParams(C.Structure):
    _fields_ = [("nn0_weight", C.c_void_p), ("nn0_bias", C.c_void_p), ("nn2_weight", C.c_void_p),
                ("nn2_bias", C.c_void_p), ("eps", C.c_float)]


class GcnParams(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p)]


class LcgnDims(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("question_dim", C.c_int32),
                ("num_iters", C.c_int32), ("seq_len", C.c_int32), ("heads", C.c_int32),
                ("negative_slope", C.c_float), ("node_bf16", C.c_int32)]


class LcgnParams(C.Structure):
    _fields_ = [("init_weight", C.c_void_p), ("init_bias", C.c_void_p), ("qinput1_weight", C.c_void_p),
                ("qinput1_bias", C.c_void_p), ("qinput2_weight", C.c_void_p * 8), ("qinput2_bias", C.c_void_p * 8),
                ("cmd_logit_weight", C.c_void_p), ("cmd_logit_bias", C.c_void_p),
                ("proj_x_loc_weight", C.c_void_p), ("proj_x_loc_bias", C.c_void_p),
                ("proj_x_ctx_weight", C.c_void_p), ("proj_x_ctx_bias", C.c_void_p),
                ("output_weight", C.c_void_p), ("output_bias", C.c_void_p), ("fin_weight", C.c_void_p),
                ("fin_bias", C.c_void_p), ("lin_l_weight", C.c_void_p), ("lin_r_weight", C.c_void_p),
                ("cal_x_weight", C.c_void_p), ("proj_cmd_weight", C.c_void_p), ("cal_cmd_weight", C.c_void_p),
                ("bias", C.c_void_p), ("packed", C.c_void_p), ("packed_bytes", C.c_size_t)]


class PoolParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("node0_weight", "node0_bias", "node2_weight", "node2_bias", "ques0_weight",
                                          "ques0_bias", "ques2_weight", "ques2_bias", "gate0_weight", "gate0_bias",
                                          "gate2_weight", "gate2_bias")]


class ClassifierParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("fc1_weight", "fc1_bias", "fc2_weight", "fc2_bias")]


class EncoderParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("embedding", "edge0_weight", "edge0_bias", "edge2_weight", "edge2_bias",
                                          "node1_0_weight", "node1_0_bias", "node1_2_weight", "node1_2_bias",
                                          "node2_0_weight", "node2_0_bias", "node2_2_weight", "node2_2_bias",
                                          "ln_weight", "ln_bias", "packed")] + [("packed_bytes", C.c_size_t)]


class MpPlan(C.Structure):
    _fields_ = [("tiled", C.c_int32), ("channel_range", C.c_int32), ("stage_buffers", C.c_int32),
                ("blocks_per_cu", C.c_int32), ("stages_per_graph", C.c_int32), ("accumulators", C.c_int32),
                ("lds_bytes", C.c_int64), ("blocks_per_graph", C.c_int32)]


class GatDims(C.Structure):
    _fields_ = [("node_dim", C.c_int32), ("edge_dim", C.c_int32), ("ins_dim", C.c_int32),
                ("out_channels", C.c_int32), ("heads", C.c_int32), ("num_hops", C.c_int32),
                ("negative_slope", C.c_float), ("bn_eps", C.c_float), ("projection", C.c_int32), ("hop_fusion", C.c_int32)]


# name -> (restype, argtypes); every symbol include/gvqa.h declares
PROTOTYPES = {
    "gvqa_last_error": (C.c_char_p, []),
    "gvqa_version": (C.c_char_p, []),
    "gvqa_gemm_backend": (C.c_char_p, []),
    "gvqa_stream_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "gvqa_mfma_stream": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p]),
    "gvqa_graph_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "gvqa_graph_build": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.POINTER(Graph)]),
    "gvqa_graph_finalize": (C.c_int, [C.POINTER(Graph), C.c_void_p]),
    "gvqa_graph_build_grouped": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Graph)]),
    "gvqa_graph_check_valid": (C.c_int, [C.POINTER(Graph), C.c_void_p]),
    "gvqa_graph_finalize_host": (C.c_int, [C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gvqa_gat_conv_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.POINTER(GatDims)]),
    "gvqa_gat_conv_forward": (C.c_int, [C.POINTER(Graph), C.POINTER(GatDims), C.POINTER(GatConvParams),
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p]),
    "gvqa_gat_seq_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.POINTER(GatDims)]),
    "gvqa_gat_seq_forward": (C.c_int, [C.POINTER(Graph), C.POINTER(GatDims), C.POINTER(GatConvParams),
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_gat_seq_weight_cache_bytes": (C.c_size_t, [C.POINTER(GatDims), C.c_int32]),
    "gvqa_gat_seq_weight_layout": (C.c_int, [C.POINTER(Graph), C.POINTER(GatDims)]),
    "gvqa_gat_seq_hop_kernel": (C.c_int, [C.POINTER(Graph), C.POINTER(GatDims)]),
    "gvqa_gat_seq_prepare_weights": (C.c_int, [C.POINTER(GatDims), C.POINTER(GatConvParams), C.c_int32, C.c_void_p, C.c_size_t,
                                               C.c_void_p]),
    "gvqa_gat_seq_forward_cached": (C.c_int, [C.POINTER(Graph), C.POINTER(GatDims), C.POINTER(GatConvParams),
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_gat_seq_forward_trainbn": (C.c_int, [C.POINTER(Graph), C.POINTER(GatDims), C.POINTER(GatConvParams),
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_linear_f32": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "gvqa_pack_weight_bf16": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gvqa_linear_bf16": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                   C.c_void_p]),
    "gvqa_split3_packed_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gvqa_split3_pack": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gvqa_linear_split3": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "gvqa_split2h_packed_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gvqa_split2h_pack": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gvqa_split2h_pack_absmax": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvqa_split2h_pack_logits": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_void_p]),
    "gvqa_linear_split2h": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "gvqa_linear_split2h_chain": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gvqa_linear_f32_ex": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                     C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "gvqa_gat_message_passing": (C.c_int, [C.POINTER(Graph), C.POINTER(GatMpDesc), C.c_void_p, C.c_size_t,
                                           C.c_void_p]),
    "gvqa_bn_relu_chain": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(BnParams), C.c_float, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "gvqa_gine_conv_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.c_int32, C.c_int32, C.c_int32]),
    "gvqa_gine_conv_forward": (C.c_int, [C.POINTER(Graph), C.c_int32, C.c_int32, C.c_int32, C.POINTER(GineParams),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "gvqa_gcn_conv_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.c_int32, C.c_int32, C.c_int32]),
    "gvqa_gcn_conv_forward": (C.c_int, [C.POINTER(Graph), C.c_int32, C.c_int32, C.c_int32, C.POINTER(GcnParams),
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_lcgn_pack_bytes": (C.c_size_t, [C.POINTER(LcgnDims)]),
    "gvqa_lcgn_pack_weights": (C.c_int, [C.POINTER(LcgnDims), C.POINTER(LcgnParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_lcgn_seq_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.POINTER(LcgnDims)]),
    "gvqa_lcgn_seq_forward": (C.c_int, [C.POINTER(Graph), C.POINTER(LcgnDims), C.POINTER(LcgnParams), C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "gvqa_attention_pool_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.c_int32, C.c_int32]),
    "gvqa_attention_pool_forward": (C.c_int, [C.POINTER(Graph), C.c_int32, C.c_int32, C.POINTER(PoolParams), C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_answer_logits_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "gvqa_answer_logits_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(ClassifierParams),
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_sg_encoder_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.c_int32]),
    "gvqa_sg_encoder_pack_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gvqa_sg_encoder_pack_weights": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(EncoderParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_sg_encoder_forward": (C.c_int, [C.POINTER(Graph), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.POINTER(EncoderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_void_p]),
    "gvqa_bn_train_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "gvqa_bn_relu_train_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_bn_relu_train_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                              C.c_void_p]),
    "gvqa_bn_relu_dropout_train_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_bn_relu_dropout_train_forward_rng": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint64, C.c_uint64,
                                                         C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_bn_relu_dropout_train_backward_rng": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                                          C.c_uint64, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                          C.c_size_t, C.c_void_p]),
    "gvqa_dropout_keep_mask": (C.c_int, [C.c_int64, C.c_int32, C.c_uint64, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p]),
    "gvqa_dropout_scale_mask": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p]),
    "gvqa_bn_relu_dropout_train_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_graph_rows_to_nodes": (C.c_int, [C.POINTER(Graph), C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                           C.c_void_p]),
    "gvqa_graph_edge_rows_sum": (C.c_int, [C.POINTER(Graph), C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "gvqa_graph_segment_sum": (C.c_int, [C.POINTER(Graph), C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "gvqa_graph_segment_mean": (C.c_int, [C.POINTER(Graph), C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "gvqa_graph_head_rows_add": (C.c_int, [C.POINTER(Graph), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_int64, C.c_void_p]),
    "gvqa_graph_head_rows_backward": (C.c_int, [C.POINTER(Graph), C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvqa_linear_tn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "gvqa_linear_tn_split2h": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_linear_backward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "gvqa_linear_backward_split2h": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_linear_backward_split2h_hint": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                                    C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                                    C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_linear_backward_split2h_ex": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                                  C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                                  C.POINTER(LinearBackwardExtras), C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_gather_add_relu": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvqa_embed_sum": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvqa_fold_attention_forward": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvqa_fold_attention_backward": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvqa_skinny_forward": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvqa_skinny_backward_weight_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "gvqa_skinny_backward_weight": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_size_t, C.c_void_p]),
    "gvqa_skinny_backward_input": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                             C.c_void_p, C.c_int64, C.c_void_p]),
    "gvqa_gat_mp_backward": (C.c_int, [C.POINTER(Graph), C.POINTER(Graph), C.POINTER(GatMpBwdDesc), C.c_void_p]),
    "gvqa_gat_mp_plan": (C.c_int, [C.POINTER(Graph), C.c_int32, C.c_int32, C.POINTER(MpPlan)]),
    "gvqa_hop2_blocks_per_cu": (C.c_int, [C.c_int32]),
    "gvqa_scene_graph_collate_sizes": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvqa_scene_graph_collate": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvqa_set_option": (C.c_int, [C.c_int, C.c_int]),
    "gvqa_get_option": (C.c_int, [C.c_int]),
    "gvqa_prof_enable": (C.c_int, [C.c_int]),
    "gvqa_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


def load():
    """Load the HIP library once; raises GvqaLibraryError if it is missing or not loadable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GvqaLibraryError(
            f"{LIB_PATH} not found. Build it with `python -m graphvqa_amd.build` (needs hipcc; "
            "cross-compiles for gfx950 without a GPU). There is no CPU fallback for this path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64, wrong arch ...
        raise GvqaLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GvqaLibraryError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int):
    if rc == GVQA_OK:
        return
    msg = load().gvqa_last_error().decode("utf-8", "replace")
    if rc == E_UNSUPPORTED:
        raise GvqaUnsupported(rc, msg)
    raise GvqaError(rc, msg)


def set_option(option: int, value: int) -> int:
    """Sets a process-wide library option; returns the previous value."""
    lib = load()
    old = lib.gvqa_get_option(option)
    check(lib.gvqa_set_option(option, int(value)))
    return old


def prof_enable(on: bool, stages=None):
    """stages: None = every stage; else an iterable of stage names (STAGES) -- only those record events (each event pair costs the
    stream a few microseconds: a timed region keeps just the kernel it reports on)."""
    v = 1 if on else 0
    if on and stages:
        v |= sum(1 << (1 + STAGES.index(s)) for s in stages)
    check(load().gvqa_prof_enable(v))


def prof_collect():
    """Returns {stage: (milliseconds, launches)} accumulated since the last collect."""
    ms = (C.c_double * NUM_STAGES)()
    cnt = (C.c_int64 * NUM_STAGES)()
    check(load().gvqa_prof_collect(ms, cnt))
    return {STAGES[i]: (ms[i], cnt[i]) for i in range(NUM_STAGES)}
