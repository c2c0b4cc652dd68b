/*
 * gvqa.h -- C ABI of the MI355X-native scene-graph execution library (libgvqa_hip.so).
 *
 * Drop-in boundary for ONE hot path of codexxxl/GraphVQA: the scene-graph execution module
 *     gat_skip.gat / gat_skip.gat_seq            (reference gat_skip.py:16-279)
 * and its variants
 *     GINEConv / gine_seq                        (baseline_and_test_models/pipeline_model_gine.py:622-674)
 *     GCNConv  / gcn_seq                         (baseline_and_test_models/pipeline_model_gcn.py:622-669)
 *     gat_lcgn / lcgn_seq                        (baseline_and_test_models/lcgn.py:17-323)
 * The reference has no FFI of its own (it is pure Python on torch_geometric / torch_scatter);
 * these entry points are what a ctypes binding of that path binds -- see INTEGRATION.md.
 *
 * Conventions
 *  - Plain C: pointers and sizes only, no torch / C++ types.  Every function returns an int
 *    status: 0 = ok, <0 = GVQA_E_* ; no C++ exception crosses the boundary.
 *    gvqa_last_error() returns a thread-local human-readable message for the last failure.
 *  - All tensor pointers are DEVICE pointers (HBM), fp32 row-major contiguous unless a leading
 *    dimension is given, indices int64 on input (the reference's dtype, gqa_dataset_entry.py:361)
 *    and int32 inside the library.
 *  - Ownership: the caller owns every buffer, including workspaces (query *_workspace_bytes,
 *    allocate with the caller's allocator, e.g. torch's caching allocator).  The library never
 *    calls hipMalloc/hipFree on the hot path.
 *  - Threading / streams: re-entrant; work is enqueued on the caller's hipStream_t (passed as
 *    void*) and the call returns without synchronising, except gvqa_graph_finalize, which waits
 *    for the graph statistics.
 *  - Inputs are never modified (the reference never mutates its inputs either).
 */
#ifndef GVQA_H
#define GVQA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Export macro: the library is built with -fvisibility=hidden; only the entry points declared here (all `gvqa_*`) are
 * dynamic symbols of libgvqa_hip.so (tests/test_host.py checks `nm -D`). */
#if defined(__GNUC__) || defined(__clang__)
#define GVQA_API __attribute__((visibility("default")))
#else
#define GVQA_API
#endif

#define GVQA_OK 0
#define GVQA_E_INVALID (-1)    /* bad argument (null pointer, negative size, unsupported shape) */
#define GVQA_E_WORKSPACE (-2)  /* workspace too small */
#define GVQA_E_HIP (-3)        /* a HIP runtime call failed; see gvqa_last_error() */
#define GVQA_E_GRAPH (-4)      /* malformed graph (index out of range, batch not sorted) */
#define GVQA_E_UNSUPPORTED (-5)

GVQA_API const char* gvqa_last_error(void);
/* Library / build identification, e.g. "gvqa-hip 0.1 gfx950". */
GVQA_API const char* gvqa_version(void);

/* ------------------------------------------------------------------------------------------
 * Graph container: destination-sorted CSR of a batched (block-diagonal) scene-graph batch.
 *
 * Replaces what PyG's MessagePassing.__collect__ / torch_scatter index on the fly from COO
 * `edge_index` on every hop (call site gat_skip.py:155-156).  Input contract = the reference's
 * collate output (gqa_dataset_entry.py:361-369, :654): edge_index[0] = source, edge_index[1] =
 * destination, batch[n] = graph id of node n, non-decreasing.  Multi-edges and explicit
 * self-loops are ordinary edges and are preserved.
 *
 * Within a row (destination) the CSR slots are ordered by ORIGINAL edge id, so per-node
 * reductions run in the reference's COO order and results are deterministic run to run.
 * ---------------------------------------------------------------------------------------- */
typedef struct gvqa_graph {
    int64_t num_nodes, num_edges, num_graphs;
    const int32_t* rowptr;      /* [N+1]  CSR row pointer by destination node                  */
    const int32_t* csr_src;     /* [E]    source node of CSR slot s                            */
    const int32_t* csr_eid;     /* [E]    original COO edge id of CSR slot s ("eperm")         */
    const int32_t* node_graph;  /* [N]    graph id of node n (int32 copy of `batch`)           */
    const int32_t* graph_ptr;   /* [B+1]  first node of graph g; graph g's in-edges are CSR
                                          slots rowptr[graph_ptr[g]] .. rowptr[graph_ptr[g+1]] */
    const int32_t* stats_dev;   /* [8]    device copy of the statistics below                  */
    /* statistics, valid after gvqa_graph_finalize(): */
    int32_t max_graph_nodes;    /* largest graph, in nodes                                     */
    int32_t max_graph_edges;    /* largest graph, in in-edges                                  */
    int32_t max_in_degree;
    int32_t intra_graph;        /* 1 iff every edge has batch[src] == batch[dst]               */
    int32_t valid;              /* 1 iff indices in range and batch non-decreasing in [0,B)    */
    int32_t finalized;
    /* Row groups for the fused hop kernel (valid after finalize): the node range cut, in order, into groups of at most
     * 128 consecutive nodes that end on graph boundaries (group r = nodes row_group_ptr[r] .. row_group_ptr[r+1]).
     * num_row_groups == 0: not applicable (a graph with more than 128 nodes, edges across graphs, empty batch). */
    const int32_t* row_group_ptr;   /* [num_row_groups + 1], device                                */
    int32_t num_row_groups;
    int32_t max_row_group_edges;    /* most in-edges of any row group                              */
    const int32_t* row_group_order; /* [N], device: the rows of every group by in-degree, largest first (slot s of group r =
                                       local row row_group_order[row_group_ptr[r] + s]); the fused hop aggregates in this order */
    /* PACKED row groups (round 6; valid after finalize, pk_num_row_groups == 0: none).  The groups above cut the node range IN
     * ORDER, so a ragged batch leaves every group partly empty (graphs of 20..40 nodes: 114 of 128 rows on average) -- and the
     * aggregate-first hop runs ONE workgroup per group and CU: config 2's 262 groups are two rounds on 256 CUs where 235 full
     * groups would be one.  When re-ordering the GRAPHS lets the groups fill up and that saves a round (GVQA_OPT_PACKED_GROUPS),
     * finalize also leaves the batch in a packed numbering: graphs placed best-fit-decreasing into groups of <= 128 nodes and
     * <= 1024 in-edges, nodes of a graph consecutive and in their order, CSR rows copied slot for slot (the per-node COO order --
     * and with it every sum -- is unchanged).  Only the aggregate-first hop kernels read these; they gather the input rows and
     * scatter the output rows through pk_node_old, so callers never see the numbering. */
    int32_t pk_num_row_groups;
    int32_t pk_max_row_group_edges;
    const int32_t* pk_row_group_ptr; /* [pk_num_row_groups + 1] first packed node of group r                                  */
    const int32_t* pk_rowptr;        /* [N+1] CSR row pointer by packed destination node                                       */
    const int32_t* pk_csr_src;       /* [E]   packed source node of packed CSR slot s                                          */
    const int32_t* pk_csr_eid;       /* [E]   COO edge id of packed CSR slot s                                                 */
    const int32_t* pk_node_graph;    /* [N]   packed graph index (position in the packed graph order) of packed node n         */
    const int32_t* pk_node_old;      /* [N]   node id (row of x / out) of packed node n                                        */
    const int32_t* pk_graph_old;     /* [B]   graph id (row of instr_vectors[i]) of packed graph index j                       */
} gvqa_graph;

GVQA_API size_t gvqa_graph_workspace_bytes(int64_t num_nodes, int64_t num_edges, int64_t num_graphs);

/* Deferred validation of a handle finalized by gvqa_graph_finalize_host (which reads nothing back): synchronises `stream`, reads
 * the contract flags the build left on the device and the statistics the device derives from the arrays, and returns GVQA_E_GRAPH
 * when the batch violates the input contract, has cross-graph edges although the handle says intra-graph, or exceeds the
 * handle's statistics (the kernels size LDS regions from them).  For loaders: call it on the first batches / in debug runs. */
GVQA_API int gvqa_graph_check_valid(const gvqa_graph* g, void* stream);

/* Enqueue the CSR build.  `ws` (>= gvqa_graph_workspace_bytes, 256-byte aligned) backs every
 * array `out` points to and must stay alive as long as `out` is used. */
GVQA_API int gvqa_graph_build(int64_t num_nodes, int64_t num_edges, int64_t num_graphs,
                     const int64_t* edge_index /* [2,E] */, const int64_t* batch /* [N] or NULL (one graph) */,
                     void* ws, size_t ws_bytes, void* stream, gvqa_graph* out);

/* Copy the statistics to the host struct (synchronises `stream`).  Returns GVQA_E_GRAPH if the
 * input violated the contract. */
GVQA_API int gvqa_graph_finalize(gvqa_graph* g, void* stream);

/* The same WITHOUT a device synchronisation, for feeds whose loader already knows the per-graph layout on the host (the
 * reference's collate builds its Batch on the CPU, gqa_dataset_entry.py:631-675): graph_ptr_host[B+1] = first node of
 * every graph, graph_edge_ptr_host[B+1] = running count of in-edges by destination graph (graph g owns CSR slots
 * [ptr[g], ptr[g+1])), max_in_degree = the largest in-degree or 0 if unknown (the largest graph's edge count is then
 * assumed).  The statistics and the row-group plan are derived from these on the host; the device-side validation flags
 * are NOT read back: the caller vouches for an intra-graph batch with in-range indices.  GVQA_E_GRAPH if the layout does
 * not span [0, N] nodes / [0, E] edges monotonically. */
GVQA_API int gvqa_graph_finalize_host(gvqa_graph* g, const int32_t* graph_ptr_host, const int32_t* graph_edge_ptr_host,
                             int32_t max_in_degree, void* stream);

/* gvqa_graph_build + gvqa_graph_finalize_host as ONE upload and ONE launch, for loader-side layouts whose COO edges are also
 * grouped by graph: graph g's edges are the COO positions [graph_edge_ptr_host[g], graph_edge_ptr_host[g+1]) -- what the
 * reference's collate produces (Batch.from_data_list concatenates the graphs' edge lists, gqa_dataset_entry.py:654).  A row
 * group's in-edges are then one contiguous COO range and a workgroup per row group builds its CSR slice, the in-row order by
 * edge id and the row order out of LDS.  Same arrays, bit for bit, as the general pair.  Returns GVQA_E_UNSUPPORTED -- with
 * nothing enqueued and no error string -- when the shape is outside its reach (an empty batch, a graph of more than 128 nodes,
 * a row group of more than 8192 edges): the caller then uses gvqa_graph_build + gvqa_graph_finalize_host.  An edge outside its
 * group's node range (a layout that is not grouped after all) is flagged on the device like any contract violation
 * (gvqa_graph_check_valid). */
GVQA_API int gvqa_graph_build_grouped(int64_t num_nodes, int64_t num_edges, int64_t num_graphs, const int64_t* edge_index,
                             const int64_t* batch, const int32_t* graph_ptr_host, const int32_t* graph_edge_ptr_host,
                             int32_t max_in_degree, void* ws, size_t ws_bytes, void* stream, gvqa_graph* out);

/* ------------------------------------------------------------------------------------------
 * GAT execution path
 * ---------------------------------------------------------------------------------------- */
typedef struct gvqa_gat_conv_params {   /* one `gat` layer; state_dict names in comments */
    const float* lin_l_weight;  /* [H*C, in]       convs.i.lin_l.weight (== lin_r, gat_skip.py:76-77) */
    const float* lin_e_weight;  /* [H*C, edge_in]  convs.i.lin_e.weight */
    const float* att_l;         /* [H*C]           convs.i.att_l  ([1,H,C]) */
    const float* att_r;         /* [H*C]           convs.i.att_r */
    const float* att_e;         /* [H*C]           convs.i.att_e */
    const float* bias;          /* [C] or NULL     convs.i.bias (concat=False) */
    /* eval-mode BatchNorm1d + ReLU applied after the skip connection (gat_skip.py:273-275);
       all NULL = no BN/ReLU after this layer (last hop, or plain `gat`). */
    const float* bn_weight;     /* [C]  bns.i.weight */
    const float* bn_bias;       /* [C]  bns.i.bias */
    const float* bn_mean;       /* [C]  bns.i.running_mean */
    const float* bn_var;        /* [C]  bns.i.running_var */
} gvqa_gat_conv_params;

typedef struct gvqa_gat_dims {
    int32_t node_dim;    /* Dn: width of x (== out_channels for gat_seq: skip connection)    */
    int32_t edge_dim;    /* De: width of edge_attr                                            */
    int32_t ins_dim;     /* Di: width of one instruction vector (0 for a plain `gat` call)    */
    int32_t out_channels;/* C                                                                 */
    int32_t heads;       /* H (1, 2, 4 or 8)                                                  */
    int32_t num_hops;    /* K                                                                 */
    float negative_slope;
    float bn_eps;
    /* Per-call overrides of the process-wide options below (two models with different settings can share a process):
     * 0 = use gvqa_get_option(...); otherwise the option value + 1. */
    int32_t projection;  /* GVQA_OPT_PROJECTION value + 1, or 0 */
    int32_t hop_fusion;  /* GVQA_OPT_HOP_FUSION value + 1, or 0 */
} gvqa_gat_dims;

/* gat.forward(x, edge_index, edge_attr) with concat=False (gat_skip.py:111-177): `x` is the
 * already-concatenated [N, node_dim] input, `edge_attr` the already-concatenated
 * [E, edge_dim] (COO order).  out [N, C].  alpha_out: NULL or [E, H] in COO edge order
 * (return_attention_weights, gat_skip.py:170-175).  dims.ins_dim must be 0, num_hops 1.
 * Works for any graph (no intra-graph requirement). */
GVQA_API size_t gvqa_gat_conv_workspace_bytes(const gvqa_graph* g, const gvqa_gat_dims* d);
GVQA_API int gvqa_gat_conv_forward(const gvqa_graph* g, const gvqa_gat_dims* d, const gvqa_gat_conv_params* p,
                          const float* x, const float* edge_attr, float* out, float* alpha_out,
                          void* ws, size_t ws_bytes, void* stream);

/* gat_seq.forward(x, edge_index, edge_attr, instr_vectors, batch) in eval mode
 * (gat_skip.py:249-279): K hops of { [h || ins[batch]], [edge_attr || ins[batch[src]]] -> gat ->
 * + h -> (BN -> ReLU unless last) }.  x [N,Dn], edge_attr [E,De] (COO order), instr [K,B,Di],
 * out [N,C] (Dn == C).  alpha_out: NULL or [K,E,H].  hop_out: NULL or [K,N,C] (h after every hop).
 * Requires g->intra_graph (true for any PyG batch); otherwise GVQA_E_UNSUPPORTED and the caller
 * falls back to K gvqa_gat_conv_forward calls on concatenated inputs. */
GVQA_API size_t gvqa_gat_seq_workspace_bytes(const gvqa_graph* g, const gvqa_gat_dims* d);
GVQA_API int gvqa_gat_seq_forward(const gvqa_graph* g, const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops /* [K], host */,
                         const float* x, const float* edge_attr, const float* instr,
                         float* out, float* alpha_out, float* hop_out,
                         void* ws, size_t ws_bytes, void* stream);

/* Weight cache: what a gat_seq forward derives from the PARAMETERS alone -- the folded attention vectors, the per-graph
 * term weights and the split3-packed projection weights of every hop -- prepared once and reused while the weights do not
 * change (serving).  `layout` is what the batch needs: gvqa_gat_seq_weight_layout(g, d) = -1 (f32 projection) or a
 * bit mask: bit 0 = head-interleaved rows (fused hop) instead of plain row order, bit 1 = two fp16 pieces (split2h)
 * instead of three bf16 pieces (split3), bit 2 = half-interleaved rows (the persistent hop kernel, GVQA_OPT_HOP_FUSION = 2)
 * instead of head-interleaved ones.  gvqa_gat_seq_forward_cached uses the cache when its layout id
 * matches the batch's, and recomputes into the workspace otherwise (results are identical either way).  The cache is
 * caller-owned device memory, 256-byte aligned; the caller re-prepares it after changing any parameter. */
GVQA_API size_t gvqa_gat_seq_weight_cache_bytes(const gvqa_gat_dims* d, int32_t layout);
GVQA_API int gvqa_gat_seq_weight_layout(const gvqa_graph* g, const gvqa_gat_dims* d);
/* Which hop kernel an eval forward of this batch runs under the current options (introspection: bench.py labels its line with it,
 * tests assert it): GVQA_HOP_* below, or a negative status.  Plain outputs assumed (per-hop fp32 outputs / batch statistics take
 * the unchained form of the same kernel), and node rows `x` that are 16-byte aligned: the query does not see `x`, and a forward whose
 * rows are not takes the 8-wave kernels where this reports an aggregate-first form (results are the same to the stated tolerance). */
#define GVQA_HOP_UNFUSED 0            /* projection GEMM + gvqa::k_gat_mp_tiled                                          */
#define GVQA_HOP_FUSED8 1             /* the 8-wave fused kernel, a pack pass per hop                                     */
#define GVQA_HOP_PERSISTENT 2         /* the persistent kernel (hop2.hip), a pack pass per hop                            */
#define GVQA_HOP_FUSED8_CHAINED 3     /* the 8-wave fused kernel, hops chained through packed operands (default when H = 4) */
#define GVQA_HOP_PERSISTENT_CHAINED 4 /* the persistent kernel, hops chained                                              */
#define GVQA_HOP_AGGREGATE_FIRST 5    /* hopagg.hip: heads concatenated along K (GVQA_OPT_HOP_FUSION = 4)                  */
#define GVQA_HOP_AGGREGATE_FIRST_SEQ 6 /* the same, the K hops as ONE launch (GVQA_OPT_HOP_FUSION = 5)                     */
#define GVQA_HOP_AGGREGATE_FIRST_PARTS 7 /* the same, one launch per hop, a row group's 512 output columns split over four workgroups
                                            (GVQA_OPT_HOP_FUSION = 6, explicit only: measured slower than the 8-wave kernel on the shards it was built for) */
GVQA_API int gvqa_gat_seq_hop_kernel(const gvqa_graph* g, const gvqa_gat_dims* d);
GVQA_API int gvqa_gat_seq_prepare_weights(const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops, int32_t layout, void* cache,
                                 size_t cache_bytes, void* stream);
GVQA_API int gvqa_gat_seq_forward_cached(const gvqa_graph* g, const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops, const float* x,
                                const float* edge_attr, const float* instr, float* out, float* alpha_out, float* hop_out,
                                const void* weight_cache, size_t weight_cache_bytes, int32_t weight_cache_layout, void* ws,
                                size_t ws_bytes, void* stream);

/* Same forward with TRAIN-mode BatchNorm (model.train(), gat_skip.py:273-276): after every hop but the
 * last, BN uses the batch statistics over all N node rows (biased variance, eps), then ReLU.  The
 * attention / feature dropouts are NOT applied (the caller must have p = 0; they are not
 * reproducible against torch's RNG).  bn_stats_out [K-1, 2, C] receives per hop the batch mean and
 * the biased batch variance, from which the caller updates running_mean / running_var
 * (momentum, unbiased variance) exactly as torch does.  bn_mean / bn_var of `hops` are ignored. */
GVQA_API int gvqa_gat_seq_forward_trainbn(const gvqa_graph* g, const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops,
                                 const float* x, const float* edge_attr, const float* instr, float* out,
                                 float* bn_stats_out, void* ws, size_t ws_bytes, void* stream);

/* BatchNorm1d with BATCH statistics + ReLU, forward and backward (gat_skip.py:273-275 under model.train(); post-ops of
 * the differentiable path).  x, y, dy, dx: fp32 [N, C] contiguous.  save_mean / save_var [C]: batch mean and BIASED
 * variance (the caller updates running statistics).  dx = w invstd (g - sum(g)/N - xhat sum(g xhat)/N),
 * g = dy [y > 0].  ws: gvqa_bn_train_workspace_bytes(N, C).  Deterministic. */
GVQA_API size_t gvqa_bn_train_workspace_bytes(int64_t N, int32_t C);
GVQA_API int gvqa_bn_relu_train_forward(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, float eps,
                               float* y, float* save_mean, float* save_var, void* ws, size_t ws_bytes, void* stream);
GVQA_API int gvqa_bn_relu_train_backward(int64_t N, int32_t C, const float* x, const float* weight, const float* bias,
                                const float* save_mean, const float* save_var, float eps, const float* dy, float* dx,
                                float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream);
/* The same with feature dropout (gat_skip.py:276, F.dropout after BatchNorm + ReLU) applied in the same passes:
 * y = relu(bn(x)) * (keep[i] ? keep_scale : 0) with keep [N, C] bytes drawn by the caller (torch's generator: masks are the
 * caller's randomness) and keep_scale = 1 / (1 - p); the backward takes dL/dy of THAT y.  keep NULL: no dropout. */
GVQA_API int gvqa_bn_relu_dropout_train_forward(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, float eps,
                                       const uint8_t* keep, float keep_scale, float* y, float* save_mean, float* save_var, void* ws,
                                       size_t ws_bytes, void* stream);
GVQA_API int gvqa_bn_relu_dropout_train_backward(int64_t N, int32_t C, const float* x, const float* weight, const float* bias,
                                        const float* save_mean, const float* save_var, float eps, const uint8_t* keep,
                                        float keep_scale, const float* dy, float* dx, float* dweight, float* dbias, void* ws,
                                        size_t ws_bytes, void* stream);
/* The same with the keep decisions DRAWN IN THE KERNELS (F.dropout of gat_skip.py:276 without a mask tensor): Philox4x32-10, counter = (index of
 * the quad of 4 consecutive channels, offset), key = seed XOR a library constant (a stream of its own: never one of torch's under the same seed) -- the caller takes (seed, offset) from its generator (torch: initial_seed() /
 * get_offset(), then set_offset() past the N*C/4 counters used) so that runs are reproducible from torch.manual_seed; an element is kept when its
 * 32-bit draw is below (1 - p) 2^32 and scaled by 1 / (1 - p).  The backward regenerates the forward's decisions from the same (seed, offset, p).
 * C % 4 == 0 and 16-byte aligned rows (GVQA_E_UNSUPPORTED otherwise: use the explicit-mask forms).  gvqa_dropout_keep_mask writes the decisions
 * as a byte mask [N, C] (tests; reproducing a run's masks). */
GVQA_API int gvqa_bn_relu_dropout_train_forward_rng(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, float eps,
                                                    uint64_t seed, uint64_t offset, float p, float* y, float* save_mean, float* save_var, void* ws,
                                                    size_t ws_bytes, void* stream);
GVQA_API int gvqa_bn_relu_dropout_train_backward_rng(int64_t N, int32_t C, const float* x, const float* weight, const float* bias,
                                                     const float* save_mean, const float* save_var, float eps, uint64_t seed, uint64_t offset, float p,
                                                     const float* dy, float* dx, float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream);
GVQA_API int gvqa_dropout_keep_mask(int64_t N, int32_t C, uint64_t seed, uint64_t offset, float p, uint8_t* keep, void* stream);
/* mask[i] = kept ? 1 / (1 - p) : 0 for n floats (n % 4 == 0), the same generator: the multiplicative mask of F.dropout on the attention
 * coefficients (gat_skip.py:205) that gvqa_gat_mp_desc.alpha_mask takes. */
GVQA_API int gvqa_dropout_scale_mask(int64_t n, uint64_t seed, uint64_t offset, float p, float* mask, void* stream);

/* Per-graph rows <-> node rows (glue of the differentiable path: the per-graph instruction terms).
 * rows_to_nodes: out[i, :F] (= or +=) rows[graph(i), :F];  segment_sum (its adjoint): out[b, :F] = sum of x[i, :F]
 * over the nodes of graph b.  Deterministic. */
GVQA_API int gvqa_graph_rows_to_nodes(const gvqa_graph* g, int64_t F, const float* rows, int64_t ld_rows, float* out, int64_t ld_out,
                             int accumulate, void* stream);
GVQA_API int gvqa_graph_segment_sum(const gvqa_graph* g, int64_t F, const float* x, int64_t ld_x, float* out, int64_t ld_out, void* stream);
/* The same divided by max(node count of graph b, 1): the per-graph mean rows that a sharded step all-gathers when the answer
 * head is not run (torch_scatter's scatter_mean by graph).  One launch. */
GVQA_API int gvqa_graph_segment_mean(const gvqa_graph* g, int64_t F, const float* x, int64_t ld_x, float* out, int64_t ld_out, void* stream);
/* out[i, :F] = sum over the in-edges e of node i of x[e, :F] (x is a per-edge tensor in COO order): scatter_add by
 * destination -- or by SOURCE when `g` is the transposed graph.  The adjoint of the per-edge gathers x[dst] / x[src]
 * (torch's own gather backward is a sort-based index_put).  Deterministic. */
GVQA_API int gvqa_graph_edge_rows_sum(const gvqa_graph* g, int64_t F, const float* x, int64_t ld_x, float* out, int64_t ld_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Building blocks exported for tests, benchmarks and the variants' host code
 * ---------------------------------------------------------------------------------------- */
/* C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) (optionally ReLU) in exact fp32 on the MFMA f32 path.
 * lda/ldb/ldc in elements.  This is torch.nn.Linear's math (F.linear) for the dense
 * projections (gat_skip.py:133,150).  bias may be NULL. */
GVQA_API int gvqa_linear_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                    const float* B, int64_t ldb, const float* bias, int relu,
                    float* C, int64_t ldc, void* stream);
/* Same with the full epilogue: v = acc + bias[n]; v += addend[m*ld_add + n]; v *= mul[m*ld_mul + n];
 * activation (`relu`: 0 none, 1 ReLU, 2 ELU).  addend may alias C (accumulate a second product in place: split-source concatenations,
 * lcgn.py:316-319).  bias / addend / mul may be NULL. */
GVQA_API int gvqa_linear_f32_ex(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       const float* B, int64_t ldb, const float* bias, const float* addend, int64_t ld_add,
                       const float* mul, int64_t ld_mul, int relu, float* C, int64_t ldc, void* stream);

/* bf16 matrix-core projection of a bf16 tensor (the LCGN bf16-node-feature mode's GEMM, exported for tests):
 * gvqa_pack_weight_bf16 writes Wpk[rows, pieces*K] (bf16): piece 0 = bf16(W), piece 1 = bf16(W - piece 0);
 * gvqa_linear_bf16 computes C[M,N] = sum_p A[M,K] . Wpk_p[N,K]^T with fp32 accumulation and the epilogue of
 * gvqa_linear_f32_ex.  A is bf16 [M, lda]; C, addend and mul are bf16 when c_bf16 != 0, fp32 otherwise; bias
 * is fp32.  K and lda must be multiples of 8, A and Wpk 16-byte aligned. */
GVQA_API int gvqa_pack_weight_bf16(int64_t rows, int64_t K, int pieces, const float* W, int64_t ldw, void* Wpk, void* stream);
GVQA_API int gvqa_linear_bf16(int64_t M, int64_t N, int64_t K, int pieces, const void* A, int64_t lda, const void* Wpk,
                     const float* bias, const void* addend, int64_t ld_add, const void* mul, int64_t ld_mul, int relu,
                     void* C, int64_t ldc, int c_bf16, void* stream);

/* fp32-accurate projection on the 16-bit matrix cores (the GAT hop projection xp = lin_l(x_cat), gat_skip.py:133).
 * "split3": every fp32 operand value is the exact sum of three round-to-nearest
 * bf16 pieces; the six largest of the nine piece products (each exact in fp32) are accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16 -- fp32-class accuracy (dropped terms <= 3 * 2^-27 relative) at 16x the f32-MFMA
 * rate for 6x the work.  gvqa_split3_pack writes an fp32 matrix X[rows, K] (leading dimension ld) as
 * fragment-major pieces P[ceil(rows/32)][ceil(K/16)][3][64 lanes][8 bf16] (zero padded; packed must be
 * 16-byte aligned and hold gvqa_split3_packed_bytes(rows, K)); gvqa_linear_split3 computes
 * C[M,N] = A[M,K] . B[N,K]^T from the packed operands with the epilogue of gvqa_linear_f32_ex (fp32 C, N % 4 == 0,
 * 16-byte aligned rows; GVQA_E_UNSUPPORTED otherwise). */
GVQA_API size_t gvqa_split3_packed_bytes(int64_t rows, int64_t K);
GVQA_API int gvqa_split3_pack(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, void* stream);
GVQA_API int gvqa_linear_split3(int64_t M, int64_t N, int64_t K, const void* Apk, const void* Bpk, const float* bias,
                       const float* addend, int64_t ld_add, const float* mul, int64_t ld_mul, int relu, float* C,
                       int64_t ldc, void* stream);
/* "split2h" (the default): every ROW of an operand is scaled by a power of two that puts its largest magnitude into
 * [2^13, 2^14) and every value is carried as two round-to-nearest fp16 pieces p1 + p2 (|x 2^e - p1 - p2| <= 2^-22 |x 2^e|,
 * 2^-38 of the row's largest magnitude for values 2^-16 below it); the three largest piece products (exact in fp32) are
 * accumulated in fp32 by v_mfma_f32_32x32x16_f16 and the accumulators are rescaled exactly.  The split is not exact, but
 * its error is below the fp32 accumulation's own: against fp64 the result is as close as the f32-input MFMA's or split3's
 * (tests/test_gpu_split3.py) at half of split3's matrix-core work and two thirds of its operand bytes.  Same calls and layout
 * with 2 pieces of fp16, followed by one fp32 inverse scale per (padded) row:
 * P[ceil(rows/32)][ceil(K/16)][2][64 lanes][8 fp16] | inv_scale[32 ceil(rows/32)]. */
GVQA_API size_t gvqa_split2h_packed_bytes(int64_t rows, int64_t K);
GVQA_API int gvqa_split2h_pack(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, void* stream);
/* gvqa_split2h_pack that also leaves the operand's largest magnitudes as GVQA_ABSMAX_SLOTS slice maxima in `absmax` (zeroed here): the hint the
 * backward's one-scale products take (gvqa_linear_backward_split2h_hint, gvqa_linear_tn_split2h). */
GVQA_API int gvqa_split2h_pack_absmax(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, float* absmax, void* stream);
/* ... and J = 8 dot products per row on the way, a_node[r, j] = sum_k X[r, k] Vn[j, k] (Vn [J, K] row-major): the node halves of the attention
 * logits with the attention vectors folded through the weights (gat_skip.py:134-135) from the SAME pass over h that packs it for the projection
 * (absmax may be NULL).  K % 4 == 0, K <= 1024 (GVQA_E_UNSUPPORTED otherwise: gvqa_skinny_forward is the general form). */
GVQA_API int gvqa_split2h_pack_logits(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, float* absmax, const float* Vn, int32_t J,
                                      float* a_node, void* stream);
GVQA_API int gvqa_linear_split2h(int64_t M, int64_t N, int64_t K, const void* Apk, const void* Bpk, const float* bias,
                        const float* addend, int64_t ld_add, const float* mul, int64_t ld_mul, int relu, float* C,
                        int64_t ldc, void* stream);

/* The same product as a link of a CHAIN of products (round 5; LCGN's node products, lcgn.py:312-319): the A operand may come as TWO K
 * segments packed by different producers -- Apk [M, K1] and A2pk [M, K2] (NULL: one segment), each with its own row scales, against ONE
 * packed weight image [N, K1 + K2] (K1 a multiple of 16) -- and the finished rows may leave as the NEXT product's packed A operand
 * (pk_out: a buffer of gvqa_split2h_packed_bytes(M, N) bytes, N <= 512, laid out and scaled exactly as gvqa_split2h_pack would; C may
 * then be NULL).  Epilogue order: bias, addend, mul (elementwise), ReLU.  No pack pass over [u | v] or over the result. */
GVQA_API int gvqa_linear_split2h_chain(int64_t M, int64_t N, int64_t K1, const void* Apk, int64_t K2, const void* A2pk, const void* Bpk,
                              const float* bias, const float* addend, int64_t ld_add, const float* mul, int64_t ld_mul, int relu,
                              float* C, int64_t ldc, void* pk_out, void* stream);

/* Tall-skinny products of the differentiable path (training; csrc/train.hip): the attention logits
 * (x_i * att).sum(-1) of gat_skip.py:134-135,151 with the attention vectors folded through the projection weights, a = X V,
 * V [D, J] row-major with J <= 32 (2H node columns, or the H edge columns of all K hops side by side), and the two products
 * of their autograd: dV = X^T G and dX = addend + G V^T.  X [R, D] (row stride ldx), Y / G [R, J] contiguous.  Each call
 * streams X (or dX) through HBM once; dV is summed in a fixed order (partial sums per 128 or 256 rows in the workspace, no atomics).
 * D % 4 == 0, D <= 1024. */
GVQA_API int gvqa_skinny_forward(int64_t R, int64_t D, int64_t J, const float* X, int64_t ldx, const float* V, float* Y, void* stream);
GVQA_API size_t gvqa_skinny_backward_weight_workspace_bytes(int64_t R, int64_t D, int64_t J);
GVQA_API int gvqa_skinny_backward_weight(int64_t R, int64_t D, int64_t J, const float* X, int64_t ldx, const float* G, float* dV,
                                void* ws, size_t ws_bytes, void* stream);
GVQA_API int gvqa_skinny_backward_input(int64_t R, int64_t D, int64_t J, const float* G, const float* V, const float* addend,
                               int64_t ld_add, float* dX, int64_t ldx, void* stream);

/* Attention vectors folded through a projection weight (differentiable path): with W [H*C, Kin] (row stride ldw; lin_l or
 * lin_e, gat_skip.py:133,150) and att [H*C] (att_l / att_r / att_e, :134-135,151), (x W^T * att).sum(-1) = x V with
 *     V[k, h] = sum_c W[h C + c, k] att_a[h C + c],   V[k, H + h] = the same with att_b     (V [Kin, J], J = 2H, or H when att_b is NULL)
 * and the adjoint: dW[r, k] = att_a[r] dV[k, h(r)] + att_b[r] dV[k, H + h(r)] (written, row stride ld_dw; NULL: skipped),
 * datt_a[r] = sum_k W[r, k] dV[k, h(r)] (NULL: skipped), datt_b likewise.  Fixed summation order. */
GVQA_API int gvqa_fold_attention_forward(int64_t H, int64_t C, int64_t Kin, const float* W, int64_t ldw, const float* att_a, const float* att_b,
                                float* V, void* stream);
GVQA_API int gvqa_fold_attention_backward(int64_t H, int64_t C, int64_t Kin, const float* W, int64_t ldw, const float* att_a, const float* att_b,
                                 const float* dV, float* dW, int64_t ld_dw, float* datt_a, float* datt_b, void* stream);

/* C[M, N] = X^T Y for X [R, M], Y [R, N] (row strides ldx / ldy): the weight gradient dW = dy^T x of the hop projection under
 * autograd (torch.nn.Linear's backward at gat_skip.py:133 -- a reduction over all R = N_nodes rows).  Both operands are packed
 * transposed into two-piece fp16 fragments (one power-of-two scale per operand), the split GEMM of gvqa_linear_split2h runs over
 * split-K chunks of rows (gridDim.z), and the partial results are added in a fixed order (no atomics).  x_absmax / y_absmax:
 * device pointers to x_absmax_n / y_absmax_n (<= GVQA_ABSMAX_SLOTS) floats whose maximum is >= max|X| / max|Y| when the
 * producer knows it (gvqa_gat_mp_bwd_desc.dxp_absmax), else NULL (computed here, one extra pass over the operand).
 * M, N, ldx, ldy, ldc multiples of 4. */
GVQA_API size_t gvqa_linear_tn_workspace_bytes(int64_t R, int64_t M, int64_t N);
GVQA_API int gvqa_linear_tn_split2h(int64_t R, int64_t M, int64_t N, const float* X, int64_t ldx, const float* Y, int64_t ldy,
                           const float* x_absmax, int x_absmax_n, const float* y_absmax, int y_absmax_n, float* C, int64_t ldc,
                           void* ws, size_t ws_bytes, void* stream);

/* Backward of the hop projection y = x W^T (torch.nn.Linear at gat_skip.py:133 under autograd) on the two-piece arithmetic:
 * dx [R, K] = dy W and / or dW [M, K] = dy^T x (either may be NULL: skipped) for dy [R, M], W [M, K], x [R, K].  dy is read ONCE:
 * one pass leaves both of its packed forms (rows as rows for dx, rows as the contraction for dW), W^T is packed transposed, the
 * split GEMM runs for dx (dx_accumulate != 0: dx += dy W, e.g. on top of the logit products' input gradient) and -- over split-K
 * chunks -- for dW (fixed-order reduction).  dy_absmax as in gvqa_linear_tn_split2h.
 * M, K and all leading dimensions multiples of 4. */
GVQA_API size_t gvqa_linear_backward_workspace_bytes(int64_t R, int64_t M, int64_t K);
GVQA_API int gvqa_linear_backward_split2h(int64_t R, int64_t M, int64_t K, const float* dy, int64_t ld_dy, const float* W, int64_t ldw,
                                 const float* x, int64_t ldx, const float* dy_absmax, int dy_absmax_n, float* dx, int64_t ld_dx,
                                 int dx_accumulate, float* dW, int64_t ld_dw, void* ws, size_t ws_bytes, void* stream);
/* The same with the largest magnitudes of x known too (x_absmax: 1 .. GVQA_ABSMAX_SLOTS slice maxima, e.g. the by-product of the forward's
 * operand pack, gvqa_split2h_pack_absmax; NULL = measured here): no pass over x before dW. */
GVQA_API int gvqa_linear_backward_split2h_hint(int64_t R, int64_t M, int64_t K, const float* dy, int64_t ld_dy, const float* W, int64_t ldw,
                                               const float* x, int64_t ldx, const float* dy_absmax, int dy_absmax_n, const float* x_absmax,
                                               int x_absmax_n, float* dx, int64_t ld_dx, int dx_accumulate, float* dW, int64_t ld_dw, void* ws,
                                               size_t ws_bytes, void* stream);
/* The general form.  `ex` (may be NULL) carries the optional operands: x_absmax as above, and two terms that ride in the EPILOGUE of the dx product
 * when it reads dy directly (GVQA_OPT_TN_DIRECT, M % 16 == 0, dy 16-byte aligned; GVQA_E_UNSUPPORTED before anything is launched otherwise):
 *     dx = [dx +] dy W + lowrank_g lowrank_v^T + addend
 * -- in the hop, the logit products' input gradient (g = d a_node [R, J], v = the folded attention vectors [K, J]; gvqa_skinny_backward_input's
 * product) and the gradient of the skip connection, so that dh is written once. */
typedef struct gvqa_linear_backward_extras {
    const float* x_absmax;      /* NULL or slice maxima of |x|                                       */
    int32_t x_absmax_n;
    int32_t J;                  /* columns of lowrank_g / lowrank_v: 4, 8, 12 or 16                  */
    const float* lowrank_g;     /* NULL or [R, J], 16-byte aligned                                   */
    const float* lowrank_v;     /* [K, J], 16-byte aligned                                           */
    const float* addend;        /* NULL or [R, ld_addend]                                            */
    int64_t ld_addend;          /* 0 -> K                                                            */
} gvqa_linear_backward_extras;
GVQA_API int gvqa_linear_backward_split2h_ex(int64_t R, int64_t M, int64_t K, const float* dy, int64_t ld_dy, const float* W, int64_t ldw,
                                             const float* x, int64_t ldx, const float* dy_absmax, int dy_absmax_n, float* dx, int64_t ld_dx,
                                             int dx_accumulate, float* dW, int64_t ld_dw, const gvqa_linear_backward_extras* ex, void* ws,
                                             size_t ws_bytes, void* stream);

/* Process-wide run-time options.  Initial values come from the environment (GVQA_PROJ=split2h|split3|f32,
 * GVQA_GEMM_BACKEND=rocblas, GVQA_SPLIT3_MIN_MFLOP, GVQA_SPLIT3_VARIANT); gvqa_set_option overrides them for calls
 * made afterwards (benchmarks and tests compare modes inside one process).  Workspace sizes depend on
 * GVQA_OPT_PROJECTION / _MIN_MFLOP: query gvqa_*_workspace_bytes after changing them. */
enum gvqa_option {
    GVQA_OPT_PROJECTION = 0,       /* arithmetic of the hop projection xp = lin_l(x_cat): GVQA_PROJECTION_* */
    GVQA_OPT_VENDOR_GEMM = 1,      /* 1: plain fp32 products >= 2 GFLOP go to rocBLAS (comparison only; default 0) */
    GVQA_OPT_SPLIT3_MIN_MFLOP = 2, /* products below this many MFLOP stay on the f32-input MFMA kernels (default 1000) */
    GVQA_OPT_SPLIT3_VARIANT = 3,   /* 0 = choose by shape; otherwise an exact k_linear_split3 instantiation (tuning / tests;
                                      < 100: three-piece kernels, >= 100: two-piece kernels) */
    GVQA_OPT_HOP_FUSION = 4,       /* how a gat_seq hop runs when the batch allows fusion (split projection, graphs <= 128 nodes, H in {1,2,4,8}):
                                      0: projection GEMM, then the message-passing kernel (xp through HBM);
                                      1: projection + aggregation + epilogue as ONE 8-wave kernel, one workgroup per CU (csrc/split3.hip);
                                      2: the same as the persistent kernel of csrc/hop2.hip (two 4-wave workgroups per CU, one's aggregation
                                         under the other's matrix-core loop; two-piece operands), hops CHAINED: a hop leaves the next hop's
                                         packed operand, so only the first hop has a pack pass;
                                      3 (default): mode 5's form (plain outputs; mode 4's otherwise) when H = 4, C == node_dim, 256 < C <= 512 and the
                                         batch's row groups -- the PACKED ones when the handle has them, GVQA_OPT_PACKED_GROUPS -- fill the CUs' last round
                                         (two rounds from 0.76 full, three from 0.82, else 0.85); else with H = 4 and >= 128 row groups, 1 with the hops chained (the 8-wave kernel writes the next
                                         hop's packed operand too: the faster of the two chained forms, round 4); otherwise 2 when the batch has
                                         >= 6 (row group, column block) items per workgroup slot, else 1;
                                      4: the aggregate-first kernel of csrc/hopagg.hip (H = 4, C == node_dim <= 512: heads concatenated along K,
                                         the attention-weighted neighbour sum formed inside the matrix-core loop, rows chunk-major between hops);
                                         falls back to 1 where it does not apply.  Modes 1 and 2 chain whenever the batch allows;
                                      5: mode 4 with the K hops as ONE launch: a workgroup owns all output columns of its row group (whole
                                         graphs), so hop i + 1 of its rows needs nothing from another workgroup -- it runs the coefficient
                                         phase (node logits, leaky-relu, segment softmax) itself between two hops; rows travel chunk-major
                                         through L2 / HBM.  Plain outputs only (attention weights / per-hop rows: mode 4's launches);
                                      6: the aggregate-first kernel with a row group's output columns split over four workgroups (128 x 128 tiles,
                                         per-hop launches; 384 < C <= 512): built for small batches / strong-scaling shards, where one workgroup per
                                         row group leaves most CUs idle; parity-green, measured SLOWER than mode 1 there (0.54 vs 0.43 ms on a 256-graph
                                         shard of config 3), so the default rule does not take it. */
    GVQA_OPT_COEFF_KERNEL = 5,     /* attention coefficients: 0 (default) the row-group kernel when a row-group plan exists, 1 always the
                                      per-(node, head) kernel (same operations in the same order: bit-identical; tests) */
    GVQA_OPT_MP_PARTS = 6,         /* stand-alone message-passing kernel: 0 (default) blocks per graph chosen by batch size, n > 0 exactly n */
    GVQA_OPT_HOP_COEFFS = 7,       /* chained hops on the 8-wave kernel: 1 = attention coefficients computed INSIDE the hop kernel -- partial node logits
                                      left by the previous hop's column blocks, edge halves gathered through the CSR edge ids, leaky-relu + segment
                                      softmax in LDS: one launch per hop, no coefficient kernel, no pack pass after hop 0 (row groups within 522 edges
                                      at H = 4).  0 = the coefficient kernels of rounds 3 / 4.  2 (default since round 6) = 1 for batches of fewer than 128 row
                                      groups (where a hop's launches are latency-bound: 256-graph shard 0.413 -> 0.394 ms), 0 above.  Built and parity-green in round 5, and
                                      measured a wash: the phase costs the hop kernel what the two small launches it replaces cost (256-graph shard
                                      0.409 vs 0.414 ms, config 2 0.723 vs 0.705 ms per forward; profiles/r05_hop_coeffs_ab.txt) */
    GVQA_OPT_HOP_HALF_TILES = 8,   /* 8-wave fused hop: 1 (default) the row blocks of a launch's last PARTIAL round of workgroups take one row group each (128-row
                                      half tiles, the empty half's waves skip their products) when that shortens the launch -- config 2: 585 workgroups
                                      = three rounds on 256 CUs for 2.29 rounds of work -> 510 + 145 half tiles; 0 = every block two row groups */
    GVQA_OPT_TN_DIRECT = 9,        /* weight gradient dW = dy^T x (gvqa_linear_tn_split2h, gvqa_linear_backward_split2h): 1 (default) the product reads the
                                      row-major fp32 operands itself and transposes them on the way into the MFMA fragment image (tn_direct.hip); 0 = both
                                      operands packed transposed in HBM first (round 3's form; same scales, pieces and chunks) */
    GVQA_OPT_PACKED_GROUPS = 10,   /* packed row groups for the aggregate-first hops (gvqa_graph::pk_*): 1 (default) built at finalize when re-ordering the
                                      graphs saves a round of workgroups on this device, and then used by those hops; 0 never; 2 built whenever it saves a
                                      row group at all (tests).  Read at gvqa_graph_finalize* / gvqa_graph_build_grouped time. */
    GVQA_NUM_OPTIONS = 11
};
#define GVQA_PROJECTION_SPLIT3 0   /* three exact bf16 pieces per fp32 value, six bf16-MFMA products, fp32 accumulate */
#define GVQA_PROJECTION_F32 1      /* f32-input MFMA (k_linear_f32*) */
#define GVQA_PROJECTION_SPLIT2H 2  /* two scaled fp16 pieces per fp32 value, three fp16-MFMA products, fp32 accumulate (default) */
GVQA_API int gvqa_set_option(int option, int value);
GVQA_API int gvqa_get_option(int option);

/* Scene-graph collate on the host (/root/reference gqa_dataset_entry.py:190-372 converter rules + :631-675 / :654
 * Batch.from_data_list offsets) over PRE-TOKENISED, flattened scene graphs -- the loader's path into the batch without a Python
 * loop over nodes and edges.  Per batch of B graphs, objects of a graph in the converter's node order (object ids sorted as
 * strings): graph_obj_ptr[B+1]; name_tok[O]; attr_ptr[O+1] / attr_tok (tokens of the DISTINCT attribute strings, first
 * occurrence order, at most 11); rel_ptr[O+1] / rel_dst (destination object as a local index inside its graph) / rel_tok.  A graph
 * without objects becomes the converter's two-node dummy graph (all tokens unk_tok).  Host pointers throughout.
 *   gvqa_scene_graph_collate_sizes -> sizes[3] = nodes N, edges E (self-loops, relations, added reverse edges), added reverse edges A
 *   gvqa_scene_graph_collate       -> x_tokens [N,12], edge_index [2,E] (row 0 sources, row 1 destinations), edge_tokens [E],
 *                                     added_sym_edge [A] (edge ids), batch [N], graph_ptr / edge_ptr [B+1] (the layout
 *                                     gvqa_graph_finalize_host takes), *max_in_degree.
 * GVQA_E_GRAPH: a relation points outside its graph; GVQA_E_INVALID: more than 11 attributes (the reference raises IndexError). */
GVQA_API int gvqa_scene_graph_collate_sizes(int64_t num_graphs, const int32_t* graph_obj_ptr, const int32_t* rel_ptr, const int32_t* rel_dst,
                                   int64_t* sizes);
GVQA_API int gvqa_scene_graph_collate(int64_t num_graphs, const int32_t* graph_obj_ptr, const int64_t* name_tok, const int32_t* attr_ptr,
                             const int64_t* attr_tok, const int32_t* rel_ptr, const int32_t* rel_dst, const int64_t* rel_tok,
                             int64_t pad_tok, int64_t self_tok, int64_t unk_tok, int64_t N, int64_t E, int64_t A, int64_t* x_tokens,
                             int64_t* edge_index, int64_t* edge_tokens, int64_t* added_sym_edge, int64_t* batch, int32_t* graph_ptr,
                             int32_t* edge_ptr, int32_t* max_in_degree);

/* Resident workgroups per CU of the persistent hop kernel (csrc/hop2.hip) as the HIP runtime reports them for head count H
 * in {1,2,4,8}: 2 is what its design needs (80 KiB of LDS, <= 256 VGPRs); < 0 = GVQA_E_*.  Diagnostics / tests. */
GVQA_API int gvqa_hop2_blocks_per_cu(int32_t H);

/* Measurement utility -- the "device copy" denominator of SURVEY 8(d) ("report HBM fractions of spec and of measured copy"):
 * copies `bytes` (a multiple of 16; both pointers 16-byte aligned) from src to dst with 16-byte accesses, `variant` 0 = plain
 * grid-stride float4 loads / stores (the form MI355X_MICROARCH.md quotes 6.29 TB/s for), 1 = the same with non-temporal stores,
 * 2 = through LDS by LDS-DMA (global_load_lds_dwordx4) and ds_read / global stores, the streaming structure of the
 * message-passing kernel.  bench.py times all three beside torch's own copy kernel and reports the best as `hbm_copy_measured`.
 * No counterpart in the reference. */
GVQA_API int gvqa_stream_copy(void* dst, const void* src, size_t bytes, int variant, void* stream);

/* Measurement entry point: a loop of nothing but v_mfma_f32_32x32x16_f16 (bf16 != 0: _bf16) on fragments taken from `operands` (any bit
 * patterns: what they hold decides the power the matrix pipes draw, hence the clock), one workgroup of eight waves per CU, 64 MFMAs per wave and
 * iteration.  sink: >= CUs x 512 floats (keeps the accumulators live).  *flops_out (host, may be NULL) = the launch's flop count.  bench.py times
 * it on random and on zero operands and reports the rates as `matrix_rate_measured` beside the data sheet's dense peak.  Bits 1-2 of `bf16`
 * pick the order in which a step's 16 products are issued (0: as the GEMM kernels do; 1, 2: operand-reuse experiments, see the kernel).
 * No counterpart in the reference. */
GVQA_API int gvqa_mfma_stream(const void* operands, size_t operand_bytes, float* sink, size_t sink_elems, int iters, int bf16, int64_t* flops_out,
                              void* stream);

/* Which GEMM backend serves the plain dense projections in this process (hand-written k_linear_f32,
 * or rocBLAS for large epilogue-free products; GVQA_GEMM_BACKEND=auto|hip|rocblas). */
GVQA_API const char* gvqa_gemm_backend(void);

/* The fused GAT message-passing kernel on its own (SURVEY 2.1 K4-K9 + K11): attention logits
 * -> leaky-relu -> softmax over incoming edges -> alpha-weighted sum of projected source
 * features -> head mean -> (x graph scale) (+graph term) + bias + skip -> (BN -> ReLU).
 * Zero-initialise the descriptor; leading dimensions of 0 mean "dense". */
typedef struct gvqa_gat_mp_desc {
    int32_t C, H;               /* out channels, heads                                               */
    float negative_slope, bn_eps;
    const float* xp;            /* [N, xp_ld] projected node features, head h at columns [h*C,(h+1)*C) */
    int64_t xp_ld;              /* 0 -> H*C                                                          */
    const float* a_node;        /* [N, 2H] (a_l | a_r) per node, or NULL (zeros)                     */
    const float* a_edge;        /* logit term of COO edge e, head h at a_edge[e*a_edge_stride + h]   */
    int64_t a_edge_stride;      /* 0 -> H                                                            */
    const float* graph_term;    /* NULL or [B, graph_term_ld]: columns [0,C) head-mean instruction
                                   projection, [C,C+H) logit offset; ld >= C+H, multiple of 4        */
    int64_t graph_term_ld;
    const float* graph_scale;   /* NULL or [B, graph_scale_ld]: per-graph channel scale of the head
                                   mean, applied before graph term / bias (LCGN cal_cmd, lcgn.py:231) */
    int64_t graph_scale_ld;     /* 0 -> C                                                            */
    const float* skip;          /* NULL or [N, skip_ld]                                              */
    int64_t skip_ld;            /* 0 -> C                                                            */
    const float* bias;          /* NULL or [C]                                                       */
    const float* bn_weight;     /* eval BatchNorm + ReLU after the skip; all four NULL = none        */
    const float* bn_bias;
    const float* bn_mean;
    const float* bn_var;
    float* out;                 /* [N, out_ld]                                                       */
    int64_t out_ld;             /* 0 -> C                                                            */
    float* alpha_out;           /* NULL or [E, H] in COO edge order (softmax output, before alpha_mask) */
    const float* alpha_mask;    /* NULL or [E, H] in COO edge order: multiplies alpha after the softmax
                                   (attention dropout, gat_skip.py:205: mask / (1 - p))              */
    int32_t force;              /* 0 = auto, 1 = LDS-tiled kernel, 2 = general CSR kernel            */
    const float* head_rows;     /* NULL or [B, head_rows_ld]: rows added to xp per GRAPH and head (the instruction half of lin_l on
                                   [h | ins[batch]], gat_skip.py:133,263-264) without forming the [N, H*C] sum:
                                   out[i] += (1/H) sum_h s[i,h] head_rows[graph(i), h*C .. h*C+C), s[i,h] = sum over the in-edges
                                   of alpha * alpha_mask (1 without a mask, 0 for a node without in-edges).  LDS-tiled kernel only
                                   (GVQA_E_UNSUPPORTED otherwise: gvqa_graph_head_rows_add is the general form); excludes
                                   graph_term / graph_scale */
    int64_t head_rows_ld;       /* 0 -> H*C                                                          */
    float* head_weight_out;     /* NULL or [N, H]: s (what gvqa_graph_head_rows_backward needs)      */
} gvqa_gat_mp_desc;
/* ws: >= 4*E*H bytes, used by the general kernel only. */
GVQA_API int gvqa_gat_message_passing(const gvqa_graph* g, const gvqa_gat_mp_desc* d, void* ws, size_t ws_bytes, void* stream);

/* Backward of gvqa_gat_message_passing in its bare form (no graph terms / scale / bias / skip / BN:
 * out[i] = (1/H) sum_h sum_{e->i} alpha[e,h] mask[e,h] xp[src_e,h,:]) -- what autograd does through PyG's
 * gather / utils.softmax / scatter_add (gat_skip.py:155,183-208) and the head mean (:162-165); the "next"
 * row SURVEY 8f-4.  `g` is the forward graph, `gt` the TRANSPOSED one: gvqa_graph_build + finalize on the
 * flipped edge_index (row 0 <-> row 1) of the same batch.  Deterministic (no atomics). */
#define GVQA_ABSMAX_SLOTS 256      /* slices of a largest-magnitude reduction (one atomic address each) */
typedef struct gvqa_gat_mp_bwd_desc {
    int32_t C, H;
    float negative_slope;
    const float* xp;            /* [N, xp_ld] as given to the forward                                   */
    int64_t xp_ld;              /* 0 -> H*C                                                              */
    const float* a_node;        /* [N, 2H] or NULL, as given to the forward                              */
    const float* a_edge;        /* as given to the forward                                               */
    int64_t a_edge_stride;      /* 0 -> H                                                                */
    const float* alpha;         /* [E, H] COO: alpha_out of the forward                                  */
    const float* alpha_mask;    /* NULL or [E, H] COO, as given to the forward                           */
    const float* dout;          /* [N, dout_ld]: gradient of the loss w.r.t. out                         */
    int64_t dout_ld;            /* 0 -> C                                                                */
    float* dxp;                 /* [N, dxp_ld]  (written, not accumulated)                               */
    int64_t dxp_ld;             /* 0 -> H*C                                                              */
    float* da_node;             /* [N, 2H]                                                               */
    float* da_edge;             /* [E, H] COO                                                            */
    const float* dalpha_node;   /* NULL or [N, H]: a term added to dL/d(alpha[e,h]) of every in-edge e of node i, before the
                                   mask -- the gradient of s[i,h] = sum_{e->i} alpha mask when per-graph rows of the projection
                                   are kept out of xp (gvqa_graph_head_rows_add / _backward below)          */
    float* dxp_absmax;          /* NULL or [GVQA_ABSMAX_SLOTS] floats (written): the largest |dxp| in slices -- the operand scale
                                   gvqa_linear_tn_split2h takes, without a pass over dxp                     */
} gvqa_gat_mp_bwd_desc;
GVQA_API int gvqa_gat_mp_backward(const gvqa_graph* g, const gvqa_graph* gt, const gvqa_gat_mp_bwd_desc* d, void* stream);

/* Per-graph rows of the hop projection kept out of xp (differentiable path).  lin_l acts on [h | ins[batch]] (gat_skip.py:133,
 * 263-264): xp[i,h,:] = xp_node[i,h,:] + R[g(i),h,:] with R = ins W_i^T one row per GRAPH.  The aggregation is linear in xp, so
 *     out[i,:] = MP(xp_node)[i,:] + (1/H) sum_h s[i,h] R[g(i),h,:],   s[i,h] = sum_{e->i} alpha[e,h] mask[e,h]   (without a mask: 1, or 0 for a node with no in-edge)
 * and the [N, H*C] sum xp_node + R[batch] (and its adjoint, a segment sum over [N, H*C]) is never formed:
 *   gvqa_graph_head_rows_add       y[i,:] += (1/H) sum_h s[i,h] R[g(i),h,:] + bias[:] + skip[i,:]   (s NULL: the mask-free values;
 *                                  R / bias / skip NULL: term absent -- bias gat_skip.py:167-168 and the skip connection :270 ride along)
 *   gvqa_graph_head_rows_backward  dR[g,h,:] = (1/H) sum_{i in g} s[i,h] dy[i,:];  ds[i,h] = (1/H) dy[i,:] . R[g(i),h,:];
 *                                  dcol[g,:] = sum_{i in g} dy[i,:] (the bias gradient per graph)   (each output NULL: not computed;
 *                                  ds is what gvqa_gat_mp_bwd_desc.dalpha_node takes)
 * R / dR [B, H*C] contiguous, dcol [B, C], y / dy / skip [N, C] with row strides, s / ds [N, H] contiguous.  C % 4 == 0, H <= 8. */
GVQA_API int gvqa_graph_head_rows_add(const gvqa_graph* g, int64_t C, int64_t H, const float* R, const float* s, const float* bias,
                             const float* skip, int64_t ld_skip, float* y, int64_t ld_y, void* stream);
GVQA_API int gvqa_graph_head_rows_backward(const gvqa_graph* g, int64_t C, int64_t H, const float* dy, int64_t ld_dy, const float* R,
                                  const float* s, float* dR, float* ds, float* dcol, void* stream);

/* Host-only introspection: the geometry gvqa_gat_message_passing would use for this (finalized)
 * graph -- LDS-tiled streaming kernel or general CSR kernels -- without launching anything. */
typedef struct gvqa_mp_plan {
    int32_t tiled;             /* 1 = k_gat_mp_tiled, 0 = general CSR kernels                              */
    int32_t channel_range;     /* cw: channels per stage (row segments of 4*cw contiguous bytes)             */
    int32_t stage_buffers;     /* LDS stage buffers (prefetch depth + 1)                                     */
    int32_t blocks_per_cu;     /* graphs resident per CU by LDS                                              */
    int32_t stages_per_graph;  /* DMA stages per graph incl. the skip-row stage of every channel range       */
    int32_t accumulators;      /* float4 accumulators per thread                                             */
    int64_t lds_bytes;         /* dynamic LDS per block                                                      */
    int32_t blocks_per_graph;  /* > 1 for small batches: the channel ranges of a graph are split over several blocks */
} gvqa_mp_plan;
GVQA_API int gvqa_gat_mp_plan(const gvqa_graph* g, int32_t C, int32_t H, gvqa_mp_plan* out);

/* ------------------------------------------------------------------------------------------
 * GINE / GCN variants (baseline_and_test_models/pipeline_model_{gine,gcn}.py:622-674)
 * ---------------------------------------------------------------------------------------- */
typedef struct gvqa_bn_params {     /* one eval-mode BatchNorm1d: bns.j.{weight,bias,running_mean,running_var} */
    const float* weight;
    const float* bias;
    const float* mean;
    const float* var;
} gvqa_bn_params;

/* out = relu(bn_{S-1}(... relu(bn_0(x)) ...)): what gine_seq.forward / gcn_seq.forward return as
 * written -- they compute conv_res and discard it (pipeline_model_gine.py:665-671,
 * pipeline_model_gcn.py:660-666).  x, out [N, C]; `stages` is a host array. */
GVQA_API int gvqa_bn_relu_chain(int64_t N, int32_t C, int32_t num_stages, const gvqa_bn_params* stages, float bn_eps,
                       const float* x, float* out, void* stream);

typedef struct gvqa_gine_params {   /* GINEConv(Seq(Lin, ReLU, Lin)): convs.i.nn.{0,2}.{weight,bias}, convs.i.eps */
    const float* nn0_weight;        /* [C, node_dim + ins_dim] */
    const float* nn0_bias;          /* [C] */
    const float* nn2_weight;        /* [C, C] */
    const float* nn2_bias;          /* [C] */
    float eps;
} gvqa_gine_params;

/* PyG GINEConv on x = [h || ins[batch]], e = [edge_attr || ins[batch[src]]] (both node_dim+ins_dim
 * wide, pipeline_model_gine.py:651-665): out = nn((1+eps) x_i + sum_{j->i} relu(x_j + e_ji)).
 * h [N, node_dim], edge_attr [E, node_dim] (COO order), ins [B, ins_dim] (NULL when ins_dim == 0:
 * h / edge_attr are then the full inputs), out [N, C].  ins_dim > 0 needs an intra-graph batch.
 * C <= 320 (the reference: 300) with 16-byte aligned rows and >= 1024 nodes: nn runs as ONE kernel (csrc/gine_mlp.hip -- the hidden
 * rows stay in registers between the two Linears); other shapes: two products with the ReLU pass between them. */
GVQA_API size_t gvqa_gine_conv_workspace_bytes(const gvqa_graph* g, int32_t node_dim, int32_t ins_dim, int32_t C);
GVQA_API int gvqa_gine_conv_forward(const gvqa_graph* g, int32_t node_dim, int32_t ins_dim, int32_t C,
                           const gvqa_gine_params* p, const float* h, const float* edge_attr, const float* ins,
                           float* out, void* ws, size_t ws_bytes, void* stream);

typedef struct gvqa_gcn_params {    /* PyG 1.6/1.7 GCNConv: convs.i.weight [in, out], convs.i.bias [out] */
    const float* weight;            /* [node_dim + ins_dim, C] */
    const float* bias;              /* [C] or NULL */
} gvqa_gcn_params;

/* PyG GCNConv on x = [h || ins[batch]] (pipeline_model_gcn.py:651-660): unit edge weights,
 * add_remaining_self_loops (existing self loops collapse into one per node), symmetric
 * normalisation.  h [N, node_dim], ins [B, ins_dim] or NULL, out [N, C].  Any graph. */
GVQA_API size_t gvqa_gcn_conv_workspace_bytes(const gvqa_graph* g, int32_t node_dim, int32_t ins_dim, int32_t C);
GVQA_API int gvqa_gcn_conv_forward(const gvqa_graph* g, int32_t node_dim, int32_t ins_dim, int32_t C,
                          const gvqa_gcn_params* p, const float* h, const float* ins, float* out,
                          void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * LCGN variant (baseline_and_test_models/lcgn.py)
 * ---------------------------------------------------------------------------------------- */
typedef struct gvqa_lcgn_dims {
    int32_t in_channels;    /* width of x (300)                                                   */
    int32_t out_channels;   /* O (512): node state width = command width = lstm output width      */
    int32_t question_dim;   /* width of q_encoding (512)                                          */
    int32_t num_iters;      /* MAX_ITER_NUM (4), <= 8                                             */
    int32_t seq_len;        /* L: number of lstm_outputs steps                                    */
    int32_t heads;          /* gat_heads (1)                                                      */
    float negative_slope;
    int32_t node_bf16;      /* 0: fp32.  1 / 2: per-node tensors (x_loc, x_ctx, their projections, messages) are
                               STORED as bf16 in HBM (BASELINE config 5) and the node GEMMs run on the bf16
                               matrix cores with fp32 accumulation, against the fp32 weights split into two
                               bf16 pieces (1: weights keep 16 significant bits) or rounded to one (2);
                               logits, softmax, aggregation and the per-graph command path stay fp32        */
} gvqa_lcgn_dims;

typedef struct gvqa_lcgn_params {   /* lcgn_seq state_dict (lcgn.py:255-282), device pointers */
    const float* init_weight;          /* init_sg_emb_input.0.weight [O, in]  */
    const float* init_bias;            /* init_sg_emb_input.0.bias   [O]      */
    const float* qinput1_weight;       /* qInput1.weight [O, Q]               */
    const float* qinput1_bias;
    const float* qinput2_weight[8];    /* qInput2_t.weight [O, O]             */
    const float* qinput2_bias[8];
    const float* cmd_logit_weight;     /* cmd_inter2logits.weight [1, O]      */
    const float* cmd_logit_bias;       /* cmd_inter2logits.bias [1]           */
    const float* proj_x_loc_weight;    /* proj_x_loc.1.weight [O, O]          */
    const float* proj_x_loc_bias;
    const float* proj_x_ctx_weight;    /* proj_x_ctx.1.weight [O, O]          */
    const float* proj_x_ctx_bias;
    const float* output_weight;        /* output_layer.weight [O, 2O]         */
    const float* output_bias;
    const float* fin_weight;           /* fin_layer.weight [O, 2O]            */
    const float* fin_bias;
    const float* lin_l_weight;         /* lcgn.lin_l.weight  [O, 3O]          */
    const float* lin_r_weight;         /* lcgn.lin_r.weight  [O, 3O]          */
    const float* cal_x_weight;         /* lcgn.cal_x.weight  [O, 3O]          */
    const float* proj_cmd_weight;      /* lcgn.proj_cmd.weight [O, O]         */
    const float* cal_cmd_weight;       /* lcgn.cal_cmd.weight  [O, O]         */
    const float* bias;                 /* lcgn.bias [O] or NULL               */
    const void* packed;                /* NULL, or the output of gvqa_lcgn_pack_weights for these weights
                                          and dims (skips the per-call repacking of the weights)        */
    size_t packed_bytes;
} gvqa_lcgn_params;

/* Call-invariant weight forms of lcgn_seq (stacked [lin_l; lin_r; cal_x] blocks, stacked qInput2 /
 * proj_cmd / cal_cmd, and in the bf16 modes the bf16 pieces of the node-GEMM weights).  A caller whose
 * weights do not change between forwards builds them once and passes them in params->packed; they must be
 * rebuilt when any weight, or dims->{in,out}_channels / num_iters / node_bf16, changes. */
GVQA_API size_t gvqa_lcgn_pack_bytes(const gvqa_lcgn_dims* d);
GVQA_API int gvqa_lcgn_pack_weights(const gvqa_lcgn_dims* d, const gvqa_lcgn_params* p, void* packed, size_t packed_bytes,
                           void* stream);

/* lcgn_seq.forward(x, edge_index, batch, q_encoding, lstm_outputs) in eval mode (lcgn.py:303-323).
 * x [N, in], q_encoding [B, Q], lstm_outputs [L, B, O], x_ctx_init [N, O] = the noise the reference
 * draws with torch.randn on the CPU generator (lcgn.py:306; the caller draws it the same way),
 * out [N, O].  Needs a finalized intra-graph batch. */
GVQA_API size_t gvqa_lcgn_seq_workspace_bytes(const gvqa_graph* g, const gvqa_lcgn_dims* d);
GVQA_API int gvqa_lcgn_seq_forward(const gvqa_graph* g, const gvqa_lcgn_dims* d, const gvqa_lcgn_params* p,
                          const float* x, const float* q_encoding, const float* lstm_outputs,
                          const float* x_ctx_init, float* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Step after the path ("next" row): language-conditioned global attention pooling and
 * the short-answer classifier (pipeline_model_gat.py:108-185, :722-728, :800-816)
 * ---------------------------------------------------------------------------------------- */
typedef struct gvqa_pool_params {     /* MyConditionalGlobalAttention: {node,ques,gate}_nn.{0,2}.{weight,bias} */
    const float* node0_weight;  /* [Ch, node_dim] */
    const float* node0_bias;
    const float* node2_weight;  /* [Ch, Ch] */
    const float* node2_bias;
    const float* ques0_weight;  /* [Ch, Ch] */
    const float* ques0_bias;
    const float* ques2_weight;  /* [Ch, Ch] */
    const float* ques2_bias;
    const float* gate0_weight;  /* [Ch, Ch] */
    const float* gate0_bias;
    const float* gate2_weight;  /* [1, Ch] */
    const float* gate2_bias;    /* [1] */
} gvqa_pool_params;

/* x' = node_nn(x); gate = gate_nn(ques_nn(u)[batch] * x'); softmax over the nodes of each graph;
 * out[g] = sum_n gate[n] x'[n].  x [N, node_dim], u [B, Ch], out [B, Ch]. */
GVQA_API size_t gvqa_attention_pool_workspace_bytes(const gvqa_graph* g, int32_t node_dim, int32_t channels);
GVQA_API int gvqa_attention_pool_forward(const gvqa_graph* g, int32_t node_dim, int32_t channels, const gvqa_pool_params* p,
                                const float* x, const float* u, float* out, void* ws, size_t ws_bytes, void* stream);

typedef struct gvqa_classifier_params {   /* logit_fc.{1,4}.{weight,bias} */
    const float* fc1_weight;    /* [hidden, 3Q] */
    const float* fc1_bias;
    const float* fc2_weight;    /* [A, hidden] */
    const float* fc2_bias;
} gvqa_classifier_params;

/* logits = fc2(ELU(fc1([g || q || g*q])))  (eval: dropouts inactive).  g_feat, q [B, Q]; logits [B, A]. */
GVQA_API size_t gvqa_answer_logits_workspace_bytes(int64_t B, int32_t Q, int32_t hidden);
GVQA_API int gvqa_answer_logits_forward(int64_t B, int32_t Q, int32_t hidden, int32_t A, const gvqa_classifier_params* p,
                               const float* g_feat, const float* q, float* logits, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Step before the path ("next" row): ground-truth scene-graph encoder
 * (pipeline_model_gat.py:63-101, 553-610; graph_utils/my_graph_layernorm.py:52-78)
 * ---------------------------------------------------------------------------------------- */
typedef struct gvqa_encoder_params {
    const float* embedding;        /* sg_vocab_embedding.weight [V, D]                                         */
    const float* edge0_weight;     /* scene_graph_encoding_layer.edge_model.edge_mlp.0.weight [D, 3D]          */
    const float* edge0_bias;
    const float* edge2_weight;     /* ...edge_mlp.2.weight [D, D]                                              */
    const float* edge2_bias;
    const float* node1_0_weight;   /* ...node_model.node_mlp_1.0.weight [D, 2D]                                */
    const float* node1_0_bias;
    const float* node1_2_weight;   /* ...node_mlp_1.2.weight [D, D]                                            */
    const float* node1_2_bias;
    const float* node2_0_weight;   /* ...node_model.node_mlp_2.0.weight [D, 2D]                                */
    const float* node2_0_bias;
    const float* node2_2_weight;   /* ...node_mlp_2.2.weight [D, D]                                            */
    const float* node2_2_bias;
    const float* ln_weight;        /* graph_layer_norm.weight [1] or NULL                                      */
    const float* ln_bias;          /* graph_layer_norm.bias   [1] or NULL                                      */
    const void* packed;            /* NULL, or the output of gvqa_sg_encoder_pack_weights for these weights, V and D:
                                      skips the per-call weight-only work of the large-batch path (projected table, stacked /
                                      folded weights, packed weight operands)                                   */
    size_t packed_bytes;
} gvqa_encoder_params;

/* Call-invariant weight forms of the encoder's large-batch path: emb W_e^T (the edge block of EdgeModel's first Linear applied to
 * the embedding table, pipeline_model_gat.py:65-76), the four per-node column blocks stacked, the folded products W_n1e W_e2 and
 * W_n2agg W_n12 with their bias vectors, and the two-piece packed images of every weight operand.  A caller whose weights do not
 * change between forwards builds them once (256-byte aligned buffer of gvqa_sg_encoder_pack_bytes(V, D) bytes) and passes them in
 * params->packed; rebuild when any weight changes.  GVQA_E_UNSUPPORTED when the large-batch path cannot take these weights
 * (D % 4 != 0, unaligned vectors, GVQA_OPT_PROJECTION = f32): call forward without `packed` then. */
GVQA_API size_t gvqa_sg_encoder_pack_bytes(int32_t V, int32_t D);
GVQA_API int gvqa_sg_encoder_pack_weights(int32_t V, int32_t D, const gvqa_encoder_params* p, void* packed, size_t packed_bytes, void* stream);

/* x_tokens int64 [N, node_tokens], edge_tokens int64 [E, edge_tokens_per_edge] (COO order),
 * added_sym_edge int64 [num_added] (indices of edges whose embedding is negated), edge_index int64
 * [2, E] (the same COO the graph was built from).  Outputs x_encoded [N, D], edge_attr_encoded [E, D]
 * -- exactly the (x, edge_attr) the execution path consumes (pipeline_model_gat.py:751, 791). */
/* out[r, :] = (negate && negate[r] ? -1 : +1) * sum_t table[tokens[r, t], :]  -- the token-embedding sums of the scene-graph encoder
 * (pipeline_model_gat.py:583-593; `negate`: one byte per row, the rows of `added_sym_edge`, :590), on any [V, D] table (the embedding
 * itself, or its projection through the edge block of EdgeModel's first Linear).  Token ids are clamped to the table. */
GVQA_API int gvqa_embed_sum(int64_t rows, int32_t T, int32_t V, int32_t D, const int64_t* tokens, const float* table, const uint8_t* negate,
                   float* out, void* stream);
/* y_out[e, :] = relu(a[ia[e], :] + b[ib[e], :] + y_in[e, :] + bias): the first Linear of an edge-level MLP of the encoder once its
 * node-side column blocks are projected per node (pipeline_model_gat.py:65-76,92-95): per-edge gathers, sum, bias and ReLU in one
 * pass.  a / b may be NULL (term absent) and have row strides lda / ldb (column blocks of one wider per-node product); y_out may be y_in. */
GVQA_API int gvqa_gather_add_relu(int64_t E, int32_t D, const float* a, int64_t lda, const int64_t* ia, const float* b, int64_t ldb, const int64_t* ib,
                         const float* bias, const float* y_in, float* y_out, void* stream);
GVQA_API size_t gvqa_sg_encoder_workspace_bytes(const gvqa_graph* g, int32_t D);
GVQA_API int gvqa_sg_encoder_forward(const gvqa_graph* g, int32_t V, int32_t D, int32_t node_tokens, int32_t edge_tokens_per_edge,
                            const gvqa_encoder_params* p, const int64_t* x_tokens, const int64_t* edge_tokens,
                            const int64_t* added_sym_edge, int64_t num_added, const int64_t* edge_index, float ln_eps,
                            float* x_encoded, float* edge_attr_encoded, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * In-library stage timing (HIP events recorded on the caller's stream around each stage).
 * Used by bench.py to obtain the message-passing kernel's launch duration inside the timed
 * region.  Off by default; costs two hipEventRecord per stage when on.
 * ---------------------------------------------------------------------------------------- */
enum {
    GVQA_STAGE_GRAPH = 0,      /* CSR build                                   */
    GVQA_STAGE_FOLD = 1,       /* weight folding (attention vectors, head mean) */
    GVQA_STAGE_EDGE_LOGIT = 2, /* a_e for all hops (skinny GEMM over edge_attr) */
    GVQA_STAGE_GRAPH_TERM = 3, /* per-graph instruction terms                  */
    GVQA_STAGE_PROJ = 4,       /* dense node projection (MFMA GEMM)            */
    GVQA_STAGE_NODE_LOGIT = 5, /* a_l / a_r                                    */
    GVQA_STAGE_MP = 6,         /* fused GAT message passing                    */
    GVQA_STAGE_OTHER = 7,
    GVQA_STAGE_PACK = 8,       /* split3 operand packing (pieces of h, of the weights) */
    GVQA_STAGE_ALPHA = 9,      /* attention coefficients as a kernel of their own (fused-hop path) */
    GVQA_NUM_STAGES = 10
};
/* on = 0: off.  on = 1: every stage.  Otherwise bit 0 set and bits 1.. = a stage mask (bit 1 + s selects GVQA_STAGE_s): only
 * those stages record events -- an event pair costs the stream ~3 us, and bench.py keeps only the dominant kernel's stage on
 * inside its timed region (e.g. 1 | (1 << (1 + GVQA_STAGE_PROJ))). */
GVQA_API int gvqa_prof_enable(int on);
/* Waits for outstanding events, ADDS elapsed milliseconds / launch counts per stage into the
 * arrays (each GVQA_NUM_STAGES long) and clears the internal list. */
GVQA_API int gvqa_prof_collect(double* ms_by_stage, int64_t* launches_by_stage);

#ifdef __cplusplus
}
#endif
#endif /* GVQA_H */
