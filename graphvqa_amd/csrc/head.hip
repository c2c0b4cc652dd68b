// Step right after the execution path ("next" row): language-conditioned global
// attention pooling and the short-answer classifier.
//
// Reference being replaced: MyConditionalGlobalAttention.forward (pipeline_model_gat.py:149-181;
// PyG softmax over `batch` + torch_scatter.scatter_add, K13 in SURVEY 2.1) and logit_fc fed with
// [g || q || g*q] (pipeline_model_gat.py:722-728, 814-816).  Per-graph segment softmax reuses the
// batch structure (graph_ptr) of the graph container; the dense layers run on k_linear_f32.
#include "common.h"

namespace gvqa {

// prod[n, c] = qn[g(n), c] * xn[n, c]                       (ques_nn(u)[batch] * x, :165)
__global__ __launch_bounds__(256) void k_scale_rows_by_graph(int64_t N, int C, const int32_t* __restrict__ node_graph,
                                                             const float* __restrict__ qn, const float* __restrict__ xn,
                                                             float* __restrict__ prod) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = (int)(i / C), c = (int)(i - (int64_t)n * C);
    prod[i] = qn[(int64_t)node_graph[n] * C + c] * xn[i];
}

// One block per graph: softmax of the node gates (PyG: exp(g - max) / (sum + 1e-16)), then
// out[g, :] = sum_n p_n xn[n, :] with nodes added in order (scatter_add order).
// PARTS = 16: gate[n] = sum of the 16 partial dot products the gate_nn product's epilogue left per node (split3.hip, rowdot) +
// gate_bias[0], summed in slot order (deterministic); PARTS = 1: gate[n] as is.  Graphs of up to 256 nodes keep their softmax
// weights in LDS (one expf per node instead of one per node and channel).
template <int PARTS>
__global__ __launch_bounds__(256) void k_graph_attention_pool(int C, const int32_t* __restrict__ graph_ptr,
                                                              const float* __restrict__ gate, const float* __restrict__ gate_bias,
                                                              const float* __restrict__ xn, float* __restrict__ out) {
    __shared__ float p_s[256];
    __shared__ float red[8];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1], cnt = n1 - n0;
    auto gate_of = [&](int n) {
        if (PARTS == 1) return gate[n];
        const float4* q = reinterpret_cast<const float4*>(gate + (int64_t)n * 16);
        const float4 a = q[0], b = q[1], c = q[2], d = q[3];
        return ((((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)))) + gate_bias[0];
    };
    if (cnt <= 256) {
        const float gv = tid < cnt ? gate_of(n0 + tid) : -INFINITY;
        float m = gv;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float ev = tid < cnt ? expf(gv - m) : 0.f;
        p_s[tid] = ev;
        __syncthreads();
        float den = 0.f;
        for (int n = 0; n < cnt; ++n) den += p_s[n];                  // in node order, like the scatter_add of the reference
        den += 1e-16f;
        for (int c = tid; c < C; c += 256) {
            float acc = 0.f;
            for (int n = 0; n < cnt; ++n) acc += (p_s[n] / den) * xn[(int64_t)(n0 + n) * C + c];
            out[(int64_t)g * C + c] = acc;
        }
        return;
    }
    float m = -INFINITY;
    for (int n = n0; n < n1; ++n) m = fmaxf(m, gate_of(n));
    float den = 0.f;
    for (int n = n0; n < n1; ++n) den += expf(gate_of(n) - m);
    den += 1e-16f;
    for (int c = tid; c < C; c += 256) {
        float acc = 0.f;
        for (int n = n0; n < n1; ++n) acc += (expf(gate_of(n) - m) / den) * xn[(int64_t)n * C + c];
        out[(int64_t)g * C + c] = acc;
    }
}

// feat[b] = [g || q || g * q]                                                            (:814)
__global__ __launch_bounds__(256) void k_head_features(int64_t B, int Q, const float* __restrict__ g, const float* __restrict__ q,
                                                       float* __restrict__ feat) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * Q) return;
    const int64_t b = i / Q;
    const int c = (int)(i - b * Q);
    const float gv = g[i], qv = q[i];
    feat[b * 3 * Q + c] = gv;
    feat[b * 3 * Q + Q + c] = qv;
    feat[b * 3 * Q + 2 * Q + c] = gv * qv;
}

struct PoolLayout { size_t h1, xn, qh, qn, prod, z, gate, scratch, scratch_bytes, apk2, total; };      // gate: [N, 16] partial sums, or [N]
static PoolLayout pool_layout(int64_t N, int64_t B, int Ch, int Dn) {
    PoolLayout L; size_t off = 0;
    auto take = [&](size_t n) { size_t r = off; off += align_up(n * sizeof(float), 256); return r; };
    L.h1 = take((size_t)N * Ch); L.xn = take((size_t)N * Ch); L.qh = take((size_t)B * Ch); L.qn = take((size_t)B * Ch);
    L.prod = take((size_t)N * Ch); L.z = take((size_t)N * Ch); L.gate = take((size_t)N * 16);
    L.scratch_bytes = linear_auto_scratch_bytes(N, Ch, Dn > Ch ? Dn : Ch);      // packed operands of the node MLP products
    L.scratch = take(L.scratch_bytes / sizeof(float));
    L.apk2 = take(split_packed_bytes(2, N, Ch) / sizeof(float) + 64);     // second packed-operand slot of the chained products
    L.total = off;
    return L;
}

}  // namespace gvqa

extern "C" {
using namespace gvqa;

size_t gvqa_attention_pool_workspace_bytes(const gvqa_graph* g, int32_t node_dim, int32_t channels) {
    if (!g) return 0;
    return pool_layout(g->num_nodes, g->num_graphs, channels, node_dim).total;
}

int gvqa_attention_pool_forward(const gvqa_graph* g, int32_t Dn, int32_t Ch, const gvqa_pool_params* p, const float* x,
                                const float* u, float* out, void* ws, size_t ws_bytes, void* stream_) {
    GVQA_REQUIRE(g && p, GVQA_E_INVALID, "attention_pool: null argument");
    GVQA_REQUIRE(Dn > 0 && Ch > 0, GVQA_E_INVALID, "attention_pool: bad dims");
    GVQA_REQUIRE(p->node0_weight && p->node0_bias && p->node2_weight && p->node2_bias && p->ques0_weight && p->ques0_bias &&
                 p->ques2_weight && p->ques2_bias && p->gate0_weight && p->gate0_bias && p->gate2_weight && p->gate2_bias,
                 GVQA_E_INVALID, "attention_pool: null weight");
    const int64_t N = g->num_nodes, B = g->num_graphs;
    PoolLayout L = pool_layout(N, B, Ch, Dn);
    GVQA_REQUIRE(ws && ws_bytes >= L.total, GVQA_E_WORKSPACE, "attention_pool: workspace %zu < required %zu", ws_bytes, L.total);
    if (B == 0) return GVQA_OK;
    GVQA_REQUIRE(u && out && (N == 0 || x), GVQA_E_INVALID, "attention_pool: null tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* base = static_cast<char*>(ws);
    auto P = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    StageTimer timer(GVQA_STAGE_OTHER, stream);
    int rc;
#define LIN(M_, N_, K_, A_, W_, b_, act_, C_)                                                                  \
    do { rc = launch_linear(M_, N_, K_, A_, K_, W_, K_, b_, act_, C_, N_, 1, 0, 0, 0, stream); if (rc) return rc; } while (0)
    // the three node-sized products: two-piece kernels with the scratch of this workspace (f32-input MFMA when not applicable)
#define NLIN(K_, A_, W_, b_, act_, C_)                                                                                    \
    do { LinearEpilogue e_{b_, nullptr, 0, nullptr, 0, act_};                                                             \
         rc = launch_linear_auto(N, Ch, K_, A_, K_, W_, K_, e_, C_, Ch, base + L.scratch, L.scratch_bytes, stream); if (rc) return rc; } while (0)
    LIN(B, Ch, Ch, u, p->ques0_weight, p->ques0_bias, 1, P(L.qh));                 // ques_nn (:165)
    LIN(B, Ch, Ch, P(L.qh), p->ques2_weight, p->ques2_bias, 0, P(L.qn));
    // gate_nn on ques_nn(u)[batch] * x' (:165).  Large batches on the two-piece kernels: the per-graph row scaling rides in the
    // operand pack (the product tensor is only ever a matrix-core operand), and the 512 -> 1 second Linear is the first product's
    // epilogue (16 partial dot products per node, summed in order by the pooling kernel: z is never stored)
    const bool fused = get_option(GVQA_OPT_PROJECTION) != GVQA_PROJECTION_F32 && N > 0 && Ch % 4 == 0 && Ch <= 512 &&
                       L.scratch_bytes >= linear_auto_scratch_bytes(N, Ch, Ch) &&
                       2.0 * (double)N * Ch * Ch >= 1e6 * (double)get_option(GVQA_OPT_SPLIT3_MIN_MFLOP) &&
                       ((reinterpret_cast<uintptr_t>(p->gate0_bias) | reinterpret_cast<uintptr_t>(p->gate2_weight)) & 15) == 0;
    // node_nn (:160).  Chained form (large batches, <= 512 channels): x is packed once; node_nn's first product leaves its result as
    // the second's packed operand (h1 never exists in fp32), the second leaves x' in fp32 (the pooling sum needs it) AND
    // ques_nn(u)[batch] * x' as gate_nn's packed operand: one pack pass instead of three (split3.hip, packed-output epilogue)
    bool chained = false;
    char* apk1 = base + L.scratch;
    char* wpk1 = apk1 + align_up(split_packed_bytes(2, N, Dn > Ch ? Dn : Ch), 256);
    char* apk2 = base + L.apk2;
    const bool al_ok = ((reinterpret_cast<uintptr_t>(p->node0_bias) | reinterpret_cast<uintptr_t>(p->node2_bias)) & 15) == 0;
    if (fused && al_ok && 2.0 * (double)N * Ch * Dn >= 1e6 * (double)get_option(GVQA_OPT_SPLIT3_MIN_MFLOP)) {
        rc = launch_split_pack(2, N, Dn, x, Dn, apk1, stream);
        if (rc) return rc;
        rc = launch_split_pack(2, Ch, Dn, p->node0_weight, Dn, wpk1, stream);
        if (rc) return rc;
        LinearEpilogue e1{p->node0_bias, nullptr, 0, nullptr, 0, 1};
        e1.pk_out = reinterpret_cast<uint16_t*>(apk2);
        rc = launch_linear_split(2, N, Ch, Dn, apk1, wpk1, e1, nullptr, Ch, stream);
        if (rc == GVQA_OK) {
            rc = launch_split_pack(2, Ch, Ch, p->node2_weight, Ch, wpk1, stream);
            if (rc) return rc;
            LinearEpilogue e2{p->node2_bias, nullptr, 0, nullptr, 0, 0};
            e2.pk_out = reinterpret_cast<uint16_t*>(apk1);
            e2.pk_mul = P(L.qn); e2.pk_mul_idx = g->node_graph; e2.pk_mul_ld = Ch;
            rc = launch_linear_split(2, N, Ch, Ch, apk2, wpk1, e2, P(L.xn), Ch, stream);
            if (rc) return rc;
            chained = true;
        } else if (rc != GVQA_E_UNSUPPORTED) return rc;
    }
    if (!chained) {
        NLIN(Dn, x, p->node0_weight, p->node0_bias, 1, P(L.h1));
        NLIN(Ch, P(L.h1), p->node2_weight, p->node2_bias, 0, P(L.xn));
    }
    bool parts = false;
    if (fused) {
        char* apk = apk1;
        char* wpk = wpk1;
        if (!chained) {
            rc = launch_split2h_pack_rowmul(N, Ch, P(L.xn), Ch, P(L.qn), g->node_graph, Ch, apk, stream);
            if (rc) return rc;
        }
        rc = launch_split_pack(2, Ch, Ch, p->gate0_weight, Ch, wpk, stream);
        if (rc) return rc;
        GVQA_HIP_CHECK(hipMemsetAsync(P(L.gate), 0, (size_t)N * 16 * sizeof(float), stream));
        LinearEpilogue ep{p->gate0_bias, nullptr, 0, nullptr, 0, 1};
        ep.rowdot_w = p->gate2_weight;
        ep.rowdot_out = P(L.gate);
        rc = launch_linear_split(2, N, Ch, Ch, apk, wpk, ep, nullptr, Ch, stream);
        if (rc == GVQA_OK) parts = true;
        else if (rc != GVQA_E_UNSUPPORTED) return rc;
        else {                                                         // (a tile shape without the row-dot epilogue: store z, then the small product)
            LinearEpilogue e1{p->gate0_bias, nullptr, 0, nullptr, 0, 1};
            rc = launch_linear_split(2, N, Ch, Ch, apk, wpk, e1, P(L.z), Ch, stream);
            if (rc) return rc;
            LIN(N, 1, Ch, P(L.z), p->gate2_weight, p->gate2_bias, 0, P(L.gate));
        }
    } else {
        if (N > 0) {
            hipLaunchKernelGGL(k_scale_rows_by_graph, dim3((unsigned)cdiv(N * Ch, 256)), dim3(256), 0, stream, N, Ch, g->node_graph,
                               P(L.qn), P(L.xn), P(L.prod));
            GVQA_LAUNCH_CHECK();
        }
        NLIN(Ch, P(L.prod), p->gate0_weight, p->gate0_bias, 1, P(L.z));                // gate_nn (:165)
        LIN(N, 1, Ch, P(L.z), p->gate2_weight, p->gate2_bias, 0, P(L.gate));
    }
#undef LIN
#undef NLIN
    if (parts) hipLaunchKernelGGL(k_graph_attention_pool<16>, dim3((unsigned)B), dim3(256), 0, stream, Ch, g->graph_ptr, P(L.gate), p->gate2_bias, P(L.xn), out);
    else hipLaunchKernelGGL(k_graph_attention_pool<1>, dim3((unsigned)B), dim3(256), 0, stream, Ch, g->graph_ptr, P(L.gate), p->gate2_bias, P(L.xn), out);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

size_t gvqa_answer_logits_workspace_bytes(int64_t B, int32_t Q, int32_t hidden) {
    if (B < 0 || Q <= 0 || hidden <= 0) return 0;
    return align_up((size_t)B * 3 * Q * 4, 256) + align_up((size_t)B * hidden * 4, 256);
}

int gvqa_answer_logits_forward(int64_t B, int32_t Q, int32_t hidden, int32_t A, const gvqa_classifier_params* p,
                               const float* g_feat, const float* q, float* logits, void* ws, size_t ws_bytes, void* stream_) {
    GVQA_REQUIRE(p && B >= 0 && Q > 0 && hidden > 0 && A > 0, GVQA_E_INVALID, "answer_logits: bad argument");
    GVQA_REQUIRE(p->fc1_weight && p->fc1_bias && p->fc2_weight && p->fc2_bias, GVQA_E_INVALID, "answer_logits: null weight");
    GVQA_REQUIRE(ws && ws_bytes >= gvqa_answer_logits_workspace_bytes(B, Q, hidden), GVQA_E_WORKSPACE, "answer_logits: workspace too small");
    if (B == 0) return GVQA_OK;
    GVQA_REQUIRE(g_feat && q && logits, GVQA_E_INVALID, "answer_logits: null tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* feat = static_cast<float*>(ws);
    float* hid = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)B * 3 * Q * 4, 256));
    StageTimer timer(GVQA_STAGE_OTHER, stream);
    hipLaunchKernelGGL(k_head_features, dim3((unsigned)cdiv(B * Q, 256)), dim3(256), 0, stream, B, Q, g_feat, q, feat);
    GVQA_LAUNCH_CHECK();
    int rc = launch_linear(B, hidden, 3 * Q, feat, 3 * Q, p->fc1_weight, 3 * Q, p->fc1_bias, 2 /* ELU */, hid, hidden, 1, 0, 0, 0, stream);
    if (rc) return rc;
    return launch_linear(B, A, hidden, hid, hidden, p->fc2_weight, hidden, p->fc2_bias, 0, logits, A, 1, 0, 0, 0, stream);
}

}  // extern "C"
