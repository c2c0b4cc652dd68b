// GINE / GCN variants of the execution module (gfx950) and the BN->ReLU chain their `*_seq`
// wrappers reduce to.
//
// Reference being replaced:
//   gine_seq / GINEConv  baseline_and_test_models/pipeline_model_gine.py:622-674  (PyG GINEConv:
//       nn((1+eps) x_i + sum_{j->i} relu(x_j + e_ji)), nn = Lin(812,300) -> ReLU -> Lin(300,300))
//   gcn_seq  / GCNConv   baseline_and_test_models/pipeline_model_gcn.py:622-669   (PyG 1.6/1.7
//       GCNConv: D^-1/2 (A + remaining self loops) D^-1/2 (x W) + b, weight stored [in, out])
// As written, both `*_seq.forward` DISCARD the conv result (`h` is never reassigned, :660 / :665-671):
// the module output is x pushed through 4 x (eval BatchNorm, ReLU) = k_bn_relu_chain.  The conv
// entry points below are the kernel-level target.
//
// Same restructuring as the GAT path: the [h || ins[batch]] / [edge_attr || ins[batch[src]]]
// concatenations are never materialised.  For GINE the instruction half of every message is
// relu(ins[g] + ins[g]) = relu(2 ins[g]), a per-GRAPH constant, so the aggregate of that half is
// in-degree x constant and its share of the first MLP layer is two tiny per-graph GEMMs; only
// the Dn node channels go through the gather kernel (HBM-bound: E*Dn edge rows streamed once).
#include <algorithm>

#include "common.h"

namespace gvqa {

constexpr int MAX_BN_STAGES = 8;

struct BnChainArgs {
    const float* w[MAX_BN_STAGES];
    const float* b[MAX_BN_STAGES];
    const float* m[MAX_BN_STAGES];
    const float* v[MAX_BN_STAGES];
    int stages;
    float eps;
};

// out = relu(bn_{S-1}( ... relu(bn_0(x)) ... )), eval statistics, torch's scale/shift form.
__global__ __launch_bounds__(256) void k_bn_relu_chain(int64_t total, int C, BnChainArgs a, const float* __restrict__ x,
                                                       float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float v = x[i];
        for (int s = 0; s < a.stages; ++s) {
            const float sc = a.w[s][c] * (1.0f / sqrtf(a.v[s][c] + a.eps));
            v = fmaxf(v * sc + (a.b[s][c] - a.m[s][c] * sc), 0.f);
        }
        out[i] = v;
    }
}

// float4 form (C % 4 == 0, C <= 512, 16-byte aligned rows): per-channel scale / shift of every stage once per block into LDS (the scalar
// kernel recomputes 1 / sqrt(var + eps) per ELEMENT and stage and moves 4 bytes per lane: 26 us for config 2's 36 + 36 MB = 2.8 TB/s)
constexpr int BN_CHAIN_CMAX = 512;
__global__ __launch_bounds__(256) void k_bn_relu_chain_v4(int64_t total4, int C, BnChainArgs a, const float4* __restrict__ x, float4* __restrict__ out) {
    __shared__ float sc_s[MAX_BN_STAGES][BN_CHAIN_CMAX], sh_s[MAX_BN_STAGES][BN_CHAIN_CMAX];
    for (int i = threadIdx.x; i < a.stages * C; i += 256) {
        const int s = i / C, c = i - s * C;
        const float sc = a.w[s][c] * (1.0f / sqrtf(a.v[s][c] + a.eps));
        sc_s[s][c] = sc;
        sh_s[s][c] = a.b[s][c] - a.m[s][c] * sc;
    }
    __syncthreads();
    const int C4 = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        float4 v = x[i];
        for (int s = 0; s < a.stages; ++s) {
            const float4 sc = *reinterpret_cast<const float4*>(&sc_s[s][c]), sh = *reinterpret_cast<const float4*>(&sh_s[s][c]);
            v.x = fmaxf(v.x * sc.x + sh.x, 0.f); v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
            v.z = fmaxf(v.z * sc.z + sh.z, 0.f); v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
        }
        out[i] = v;
    }
}

// ---- GINE -------------------------------------------------------------------------------------
// z[i, :] = sum_{e: dst(e)=i} relu(h[src(e), :] + edge_attr[eid(e), :]) + (1 + eps) * h[i, :]
// One wave per destination node, lanes stride the channels (16 B per lane when D % 4 == 0).
// Row order inside a node = original COO order (reference scatter_add order); the (1+eps) x_i term
// is added after the sum, as PyG does (out = propagate(...); out += (1 + eps) * x_r).
template <bool VEC>
__global__ __launch_bounds__(256) void k_gine_aggregate(int N, int D, const float* __restrict__ h,
                                                        const float* __restrict__ ea, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ csr_src,
                                                        const int32_t* __restrict__ csr_eid, float eps,
                                                        float* __restrict__ z, float* __restrict__ zmax) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const int lo = rowptr[i], hi = rowptr[i + 1];
    const float one_eps = 1.0f + eps;
    float vmax = 0.f;                                 // largest |z| of the row: the fused MLP's row scale (gine_mlp.hip) comes for free here
    if (VEC) {
        for (int c = lane * 4; c < D; c += 256) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int s = lo;
            for (; s + 1 < hi; s += 2) {      // two edges in flight
                const float4 v0 = *reinterpret_cast<const float4*>(h + (int64_t)csr_src[s] * D + c);
                const float4 e0 = *reinterpret_cast<const float4*>(ea + (int64_t)csr_eid[s] * D + c);
                const float4 v1 = *reinterpret_cast<const float4*>(h + (int64_t)csr_src[s + 1] * D + c);
                const float4 e1 = *reinterpret_cast<const float4*>(ea + (int64_t)csr_eid[s + 1] * D + c);
                acc.x += fmaxf(v0.x + e0.x, 0.f); acc.y += fmaxf(v0.y + e0.y, 0.f);
                acc.z += fmaxf(v0.z + e0.z, 0.f); acc.w += fmaxf(v0.w + e0.w, 0.f);
                acc.x += fmaxf(v1.x + e1.x, 0.f); acc.y += fmaxf(v1.y + e1.y, 0.f);
                acc.z += fmaxf(v1.z + e1.z, 0.f); acc.w += fmaxf(v1.w + e1.w, 0.f);
            }
            if (s < hi) {
                const float4 v0 = *reinterpret_cast<const float4*>(h + (int64_t)csr_src[s] * D + c);
                const float4 e0 = *reinterpret_cast<const float4*>(ea + (int64_t)csr_eid[s] * D + c);
                acc.x += fmaxf(v0.x + e0.x, 0.f); acc.y += fmaxf(v0.y + e0.y, 0.f);
                acc.z += fmaxf(v0.z + e0.z, 0.f); acc.w += fmaxf(v0.w + e0.w, 0.f);
            }
            const float4 xi = *reinterpret_cast<const float4*>(h + (int64_t)i * D + c);
            acc.x += one_eps * xi.x; acc.y += one_eps * xi.y; acc.z += one_eps * xi.z; acc.w += one_eps * xi.w;
            *reinterpret_cast<float4*>(z + (int64_t)i * D + c) = acc;
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w))));
        }
    } else {
        for (int c = lane; c < D; c += 64) {
            float acc = 0.f;
            for (int s = lo; s < hi; ++s)
                acc += fmaxf(h[(int64_t)csr_src[s] * D + c] + ea[(int64_t)csr_eid[s] * D + c], 0.f);
            const float v = acc + one_eps * h[(int64_t)i * D + c];
            z[(int64_t)i * D + c] = v;
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    if (zmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        if (lane == 0) zmax[i] = vmax;
    }
}

__global__ __launch_bounds__(256) void k_relu2(int64_t n, const float* __restrict__ in, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = fmaxf(in[i] + in[i], 0.f);      // relu(x_j + e) with both halves = ins[g]
}
// ... and [ins ; relu(2 ins)] as one [2 B, Di] operand: both per-graph products of the instruction half as ONE launch
__global__ __launch_bounds__(256) void k_ins_cat(int64_t n, const float* __restrict__ in, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float v = in[i]; out[i] = v; out[n + i] = fmaxf(v + v, 0.f); }
}

// y = relu(y + b1 + (1+eps) P1[g] + deg * P2[g])      (first MLP layer, instruction share)
__global__ __launch_bounds__(256) void k_gine_mid(int64_t N, int C, const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ node_graph, const float* __restrict__ b1,
                                                  const float* __restrict__ P1, const float* __restrict__ P2, float eps,
                                                  float* __restrict__ y) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = (int)(i / C), c = (int)(i - (int64_t)n * C);
    const int g = node_graph[n];
    const float deg = (float)(rowptr[n + 1] - rowptr[n]);
    float v = y[i] + b1[c];
    v += deg * P2[(int64_t)g * C + c] + (1.0f + eps) * P1[(int64_t)g * C + c];
    y[i] = fmaxf(v, 0.f);
}

// ---- GCN --------------------------------------------------------------------------------------
// PyG add_remaining_self_loops: existing self loops are dropped and exactly one unit self loop per
// node is appended, so deg[i] = 1 + #{e: dst(e) = i, src(e) != i}.
__global__ __launch_bounds__(256) void k_gcn_dis(int N, const int32_t* __restrict__ rowptr,
                                                 const int32_t* __restrict__ csr_src, float* __restrict__ dis) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    int deg = 1;
    for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) deg += (csr_src[s] != i) ? 1 : 0;
    dis[i] = 1.0f / sqrtf((float)deg);
}

// out[c, r] = in[r, c]   (GCNConv stores weight as [in, out]; the GEMM wants [out, in])
__global__ __launch_bounds__(256) void k_transpose(int rows, int cols, const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < rows && c0 + tx < cols) tile[k][tx] = in[(int64_t)(r0 + k) * cols + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < cols && r0 + tx < rows) out[(int64_t)(c0 + k) * rows + r0 + tx] = tile[tx][k];
}

// out[i] = sum_{e: src != i} dis[src] dis[i] (xw[src] + P[g(src)]) + dis[i]^2 (xw[i] + P[g(i)]) + b
// (non-self edges in COO order, then the appended self loop -- the reference's edge order).
__global__ __launch_bounds__(256) void k_gcn_aggregate(int N, int C, const float* __restrict__ xw,
                                                       const float* __restrict__ P, const float* __restrict__ dis,
                                                       const float* __restrict__ bias, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ csr_src,
                                                       const int32_t* __restrict__ node_graph, float* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const int lo = rowptr[i], hi = rowptr[i + 1];
    const float di = dis[i];
    for (int c = lane; c < C; c += 64) {
        float acc = 0.f;
        for (int s = lo; s < hi; ++s) {
            const int src = csr_src[s];
            if (src == i) continue;
            float v = xw[(int64_t)src * C + c];
            if (P) v += P[(int64_t)node_graph[src] * C + c];
            acc += (dis[src] * di) * v;
        }
        float v = xw[(int64_t)i * C + c];
        if (P) v += P[(int64_t)node_graph[i] * C + c];
        acc += (di * di) * v;
        out[(int64_t)i * C + c] = acc + (bias ? bias[c] : 0.f);
    }
}

// float4 form (C % 4 == 0, 16-byte aligned rows): ONE WAVE PER NODE, lanes over the row's channel quads -- the edge indices, the
// neighbours' normalisers and graph ids are wave-uniform (scalar loads, once per edge instead of once per (edge, quad) as in a
// thread-per-(node, quad) mapping: 25 us at config 2 = 2.8 TB/s), every neighbour row is read as whole 16-byte segments by
// consecutive lanes.  Same summation order as the scalar kernel: non-self edges in CSR (= COO) order, then the self loop.
__global__ __launch_bounds__(256) void k_gcn_aggregate_v4(int N, int C4, const float4* __restrict__ xw, const float4* __restrict__ P,
                                                          const float* __restrict__ dis, const float4* __restrict__ bias,
                                                          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_src,
                                                          const int32_t* __restrict__ node_graph, float4* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (i >= N) return;
    const int lo = rowptr[i], hi = rowptr[i + 1];
    const float di = dis[i];
    const float4* pi = P ? P + (int64_t)node_graph[i] * C4 : nullptr;
    // a lane owns the quads lane, lane + 64 (, ... in further rounds): the edge loop runs once per round of 128 quads
    for (int q0 = 0; q0 < C4; q0 += 128) {
        const int qa = q0 + lane, qb = q0 + 64 + lane;
        const bool ona = qa < C4, onb = qb < C4;
        float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        auto add = [&](int k, int q, const float4* row, const float4* prow, float w) {
            float4 v = row[q];
            if (prow) {
                const float4 pg = prow[q];
                v.x += pg.x; v.y += pg.y; v.z += pg.z; v.w += pg.w;
            }
            acc[k].x += w * v.x; acc[k].y += w * v.y; acc[k].z += w * v.z; acc[k].w += w * v.w;
        };
        for (int s = lo; s < hi; ++s) {
            const int src = csr_src[s];
            if (src == i) continue;
            const float4* row = xw + (int64_t)src * C4;
            const float4* prow = P ? P + (int64_t)node_graph[src] * C4 : nullptr;
            const float w = dis[src] * di;
            if (ona) add(0, qa, row, prow, w);
            if (onb) add(1, qb, row, prow, w);
        }
        const float4* row = xw + (int64_t)i * C4;
        if (ona) add(0, qa, row, pi, di * di);
        if (onb) add(1, qb, row, pi, di * di);
        if (bias) {
            if (ona) { const float4 b = bias[qa]; acc[0].x += b.x; acc[0].y += b.y; acc[0].z += b.z; acc[0].w += b.w; }
            if (onb) { const float4 b = bias[qb]; acc[1].x += b.x; acc[1].y += b.y; acc[1].z += b.z; acc[1].w += b.w; }
        }
        if (ona) out[(int64_t)i * C4 + qa] = acc[0];
        if (onb) out[(int64_t)i * C4 + qb] = acc[1];
    }
}

struct GineLayout { size_t z, y, tmp, P1, P2, scr, scr_bytes, zmax, mlp, total; };
static GineLayout gine_layout(int64_t N, int64_t B, int Dn, int Di, int C) {
    GineLayout L; size_t off = 0;
    auto take = [&](size_t n) { size_t r = off; off += align_up(n * sizeof(float), 256); return r; };
    L.z = take((size_t)N * Dn + 64); L.y = take((size_t)N * C); L.tmp = take((size_t)2 * B * Di);      // (tmp: [ins ; relu(2 ins)] of the fused form)
    L.P1 = take((size_t)B * C); L.P2 = take((size_t)B * C);                                             // (contiguous: the fused form's [2 B, C] product)
    L.zmax = take((size_t)N);
    L.mlp = take(gine_mlp_packed_bytes(C, Dn) / sizeof(float) + 64);
    // scratch of the two node-sized Linears on the two-piece products (operands packed per call; round 5: 54 -> ~36 us each at config 4)
    L.scr_bytes = std::max(linear_auto_scratch_bytes(N, C, Dn), linear_auto_scratch_bytes(N, C, C));
    L.scr = take(L.scr_bytes / sizeof(float) + 64);
    L.total = off;
    return L;
}
struct GcnLayout { size_t wt, xw, P, dis, total; };
static GcnLayout gcn_layout(int64_t N, int64_t B, int Dn, int Di, int C) {
    GcnLayout L; size_t off = 0;
    auto take = [&](size_t n) { size_t r = off; off += align_up(n * sizeof(float), 256); return r; };
    L.wt = take((size_t)C * (Dn + Di)); L.xw = take((size_t)N * C); L.P = take((size_t)B * C); L.dis = take((size_t)N);
    L.total = off;
    return L;
}

}  // namespace gvqa

extern "C" {
using namespace gvqa;

int gvqa_bn_relu_chain(int64_t N, int32_t C, int32_t num_stages, const gvqa_bn_params* stages, float bn_eps,
                       const float* x, float* out, void* stream_) {
    GVQA_REQUIRE(N >= 0 && C > 0 && num_stages >= 0 && num_stages <= MAX_BN_STAGES, GVQA_E_INVALID,
                 "bn_relu_chain: bad size (stages must be <= %d)", MAX_BN_STAGES);
    if (N == 0) return GVQA_OK;
    GVQA_REQUIRE(x && out && (num_stages == 0 || stages), GVQA_E_INVALID, "bn_relu_chain: null argument");
    BnChainArgs a;
    memset(&a, 0, sizeof(a));
    for (int s = 0; s < num_stages; ++s) {
        GVQA_REQUIRE(stages[s].weight && stages[s].bias && stages[s].mean && stages[s].var, GVQA_E_INVALID,
                     "bn_relu_chain: stage %d has a null tensor", s);
        a.w[s] = stages[s].weight; a.b[s] = stages[s].bias; a.m[s] = stages[s].mean; a.v[s] = stages[s].var;
    }
    a.stages = num_stages; a.eps = bn_eps;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer t(GVQA_STAGE_OTHER, stream);
    const int64_t total = N * C;
    int64_t blocks = cdiv(total, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (C % 4 == 0 && C <= BN_CHAIN_CMAX && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const int64_t total4 = total / 4;
        const int64_t b4 = std::min<int64_t>(cdiv(total4, 256), (int64_t)device_cu_count() * 8);
        hipLaunchKernelGGL(k_bn_relu_chain_v4, dim3((unsigned)b4), dim3(256), 0, stream, total4, C, a, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<float4*>(out));
    } else
        hipLaunchKernelGGL(k_bn_relu_chain, dim3((unsigned)blocks), dim3(256), 0, stream, total, C, a, x, out);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

size_t gvqa_gine_conv_workspace_bytes(const gvqa_graph* g, int32_t node_dim, int32_t ins_dim, int32_t C) {
    if (!g) return 0;
    return gine_layout(g->num_nodes, g->num_graphs, node_dim, ins_dim, C).total;
}

int gvqa_gine_conv_forward(const gvqa_graph* g, int32_t Dn, int32_t Di, int32_t C, const gvqa_gine_params* p,
                           const float* h, const float* edge_attr, const float* ins, float* out, void* ws,
                           size_t ws_bytes, void* stream_) {
    GVQA_REQUIRE(g && p, GVQA_E_INVALID, "gine_conv: null argument");
    GVQA_REQUIRE(Dn > 0 && Di >= 0 && C > 0, GVQA_E_INVALID, "gine_conv: bad dims");
    GVQA_REQUIRE(p->nn0_weight && p->nn0_bias && p->nn2_weight && p->nn2_bias, GVQA_E_INVALID, "gine_conv: null weight");
    const int64_t N = g->num_nodes, E = g->num_edges, B = g->num_graphs;
    GVQA_REQUIRE(Di == 0 || (g->finalized && g->intra_graph), GVQA_E_UNSUPPORTED,
                 "gine_conv: the per-graph instruction shortcut needs a finalized intra-graph batch; "
                 "pass concatenated inputs with ins_dim = 0 instead");
    GineLayout L = gine_layout(N, B, Dn, Di, C);
    GVQA_REQUIRE(ws && ws_bytes >= L.total, GVQA_E_WORKSPACE, "gine_conv: workspace %zu < required %zu", ws_bytes, L.total);
    if (N == 0) return GVQA_OK;
    GVQA_REQUIRE(h && out && (E == 0 || edge_attr) && (Di == 0 || ins), GVQA_E_INVALID, "gine_conv: null tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* base = static_cast<char*>(ws);
    auto P = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    const int ld1 = Dn + Di;
    {
        StageTimer t(GVQA_STAGE_MP, stream);
        const bool vec = (Dn % 4 == 0) && ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(edge_attr)) & 15) == 0;
        if (vec)
            hipLaunchKernelGGL(k_gine_aggregate<true>, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, stream, (int)N, Dn, h,
                               edge_attr, g->rowptr, g->csr_src, g->csr_eid, p->eps, P(L.z), P(L.zmax));
        else
            hipLaunchKernelGGL(k_gine_aggregate<false>, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, stream, (int)N, Dn, h,
                               edge_attr, g->rowptr, g->csr_src, g->csr_eid, p->eps, P(L.z), P(L.zmax));
        GVQA_LAUNCH_CHECK();
    }
    StageTimer t(GVQA_STAGE_PROJ, stream);
    int rc;
    // nn = Lin -> ReLU -> Lin as ONE kernel, the hidden rows in registers (gine_mlp.hip): C <= 320, the two-piece arithmetic of the products below
    static const bool fused_off = []() { const char* v = getenv("GVQA_GINE_FUSED"); return v && v[0] == '0'; }();      // (A/B switch)
    if (!fused_off && get_option(GVQA_OPT_PROJECTION) == GVQA_PROJECTION_SPLIT2H && N >= 1024 &&
        gine_mlp_supported(N, C, Dn, P(L.z), Dn, out, C, C) && ld1 % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(p->nn0_weight) | reinterpret_cast<uintptr_t>(p->nn2_weight)) & 15) == 0) {
        const float *P1 = nullptr, *P2 = nullptr;
        if (Di > 0) {       // both per-graph products of the instruction half in one launch: [ins ; relu(2 ins)] x W1[:, Dn:]^T -> [P1 ; P2]
            hipLaunchKernelGGL(k_ins_cat, dim3((unsigned)cdiv(B * Di, 256)), dim3(256), 0, stream, B * Di, ins, P(L.tmp));
            GVQA_LAUNCH_CHECK();
            rc = launch_linear(2 * B, C, Di, P(L.tmp), Di, p->nn0_weight + Dn, ld1, nullptr, 0, P(L.P1), C, 1, 0, 0, 0, stream);
            if (rc) return rc;
            P1 = P(L.P1); P2 = P(L.P1) + (size_t)B * C;
        }
        return launch_gine_mlp(N, C, Dn, P(L.z), Dn, P(L.zmax), p->nn0_weight, ld1, p->nn0_bias, p->nn2_weight, C, p->nn2_bias, P1, P2, C,
                               g->node_graph, g->rowptr, p->eps, out, C, base + L.mlp, stream);
    }
    if (Di == 0) {
        rc = launch_linear_auto(N, C, Dn, P(L.z), Dn, p->nn0_weight, ld1, LinearEpilogue{p->nn0_bias, nullptr, 0, nullptr, 0, 1}, P(L.y), C,
                                base + L.scr, L.scr_bytes, stream);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(k_relu2, dim3((unsigned)cdiv(B * Di, 256)), dim3(256), 0, stream, B * Di, ins, P(L.tmp));
        GVQA_LAUNCH_CHECK();
        rc = launch_linear(B, C, Di, ins, Di, p->nn0_weight + Dn, ld1, nullptr, 0, P(L.P1), C, 1, 0, 0, 0, stream);
        if (rc) return rc;
        rc = launch_linear(B, C, Di, P(L.tmp), Di, p->nn0_weight + Dn, ld1, nullptr, 0, P(L.P2), C, 1, 0, 0, 0, stream);
        if (rc) return rc;
        rc = launch_linear_auto(N, C, Dn, P(L.z), Dn, p->nn0_weight, ld1, LinearEpilogue{}, P(L.y), C, base + L.scr, L.scr_bytes, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gine_mid, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, stream, N, C, g->rowptr,
                           g->node_graph, p->nn0_bias, P(L.P1), P(L.P2), p->eps, P(L.y));
        GVQA_LAUNCH_CHECK();
    }
    return launch_linear_auto(N, C, C, P(L.y), C, p->nn2_weight, C, LinearEpilogue{p->nn2_bias, nullptr, 0, nullptr, 0, 0}, out, C, base + L.scr,
                              L.scr_bytes, stream);
}

size_t gvqa_gcn_conv_workspace_bytes(const gvqa_graph* g, int32_t node_dim, int32_t ins_dim, int32_t C) {
    if (!g) return 0;
    return gcn_layout(g->num_nodes, g->num_graphs, node_dim, ins_dim, C).total;
}

int gvqa_gcn_conv_forward(const gvqa_graph* g, int32_t Dn, int32_t Di, int32_t C, const gvqa_gcn_params* p, const float* h,
                          const float* ins, float* out, void* ws, size_t ws_bytes, void* stream_) {
    GVQA_REQUIRE(g && p, GVQA_E_INVALID, "gcn_conv: null argument");
    GVQA_REQUIRE(Dn > 0 && Di >= 0 && C > 0 && p->weight, GVQA_E_INVALID, "gcn_conv: bad dims / null weight");
    const int64_t N = g->num_nodes, B = g->num_graphs;
    GcnLayout L = gcn_layout(N, B, Dn, Di, C);
    GVQA_REQUIRE(ws && ws_bytes >= L.total, GVQA_E_WORKSPACE, "gcn_conv: workspace %zu < required %zu", ws_bytes, L.total);
    if (N == 0) return GVQA_OK;
    GVQA_REQUIRE(h && out && (Di == 0 || ins), GVQA_E_INVALID, "gcn_conv: null tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* base = static_cast<char*>(ws);
    auto P = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    const int D = Dn + Di;
    int rc;
    {
        StageTimer t(GVQA_STAGE_PROJ, stream);
        hipLaunchKernelGGL(k_transpose, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(D, 32)), dim3(256), 0, stream, D, C,
                           p->weight, P(L.wt));
        GVQA_LAUNCH_CHECK();
        rc = launch_linear(N, C, Dn, h, Dn, P(L.wt), D, nullptr, 0, P(L.xw), C, 1, 0, 0, 0, stream);
        if (rc) return rc;
        if (Di > 0) {
            rc = launch_linear(B, C, Di, ins, Di, P(L.wt) + Dn, D, nullptr, 0, P(L.P), C, 1, 0, 0, 0, stream);
            if (rc) return rc;
        }
    }
    StageTimer t(GVQA_STAGE_MP, stream);
    hipLaunchKernelGGL(k_gcn_dis, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, stream, (int)N, g->rowptr, g->csr_src, P(L.dis));
    const bool v4 = C % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (!p->bias || (reinterpret_cast<uintptr_t>(p->bias) & 15) == 0);
    if (v4) {
        hipLaunchKernelGGL(k_gcn_aggregate_v4, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, stream, (int)N,
                           C / 4, reinterpret_cast<const float4*>(P(L.xw)), Di > 0 ? reinterpret_cast<const float4*>(P(L.P)) : nullptr,
                           P(L.dis), reinterpret_cast<const float4*>(p->bias), g->rowptr, g->csr_src, g->node_graph,
                           reinterpret_cast<float4*>(out));
    } else {
        hipLaunchKernelGGL(k_gcn_aggregate, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, stream, (int)N, C, P(L.xw),
                           Di > 0 ? P(L.P) : nullptr, P(L.dis), p->bias, g->rowptr, g->csr_src, g->node_graph, out);
    }
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // extern "C"
