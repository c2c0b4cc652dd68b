// Backward of the GAT message-passing step (what PyG's autograd does through
// gather / utils.softmax / scatter_add, gat_skip.py:155,183-208, + the head mean :162-165).
//
// Forward (gvqa_gat_message_passing without graph terms / bias / skip / BN):
//     z[e,h]  = a_node[src_e, h] + a_node[dst_e, H+h] + a_edge[e, h]
//     alpha   = softmax over the in-edges of dst of leaky_relu(z)          (saved: alpha_out)
//     out[i]  = (1/H) sum_h sum_{e -> i} alpha[e,h] * mask[e,h] * xp[src_e, h, :]
// Given dout = dL/dout this file produces
//     dxp[j,h,:]    = (1/H) sum_{e: src_e = j} alpha*mask [e,h] * dout[dst_e, :]            (k_gat_mp_bwd_src)
//     dalpha'[e,h]  = (1/H) dout[dst_e] . xp[src_e,h,:]                                      (k_gat_mp_bwd_dst)
//     dz[e,h]       = alpha (mask dalpha' - sum_{e' -> dst} alpha mask dalpha') * leaky'(z)  (softmax + leaky backward)
//     da_edge = dz,  da_node[i, H+h] = sum_{e -> i} dz,  da_node[j, h] = sum_{e: src = j} dz.
// No atomics: the by-destination kernel walks the CSR of the forward graph, the by-source kernel walks the
// CSR of the TRANSPOSED graph (gvqa_graph_build on the flipped edge_index; its csr_src holds destinations,
// its csr_eid the same COO edge ids) -- both are deterministic.  One wave per node; a lane owns W
// consecutive channels per 64*W-channel chunk (W = 4: float4 rows; W = 1: any C).  HBM-bound: dout rows and
// xp rows are gathered once per edge (L2-resident within a graph), dxp is written once.
#include "common.h"

namespace gvqa {

namespace {

constexpr int BWD_MAXH = 8;
constexpr int BWD_CAP = 64;          // in-edges per node whose dalpha' is staged in LDS (beyond: recomputed)

template <int W>
__device__ __forceinline__ void loadw(const float* p, float (&v)[W]) {
    if (W == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int q = 0; q < W; ++q) v[q] = p[q];
    }
}
template <int W>
__device__ __forceinline__ void storew(float* p, const float (&v)[W]) {
    if (W == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
        for (int q = 0; q < W; ++q) p[q] = v[q];
    }
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct BwdArgs {
    int N, C, H;
    float slope;
    const float* xp; int64_t xp_ld;
    const float* a_node;               // [N, 2H] or NULL
    const float* a_edge; int64_t a_edge_stride;
    const float* alpha;                // [E, H] COO (softmax output, before mask)
    const float* mask;                 // [E, H] COO or NULL
    const float* dout; int64_t dout_ld;
    float* dxp; int64_t dxp_ld;
    float* da_node;                    // [N, 2H]
    float* da_edge;                    // [E, H]
    // forward graph (CSR by destination) and transposed graph (CSR by source)
    const int32_t *rowptr, *csr_src, *csr_eid;
    const int32_t *t_rowptr, *t_csr_dst, *t_csr_eid;
};

// ---- by destination: dalpha', softmax / leaky backward, da_edge, da_node[:, H:] -------------------
template <int W, int KC, int HT>      // HT: heads whose xp rows are in flight together (H <= HT)
__global__ __launch_bounds__(256) void k_gat_mp_bwd_dst(BwdArgs a) {
    __shared__ float stage[4][BWD_CAP * BWD_MAXH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (i >= a.N) return;
    const int H = a.H, C = a.C;
    const int lo = a.rowptr[i], hi = a.rowptr[i + 1];
    const float inv_h = 1.0f / H;
    float dco[KC][W];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = (lane + k * 64) * W;
        if (c < C) {
            loadw<W>(a.dout + (int64_t)i * a.dout_ld + c, dco[k]);
#pragma unroll
            for (int q = 0; q < W; ++q) dco[k][q] *= inv_h;
        } else {
#pragma unroll
            for (int q = 0; q < W; ++q) dco[k][q] = 0.f;
        }
    }
    // dalpha'[e,h] = dout[i]/H . xp[src,h,:]  (every lane ends up with the wave-reduced value)
    auto dalpha_of = [&](int src, int h) -> float {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = (lane + k * 64) * W;
            if (c < C) {
                float v[W];
                loadw<W>(a.xp + (int64_t)src * a.xp_ld + (int64_t)h * C + c, v);
#pragma unroll
                for (int q = 0; q < W; ++q) acc += v[q] * dco[k][q];
            }
        }
        return wsum(acc);
    };
    float t[BWD_MAXH];
#pragma unroll
    for (int h = 0; h < BWD_MAXH; ++h) t[h] = 0.f;
    // two in-edges per trip: the xp rows of ALL heads of both edges are loaded before the first dot product (one trip to memory per
    // edge pair instead of one per (edge, head)), and the 2 H wave reductions run interleaved
    for (int s = lo; s < hi; s += 2) {
        const bool two = s + 1 < hi;
        int src[2], eid[2];
        src[0] = a.csr_src[s]; eid[0] = a.csr_eid[s];
        src[1] = a.csr_src[two ? s + 1 : s]; eid[1] = a.csr_eid[two ? s + 1 : s];
        float v[2][HT][KC][W];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int h = 0; h < HT; ++h)
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const int c = (lane + k * 64) * W;
                    if (h < H && c < C) loadw<W>(a.xp + (int64_t)src[e] * a.xp_ld + (int64_t)h * C + c, v[e][h][k]);
                    else {
#pragma unroll
                        for (int q = 0; q < W; ++q) v[e][h][k][q] = 0.f;
                    }
                }
        float al[2][HT], mk[2][HT], d[2][HT];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                al[e][h] = h < H ? a.alpha[(int64_t)eid[e] * H + h] : 0.f;
                mk[e][h] = (h < H && a.mask) ? a.mask[(int64_t)eid[e] * H + h] : 1.f;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < KC; ++k)
#pragma unroll
                    for (int q = 0; q < W; ++q) acc += v[e][h][k][q] * dco[k][q];
                d[e][h] = acc;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int h = 0; h < HT; ++h) d[e][h] += __shfl_xor(d[e][h], o, 64);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e == 1 && !two) break;
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                if (h < H) {
                    const float dd = a.mask ? d[e][h] * mk[e][h] : d[e][h];
                    t[h] += al[e][h] * dd;
                    if (s + e - lo < BWD_CAP && lane == 0) stage[wave][(s + e - lo) * BWD_MAXH + h] = dd;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // dz = alpha (dalpha - t) leaky'(z); lanes over in-edges, one head at a time
#pragma unroll
    for (int h = 0; h < BWD_MAXH; ++h) {
        if (h < H) {
            const float ar = a.a_node ? a.a_node[(int64_t)i * 2 * H + H + h] : 0.f;
            float dsum = 0.f;
            for (int s0 = lo; s0 < hi; s0 += 64) {
                const int s = s0 + lane;
                float dz = 0.f;
                // edges beyond the LDS capacity recompute their dot product: wave-uniform loop, all lanes take part
                float d_lane = 0.f;
                if (s0 - lo >= BWD_CAP) {
                    const int cnt = min(64, hi - s0);
                    for (int j = 0; j < cnt; ++j) {
                        const int src_j = a.csr_src[s0 + j], eid_j = a.csr_eid[s0 + j];
                        float d = dalpha_of(src_j, h);
                        if (a.mask) d *= a.mask[(int64_t)eid_j * H + h];
                        if (lane == j) d_lane = d;
                    }
                }
                if (s < hi) {
                    const int src = a.csr_src[s], eid = a.csr_eid[s];
                    const float d = (s - lo < BWD_CAP) ? stage[wave][(s - lo) * BWD_MAXH + h] : d_lane;
                    const float al = a.alpha[(int64_t)eid * H + h];
                    const float z = (a.a_node ? a.a_node[(int64_t)src * 2 * H + h] : 0.f) + ar +
                                    a.a_edge[(int64_t)eid * a.a_edge_stride + h];
                    dz = al * (d - t[h]) * (z > 0.f ? 1.f : a.slope);
                    a.da_edge[(int64_t)eid * H + h] = dz;
                }
                dsum += dz;
            }
            dsum = wsum(dsum);
            if (lane == 0) a.da_node[(int64_t)i * 2 * H + H + h] = dsum;
        }
    }
}

// ---- by source: dxp, da_node[:, :H] ------------------------------------------------------------------
template <int W, int KC, int HT>      // HT: heads held in registers at once (H <= HT)
__global__ __launch_bounds__(256) void k_gat_mp_bwd_src(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (j >= a.N) return;
    const int H = a.H, C = a.C;
    const int lo = a.t_rowptr[j], hi = a.t_rowptr[j + 1];
    const float inv_h = 1.0f / H;
    float acc[HT][KC][W];
#pragma unroll
    for (int h = 0; h < HT; ++h)
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int q = 0; q < W; ++q) acc[h][k][q] = 0.f;
    // two out-edges per trip: their index loads, then their coefficient and dout-row loads, are issued together (one dependent
    // chain per trip instead of one per edge); the second edge of an odd tail is a repeat of the first with zero weight.  The sums
    // keep the edge order.
    for (int s = lo; s < hi; s += 2) {
        const bool two = s + 1 < hi;
        const int dst0 = a.t_csr_dst[s], eid0 = a.t_csr_eid[s];
        const int dst1 = a.t_csr_dst[two ? s + 1 : s], eid1 = a.t_csr_eid[two ? s + 1 : s];
        float am0[HT], am1[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) {
            am0[h] = am1[h] = 0.f;
            if (h < H) {
                am0[h] = a.alpha[(int64_t)eid0 * H + h] * inv_h;
                am1[h] = two ? a.alpha[(int64_t)eid1 * H + h] * inv_h : 0.f;
                if (a.mask) { am0[h] *= a.mask[(int64_t)eid0 * H + h]; am1[h] *= a.mask[(int64_t)eid1 * H + h]; }
            }
        }
        float v0[KC][W], v1[KC][W];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = (lane + k * 64) * W;
            if (c < C) {
                loadw<W>(a.dout + (int64_t)dst0 * a.dout_ld + c, v0[k]);
                loadw<W>(a.dout + (int64_t)dst1 * a.dout_ld + c, v1[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = (lane + k * 64) * W;
            if (c < C) {
#pragma unroll
                for (int h = 0; h < HT; ++h)
#pragma unroll
                    for (int q = 0; q < W; ++q) acc[h][k][q] += am0[h] * v0[k][q];
                if (two) {
#pragma unroll
                    for (int h = 0; h < HT; ++h)
#pragma unroll
                        for (int q = 0; q < W; ++q) acc[h][k][q] += am1[h] * v1[k][q];
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < HT; ++h) {
        if (h < H) {
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int c = (lane + k * 64) * W;
                if (c < C) storew<W>(a.dxp + (int64_t)j * a.dxp_ld + (int64_t)h * C + c, acc[h][k]);
            }
            // da_l[j,h] = sum over the out-edges of dz (written by k_gat_mp_bwd_dst, an earlier launch)
            float dsum = 0.f;
            for (int s = lo + lane; s < hi; s += 64) dsum += a.da_edge[(int64_t)a.t_csr_eid[s] * H + h];
            dsum = wsum(dsum);
            if (lane == 0) a.da_node[(int64_t)j * 2 * H + h] = dsum;
        }
    }
}

template <int W, int KC>
static void launch_bwd(const BwdArgs& a, hipStream_t stream) {
    const dim3 grid((unsigned)cdiv(a.N, 4));
    if (a.H <= 4) hipLaunchKernelGGL((k_gat_mp_bwd_dst<W, KC, 4>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_gat_mp_bwd_dst<W, KC, 8>), grid, dim3(256), 0, stream, a);
    if (a.H <= 4) hipLaunchKernelGGL((k_gat_mp_bwd_src<W, KC, 4>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_gat_mp_bwd_src<W, KC, 8>), grid, dim3(256), 0, stream, a);
}

}  // namespace
}  // namespace gvqa

extern "C" int gvqa_gat_mp_backward(const gvqa_graph* g, const gvqa_graph* gt, const gvqa_gat_mp_bwd_desc* d, void* stream_) {
    using namespace gvqa;
    GVQA_REQUIRE(g && gt && d, GVQA_E_INVALID, "gat_mp_backward: null argument");
    GVQA_REQUIRE(g->valid && gt->valid && g->num_nodes == gt->num_nodes && g->num_edges == gt->num_edges, GVQA_E_INVALID,
                 "gat_mp_backward: the transposed graph must be built from the flipped edge_index of the same batch");
    const int C = d->C, H = d->H;
    GVQA_REQUIRE(C > 0 && H > 0 && H <= BWD_MAXH, GVQA_E_UNSUPPORTED, "gat_mp_backward: needs 1 <= H <= %d", BWD_MAXH);
    GVQA_REQUIRE(C <= 1024, GVQA_E_UNSUPPORTED, "gat_mp_backward: C > 1024");
    if (g->num_nodes == 0) return GVQA_OK;
    GVQA_REQUIRE(d->xp && d->a_edge && d->alpha && d->dout && d->dxp && d->da_node && d->da_edge, GVQA_E_INVALID,
                 "gat_mp_backward: null tensor");
    BwdArgs a;
    a.N = (int)g->num_nodes; a.C = C; a.H = H; a.slope = d->negative_slope;
    a.xp = d->xp; a.xp_ld = d->xp_ld ? d->xp_ld : (int64_t)H * C;
    a.a_node = d->a_node; a.a_edge = d->a_edge; a.a_edge_stride = d->a_edge_stride ? d->a_edge_stride : H;
    a.alpha = d->alpha; a.mask = d->alpha_mask;
    a.dout = d->dout; a.dout_ld = d->dout_ld ? d->dout_ld : C;
    a.dxp = d->dxp; a.dxp_ld = d->dxp_ld ? d->dxp_ld : (int64_t)H * C;
    a.da_node = d->da_node; a.da_edge = d->da_edge;
    a.rowptr = g->rowptr; a.csr_src = g->csr_src; a.csr_eid = g->csr_eid;
    a.t_rowptr = gt->rowptr; a.t_csr_dst = gt->csr_src; a.t_csr_eid = gt->csr_eid;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer timer(GVQA_STAGE_MP, stream);
    const bool vec = C % 4 == 0 && a.xp_ld % 4 == 0 && a.dout_ld % 4 == 0 && a.dxp_ld % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(a.xp) | reinterpret_cast<uintptr_t>(a.dout) |
                       reinterpret_cast<uintptr_t>(a.dxp)) & 15) == 0;
    if (vec) {
        if (C <= 256) launch_bwd<4, 1>(a, stream);
        else if (C <= 512) launch_bwd<4, 2>(a, stream);
        else launch_bwd<4, 4>(a, stream);
    } else {
        if (C <= 256) launch_bwd<1, 4>(a, stream);
        else launch_bwd<1, 16>(a, stream);
    }
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}
