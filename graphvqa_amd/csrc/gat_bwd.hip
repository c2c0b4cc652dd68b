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
#include <mutex>

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
    const float* dsum;                 // [N, H] or NULL: added to dalpha' of every in-edge of a node before the mask (gradient of
                                       // s[i,h] = sum_e alpha mask -- the per-graph rows of the projection folded out of xp)
    const float* dout; int64_t dout_ld;
    float* dxp; int64_t dxp_ld;
    unsigned* dxp_absmax;              // NULL or [GVQA_ABSMAX_SLOTS] (zeroed): largest |dxp| per slot, as uint bit patterns
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
                    const float dx = d[e][h] + (a.dsum ? a.dsum[(int64_t)i * H + h] : 0.f);
                    const float dd = a.mask ? dx * mk[e][h] : dx;
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
                        float d = dalpha_of(src_j, h) + (a.dsum ? a.dsum[(int64_t)i * H + h] : 0.f);
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
    if (a.dxp_absmax) {      // largest magnitude of the rows this wave writes (the scale of the weight-gradient product's operand)
        float m = 0.f;
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int q = 0; q < W; ++q) m = fmaxf(m, fabsf(acc[h][k][q]));      // (unused heads / channels hold zeros)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) atomicMax(a.dxp_absmax + (j & (GVQA_ABSMAX_SLOTS - 1)), __float_as_uint(m));
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

// ---- by destination, LDS-tiled: one block per graph (intra-graph batches) -------------------------------------------
// The wave-per-node kernel above gathers a whole xp row (H C floats) per (edge, head) out of L2: 2.1 GB of L2 reads per
// launch at config 3 for 0.69 GB of distinct data.  Here -- as in the forward kernel k_gat_mp_tiled -- a graph's slabs
// dout[n0:n1, :] and xp[n0:n1, :, :] are streamed through LDS once, in stages of one channel range: a [n x cw] tile of dout and
// one [n x cw] tile of xp per head (H + 1 tiles, cw chosen so that two stages fit 40 KiB: three blocks per CU), by
// global_load_lds, one stage in flight while the previous one is consumed.  A stage adds to the dot products
// dalpha'[e, h] = dout[dst_e] . xp[src_e, h] of ALL the graph's edges and heads: 4 lanes per edge, each over a quarter of the
// range's float4 columns, accumulators in registers across the stages.  Afterwards the softmax / leaky-relu backward of the
// graph runs out of LDS.  No ordinary global load inside the stage loop.
constexpr int BT_THREADS = 512, BT_LPE = 4, BT_ITEMS = 4;            // threads, lanes per edge, edges per thread (<= 512 edges per graph)
typedef __attribute__((address_space(3))) char* bt_lds_ptr_t;
__device__ __forceinline__ void bt_dma16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
struct BwdTiledArgs {
    BwdArgs a;
    const int32_t* graph_ptr;
    int cw, e_cap, n_cap;
};
static size_t bwd_tiled_lds_bytes(int e_cap, int n_cap, int H, int cw) {
    // [src | dst | eid][e_cap] ints, rowptr [n_cap + 1], [alpha | zraw | d][e_cap * H] floats, 2 stages x (H + 1) tiles
    size_t off = (size_t)3 * e_cap * 4 + (size_t)(n_cap + 1) * 4;
    off = (off + 15) & ~(size_t)15;
    off += (size_t)3 * e_cap * H * 4;
    off = (off + 15) & ~(size_t)15;
    const size_t stage = (((size_t)(H + 1) * n_cap * (cw / 4) + BT_THREADS - 1) / BT_THREADS) * BT_THREADS * 16;
    return off + 2 * stage;
}

template <int H>
__global__ __launch_bounds__(BT_THREADS) void k_gat_mp_bwd_dst_tiled(BwdTiledArgs t) {
    extern __shared__ __attribute__((aligned(16))) char bt_smem[];
    const BwdArgs& a = t.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = blockIdx.x;
    const int n0 = t.graph_ptr[g], tn = t.graph_ptr[g + 1] - n0;
    if (tn <= 0) return;
    const int e0 = a.rowptr[n0], ne = a.rowptr[n0 + tn] - e0;
    const int C = a.C;
    int* src_l = reinterpret_cast<int*>(bt_smem);
    int* dst_l = src_l + t.e_cap;
    int* eid_l = dst_l + t.e_cap;
    int* rowp_l = eid_l + t.e_cap;
    size_t off = ((size_t)3 * t.e_cap * 4 + (size_t)(t.n_cap + 1) * 4 + 15) & ~(size_t)15;
    float* al_s = reinterpret_cast<float*>(bt_smem + off);            // alpha [e][H]
    float* zr_s = al_s + (size_t)t.e_cap * H;                          // a_l[src] + a_e[eid]
    float* d_s = zr_s + (size_t)t.e_cap * H;                           // dalpha' (x mask)
    off = (off + (size_t)3 * t.e_cap * H * 4 + 15) & ~(size_t)15;
    // a stage = H + 1 dense [tn x q4c] float4 tiles back to back (dout, xp head 0 .. H-1), padded to whole DMA rounds at its end
    const unsigned stage_bytes = (unsigned)((((size_t)(H + 1) * t.n_cap * (t.cw >> 2) + BT_THREADS - 1) / BT_THREADS) * BT_THREADS * 16);
    const unsigned lds_base = (unsigned)(size_t)(bt_lds_ptr_t)bt_smem + (unsigned)off;
    const char* bufs = bt_smem + off;
    const int wave_unit0 = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int nch = (C + t.cw - 1) / t.cw;

    // DMA of stage r (channel range r) into stage buffer r & 1: H + 1 [tn x q4c] float4 tiles; every wave issues the same number
    // of DMAs (lanes past the end re-load the last unit into the tile's padding)
    auto prefetch = [&](int r) {
        const int c0 = r * t.cw, q4c = min(t.cw, C - c0) >> 2, tu = tn * q4c, units = (H + 1) * tu;
        unsigned dst0 = lds_base + (unsigned)(r & 1) * stage_bytes + (unsigned)wave_unit0 * 16u;
        for (int u0 = 0; u0 < units; u0 += BT_THREADS) {
            const int u = min(u0 + tid, units - 1);
            const int tile = u / tu, v = u - tile * tu, row = v / q4c, col = v - row * q4c;
            const float* src = tile == 0 ? a.dout + (int64_t)(n0 + row) * a.dout_ld + c0 + col * 4
                                         : a.xp + (int64_t)(n0 + row) * a.xp_ld + (int64_t)(tile - 1) * C + c0 + col * 4;
            bt_dma16(src, __builtin_amdgcn_readfirstlane(dst0));
            dst0 += BT_THREADS * 16;
        }
    };
    prefetch(0);

    // ---- prologue: local CSR with destinations, coefficients, destination-independent logit terms ----
    for (int s = tid; s < ne; s += BT_THREADS) {
        const int src = a.csr_src[e0 + s], eid = a.csr_eid[e0 + s];
        src_l[s] = src - n0;
        eid_l[s] = eid;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            al_s[s * H + h] = a.alpha[(int64_t)eid * H + h];
            zr_s[s * H + h] = (a.a_node ? a.a_node[(int64_t)src * 2 * H + h] : 0.f) + a.a_edge[(int64_t)eid * a.a_edge_stride + h];
        }
    }
    for (int i = tid; i <= tn; i += BT_THREADS) rowp_l[i] = a.rowptr[n0 + i] - e0;
    __syncthreads();
    for (int i = tid; i < tn; i += BT_THREADS)
        for (int s = rowp_l[i]; s < rowp_l[i + 1]; ++s) dst_l[s] = i;
    __syncthreads();

    // ---- stage loop: partial dot products of this thread's (edge, column quarter) items ----
    const int part = tid & (BT_LPE - 1);
    int e_it[BT_ITEMS], so[BT_ITEMS], dof[BT_ITEMS];
    float acc[BT_ITEMS][H];
#pragma unroll
    for (int k = 0; k < BT_ITEMS; ++k) {
        const int e = (tid >> 2) + k * (BT_THREADS / BT_LPE);
        e_it[k] = e < ne ? e : -1;
        so[k] = e < ne ? src_l[e] : 0;
        dof[k] = e < ne ? dst_l[e] : 0;
#pragma unroll
        for (int h = 0; h < H; ++h) acc[k][h] = 0.f;
    }
    for (int r = 0; r < nch; ++r) {
        const int q4c = min(t.cw, C - r * t.cw) >> 2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (r + 1 < nch) prefetch(r + 1);
        const char* sb = bufs + (size_t)(r & 1) * stage_bytes;
        const float4* db = reinterpret_cast<const float4*>(sb);
        const int tu = tn * q4c;
#pragma unroll
        for (int k = 0; k < BT_ITEMS; ++k) {
            if (e_it[k] < 0) continue;
            const float4* dr = db + dof[k] * q4c;
            for (int q = part; q < q4c; q += BT_LPE) {
                const float4 dv = dr[q];
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float4 xv = db[(h + 1) * tu + so[k] * q4c + q];
                    acc[k][h] += dv.x * xv.x + dv.y * xv.y + dv.z * xv.z + dv.w * xv.w;
                }
            }
        }
    }
    // ---- the four column quarters of an edge -> dalpha'[e, h] (x 1/H, x mask) in LDS ----
    const float inv_h = 1.0f / H;
#pragma unroll
    for (int k = 0; k < BT_ITEMS; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float v = acc[k][h];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            if (part == 0 && e_it[k] >= 0) {
                v *= inv_h;
                if (a.dsum) v += a.dsum[(int64_t)(n0 + dof[k]) * H + h];
                if (a.mask) v *= a.mask[(int64_t)eid_l[e_it[k]] * H + h];
                d_s[e_it[k] * H + h] = v;
            }
        }
    __syncthreads();
    // ---- softmax + leaky-relu backward per (node, head): dz = alpha (d - sum_row alpha d) leaky'(z) ----
    for (int it = tid; it < tn * H; it += BT_THREADS) {
        const int i = it / H, h = it - i * H;
        const int lo = rowp_l[i], hi = rowp_l[i + 1];
        const float ar = a.a_node ? a.a_node[(int64_t)(n0 + i) * 2 * H + H + h] : 0.f;
        float tsum = 0.f;
        for (int s = lo; s < hi; ++s) tsum += al_s[s * H + h] * d_s[s * H + h];
        float dsum = 0.f;
        for (int s = lo; s < hi; ++s) {
            const float z = zr_s[s * H + h] + ar;
            const float dz = al_s[s * H + h] * (d_s[s * H + h] - tsum) * (z > 0.f ? 1.f : a.slope);
            a.da_edge[(int64_t)eid_l[s] * H + h] = dz;
            dsum += dz;
        }
        a.da_node[(int64_t)(n0 + i) * 2 * H + H + h] = dsum;
    }
}

// the by-destination pass on the tiled kernel when the batch allows it; false = not applicable (the caller runs the wave-per-node one)
template <int H>
static bool launch_bwd_dst_tiled_h(const BwdTiledArgs& t, int64_t B, size_t lds, hipStream_t stream) {
    static std::mutex mu;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_set[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gat_mp_bwd_dst_tiled<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess) return false;
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL((k_gat_mp_bwd_dst_tiled<H>), dim3((unsigned)B), dim3(BT_THREADS), lds, stream, t);
    return true;
}
static bool launch_bwd_dst_tiled(const gvqa_graph* g, const BwdArgs& a, bool vec, hipStream_t stream) {
    static const bool off = []() { const char* v = getenv("GVQA_BWD_TILED"); return v && v[0] == '0'; }();
    if (off || !vec || !g->finalized || !g->intra_graph || !g->graph_ptr || g->num_graphs <= 0 || g->num_graphs > 0x7fffffff) return false;
    if (!(a.H == 1 || a.H == 2 || a.H == 4 || a.H == 8)) return false;
    if (g->max_graph_edges > BT_ITEMS * (BT_THREADS / BT_LPE) || g->max_graph_nodes <= 0) return false;
    BwdTiledArgs t;
    t.a = a; t.graph_ptr = g->graph_ptr;
    t.e_cap = g->max_graph_edges > 0 ? g->max_graph_edges : 1;
    t.n_cap = g->max_graph_nodes;
    // channel range: two stages of (H + 1) tiles within ~40 KiB (three blocks per CU), row segments of >= 64 bytes
    static const size_t target = []() { const char* v = getenv("GVQA_BWD_LDS"); return v ? (size_t)atoi(v) : (size_t)(160 * 1024 / 3); }();
    int cw = 0;
    for (int c = 128; c >= 16; c -= 4)                   // widest range whose two stages fit three blocks per CU
        if (c <= a.C && bwd_tiled_lds_bytes(t.e_cap, t.n_cap, a.H, c) <= target) { cw = c; break; }
    if (cw == 0) {
        for (int c = 128; c >= 16; c -= 4)               // ... or two
            if (c <= a.C && bwd_tiled_lds_bytes(t.e_cap, t.n_cap, a.H, c) <= 80 * 1024) { cw = c; break; }
    }
    if (cw == 0) return false;                           // the wave-per-node kernel
    t.cw = cw;
    const size_t lds = bwd_tiled_lds_bytes(t.e_cap, t.n_cap, a.H, t.cw);
    switch (a.H) {
        case 1: return launch_bwd_dst_tiled_h<1>(t, g->num_graphs, lds, stream);
        case 2: return launch_bwd_dst_tiled_h<2>(t, g->num_graphs, lds, stream);
        case 4: return launch_bwd_dst_tiled_h<4>(t, g->num_graphs, lds, stream);
        default: return launch_bwd_dst_tiled_h<8>(t, g->num_graphs, lds, stream);
    }
}

template <int W, int KC>
static void launch_bwd(const BwdArgs& a, hipStream_t stream, bool dst_done) {
    const dim3 grid((unsigned)cdiv(a.N, 4));
    if (dst_done) {}
    else if (a.H <= 4) hipLaunchKernelGGL((k_gat_mp_bwd_dst<W, KC, 4>), grid, dim3(256), 0, stream, a);
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
    a.alpha = d->alpha; a.mask = d->alpha_mask; a.dsum = d->dalpha_node;
    a.dout = d->dout; a.dout_ld = d->dout_ld ? d->dout_ld : C;
    a.dxp = d->dxp; a.dxp_ld = d->dxp_ld ? d->dxp_ld : (int64_t)H * C;
    a.dxp_absmax = reinterpret_cast<unsigned*>(d->dxp_absmax);
    a.da_node = d->da_node; a.da_edge = d->da_edge;
    a.rowptr = g->rowptr; a.csr_src = g->csr_src; a.csr_eid = g->csr_eid;
    a.t_rowptr = gt->rowptr; a.t_csr_dst = gt->csr_src; a.t_csr_eid = gt->csr_eid;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer timer(GVQA_STAGE_MP, stream);
    if (a.dxp_absmax) GVQA_HIP_CHECK(hipMemsetAsync(a.dxp_absmax, 0, GVQA_ABSMAX_SLOTS * sizeof(unsigned), stream));
    const bool vec = C % 4 == 0 && a.xp_ld % 4 == 0 && a.dout_ld % 4 == 0 && a.dxp_ld % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(a.xp) | reinterpret_cast<uintptr_t>(a.dout) |
                       reinterpret_cast<uintptr_t>(a.dxp)) & 15) == 0;
    const bool dst_done = launch_bwd_dst_tiled(g, a, vec, stream);
    if (vec) {
        if (C <= 256) launch_bwd<4, 1>(a, stream, dst_done);
        else if (C <= 512) launch_bwd<4, 2>(a, stream, dst_done);
        else launch_bwd<4, 4>(a, stream, dst_done);
    } else {
        if (C <= 256) launch_bwd<1, 4>(a, stream, dst_done);
        else launch_bwd<1, 16>(a, stream, dst_done);
    }
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}
