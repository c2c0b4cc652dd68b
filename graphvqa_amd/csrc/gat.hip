// GAT execution path for gfx950: weight folding, the fused message-passing kernels and the
// K-hop driver behind gvqa_gat_conv_forward / gvqa_gat_seq_forward.
//
// Reference being replaced: gat_skip.py:111-208 (gat.forward + message, aggregated by PyG's
// MessagePassing.propagate / torch_scatter) and gat_skip.py:249-279 (gat_seq.forward).
//
// What is restructured relative to the reference (same math, fp32, re-associated sums):
//   * a_l/a_r/a_e (gat_skip.py:134-135,150-151) are dot products of a projection with a fixed
//     attention vector, i.e. x . (W^T att).  The folded vectors V = W^T att are computed once
//     per forward from the weights (k_fold_*), which turns the E x (De+Di) x H*C edge GEMM
//     (two thirds of the reference's FLOPs, its result is used for nothing else) into an
//     E x De x H mat-vec, done for all K hops in ONE pass over edge_attr.
//   * the [h || ins[batch]] / [edge_attr || ins[batch[src]]] concatenations (gat_skip.py:256-264)
//     are never materialised: W = [W_h | W_ins] splits the projection into a per-node part and
//     a per-GRAPH part.  Because softmax weights sum to one per destination and edges never
//     leave their graph, the per-graph part passes through the aggregation unchanged and
//     reduces to one [B, C] row (mean over heads) plus one logit offset per (graph, head).
//   * gather (K4), logits (K5), segment softmax (K6), weighted scatter-add (K8), head mean +
//     bias (K9), skip + BatchNorm(eval) + ReLU (K11) are ONE kernel per hop; nothing of size
//     E x H x C ever reaches HBM (the reference writes and re-reads it three times).
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include "common.h"

namespace gvqa {

constexpr int MAX_HOPS = 8;

// ==============================================================================================
// Weight folding
// ==============================================================================================
struct FoldArgs {
    const float* W_l[MAX_HOPS];
    const float* W_e[MAX_HOPS];
    const float* att_l[MAX_HOPS];
    const float* att_r[MAX_HOPS];
    const float* att_e[MAX_HOPS];
    float* Vn;    // [K][2H][Dn]
    float* Ve;    // [K*H][De]
    float* Gw;    // [K][C+H][Di]   (rows C.. : logit offsets)   may be NULL when Di == 0
    int Dn, De, Di, C, H, K;
};

// out[h, k] = sum_c W[(h*C + c), koff + k] * att[h*C + c]      (V = W^T att, per head)
// grid: (ceil(maxK/64), H, K*4): z = hop*4 + which; which 0: V_l -> Vn rows [0,H); 1: V_r -> Vn rows
// [H,2H); 2: V_e -> Ve; 3: instruction-column part of (V_l + V_r + V_e) -> Gw rows C + h.
__global__ __launch_bounds__(256) void k_fold_att(FoldArgs a) {
    __shared__ float red[4][64];
    const int hop = blockIdx.z >> 2, which = blockIdx.z & 3, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const int ldl = a.Dn + a.Di, lde = a.De + a.Di;
    int nk;
    if (which < 2) nk = a.Dn; else if (which == 2) nk = a.De; else nk = a.Di;
    if (blockIdx.x * 64 >= nk) return;
    // the C-range is split over the 4 waves, partial sums combined in a fixed order
    const int cpw = (a.C + 3) / 4, c_lo = wave * cpw, c_hi = min(a.C, c_lo + cpw);
    float acc = 0.f;
    if (k < nk) {
        if (which < 2) {
            const float* W = a.W_l[hop];
            const float* att = which == 0 ? a.att_l[hop] : a.att_r[hop];
#pragma unroll 8
            for (int c = c_lo; c < c_hi; ++c) acc += W[(int64_t)(h * a.C + c) * ldl + k] * att[h * a.C + c];
        } else if (which == 2) {
            const float* W = a.W_e[hop];
            const float* att = a.att_e[hop];
#pragma unroll 8
            for (int c = c_lo; c < c_hi; ++c) acc += W[(int64_t)(h * a.C + c) * lde + k] * att[h * a.C + c];
        } else {
            const float* Wl = a.W_l[hop];
            const float* We = a.W_e[hop];
#pragma unroll 4
            for (int c = c_lo; c < c_hi; ++c) {
                const int r = h * a.C + c;
                const float wl = Wl[(int64_t)r * ldl + a.Dn + k];
                acc += wl * a.att_l[hop][r] + wl * a.att_r[hop][r] + We[(int64_t)r * lde + a.De + k] * a.att_e[hop][r];
            }
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && k < nk) {
        float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (which == 0) a.Vn[((int64_t)hop * 2 * a.H + h) * a.Dn + k] = s;
        else if (which == 1) a.Vn[((int64_t)hop * 2 * a.H + a.H + h) * a.Dn + k] = s;
        else if (which == 2) a.Ve[((int64_t)hop * a.H + h) * a.De + k] = s;
        else a.Gw[((int64_t)hop * (a.C + a.H) + a.C + h) * a.Di + k] = s;
    }
}

// Gw[hop][c, k] = (1/H) sum_h W_l[h*C + c, Dn + k]     (head-mean of the instruction columns)
__global__ __launch_bounds__(256) void k_fold_headmean(FoldArgs a) {
    const int hop = blockIdx.z;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.C * a.Di) return;
    const int c = (int)(idx / a.Di), k = (int)(idx - (int64_t)c * a.Di);
    const int ldl = a.Dn + a.Di;
    const float* W = a.W_l[hop];
    float s = 0.f;
    for (int h = 0; h < a.H; ++h) s += W[(int64_t)(h * a.C + c) * ldl + a.Dn + k];
    a.Gw[((int64_t)hop * (a.C + a.H) + c) * a.Di + k] = s * (1.0f / a.H);
}

// ==============================================================================================
// Fused message passing
// ==============================================================================================
struct MpArgs {
    const int32_t* rowptr;
    const int32_t* csr_src;
    const int32_t* csr_eid;
    const int32_t* node_graph;
    const int32_t* graph_ptr;
    const float* xp;        // [N, xp_ld]: head h of node n at xp[n*xp_ld + h*C .. + C)
    int64_t xp_ld;
    const float* a_node;    // [N, 2H] or NULL (zeros)
    int a_node_parts;       // 0 / 1: a_node as it is; P > 1: a_node = sum of P partial arrays, part p at a_node + p * a_node_part_stride
    int64_t a_node_part_stride;   // (the chained hop kernel leaves one partial logit array per column block, hop2.hip)
    const float* a_edge;    // COO-indexed: a_edge[eid * a_edge_stride + h]
    int64_t a_edge_stride;
    const float* graph_term;  // NULL or [B, t_ld]: columns [0,C) head-mean instruction term, [C,C+H) logit offset
    int64_t t_ld;
    const float* graph_scale; // NULL or [B, gs_ld]: per-graph channel scale of the head mean (LCGN cal_cmd)
    int64_t gs_ld;
    const float* skip;        // NULL or [N, skip_ld]
    int64_t skip_ld;
    const float* bias;        // NULL or [C]
    const float* bn_w;        // all four NULL = no BN / ReLU
    const float* bn_b;
    const float* bn_m;
    const float* bn_v;
    float* out;               // [N, out_ld]
    int64_t out_ld;
    float* alpha_out;         // NULL or [E, H] COO order (the softmax output, before the mask)
    const float* alpha_mask;  // NULL or [E, H] COO order: multiplies alpha after the softmax (attention dropout)
    float* alpha_csr;         // general kernel only: [E, H] in CSR slot order
    const float* head_rows;   // tiled kernel: NULL or [B, hr_ld] per-graph rows added per HEAD, weighted by the node's coefficient sums (below)
    int64_t hr_ld;
    float* head_weight_out;   // NULL or [N, H]: s[i, h] = sum over the in-edges of alpha * mask (written with head_rows; the backward's operand)
    int N, C, cw;             // cw: channel chunk width handled by one block (tiled kernel)
    int e_cap, n_cap;         // LDS capacity in edges / nodes per graph (tiled kernel)
    int nbuf;                 // stage buffers in LDS (prefetch depth = nbuf - 1)
    int nparts;               // blocks per graph: block (g, part) owns the channel ranges [part, part+1) * nch / nparts
    int lpn_log;              // log2(lanes per node) in the aggregation mapping
    float slope, bn_eps;
    int debug;                // measurement aid (GVQA_MP_DEBUG bit mask, tiled kernel): 1 no aggregation loops, 2 no softmax passes,
};                            // 4 no output stores, 8 no CSR / logit / constant loads in the prologue -- wrong results; scripts/bench_mp_plan.py prices the parts with them

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// ablation switches of the tiled kernel: measurement build only
#ifdef GVQA_PROBES
#define GVQA_MP_DBG(bit_) (a.debug & (bit_))
#else
#define GVQA_MP_DBG(bit_) false
#endif

// node logit a_node[idx] (idx = node * 2H + j), the partial arrays added in part order
__device__ __forceinline__ float a_node_at(const MpArgs& a, int64_t idx) {
    float v = a.a_node[idx];
    for (int p = 1; p < a.a_node_parts; ++p) v += a.a_node[idx + (int64_t)p * a.a_node_part_stride];
    return v;
}
__device__ __forceinline__ float4 a_node_at4(const MpArgs& a, int64_t idx) {
    float4 v = *reinterpret_cast<const float4*>(a.a_node + idx);
    for (int p = 1; p < a.a_node_parts; ++p) {
        const float4 w = *reinterpret_cast<const float4*>(a.a_node + idx + (int64_t)p * a.a_node_part_stride);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    return v;
}

// bias + skip + BatchNorm(eval) + ReLU on 4 consecutive channels (gat_skip.py:168,270,273-275)
__device__ __forceinline__ float4 mp_epilogue(const MpArgs& a, float4 r, int node, int c) {
    if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + c);
        r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w;
    }
    if (a.skip) {
        const float4 s = *reinterpret_cast<const float4*>(a.skip + (int64_t)node * a.C + c);
        r.x += s.x; r.y += s.y; r.z += s.z; r.w += s.w;
    }
    if (a.bn_w) {
        const float4 m = *reinterpret_cast<const float4*>(a.bn_m + c);
        const float4 v = *reinterpret_cast<const float4*>(a.bn_v + c);
        const float4 w = *reinterpret_cast<const float4*>(a.bn_w + c);
        const float4 b = *reinterpret_cast<const float4*>(a.bn_b + c);
        r.x = fmaxf((r.x - m.x) * (1.0f / sqrtf(v.x + a.bn_eps)) * w.x + b.x, 0.f);
        r.y = fmaxf((r.y - m.y) * (1.0f / sqrtf(v.y + a.bn_eps)) * w.y + b.y, 0.f);
        r.z = fmaxf((r.z - m.z) * (1.0f / sqrtf(v.z + a.bn_eps)) * w.z + b.z, 0.f);
        r.w = fmaxf((r.w - m.w) * (1.0f / sqrtf(v.w + a.bn_eps)) * w.w + b.w, 0.f);
    }
    return r;
}

// ----------------------------------------------------------------------------------------------
// LDS-tiled streaming kernel: one block = one graph.
//
// Scene graphs are small and the batch adjacency is block-diagonal, so every neighbour of a
// node lives in the same graph.  The block
//   1. computes the attention coefficients of the graph's edges ONCE (logits -> leaky-relu ->
//      softmax over incoming edges), kept in LDS together with the local CSR;
//   2. walks the graph's slab xp[n0:n1, :, :] in stages of (head h, channel range cr): each stage
//      is an n x cw tile (row segments of cw*4 contiguous bytes) DMA'd HBM -> LDS with
//      global_load_lds (no VGPR round trip, 16 B per lane), double buffered: stage t+1 is in
//      flight while stage t is aggregated out of LDS.  Every byte of xp is read from HBM exactly
//      once; the E x H gathers of K4/K8 are LDS reads; nothing of size E x H x C exists anywhere.
//   3. after the H head stages of a channel range, a final stage brings the skip rows h[n, cr]
//      and the epilogue (head mean, per-graph instruction term, bias, skip, BN, ReLU) writes
//      out[n0:n1, cr] with 16-byte stores.
// Inside the stage loop there are no ordinary global loads (they would make the compiler drain
// the DMA queue early): CSR, alpha and the per-channel epilogue constants all live in LDS.
// grid = B.  dynamic LDS = [alpha e_cap*H][src e_cap][rowptr n_cap+1][consts 4*C][2 stage buffers].
// ----------------------------------------------------------------------------------------------
// quads (4 slots) of a graph's padded edge table: every node rounds up to whole quads (<= 3 pad slots each) + one all-zero quad
__host__ __device__ inline int mp_padded_quads(int e_cap, int n_cap) { return (e_cap + 3 * n_cap) / 4 + 2; }
constexpr int MP_THREADS = 512;  // 8 waves: more LDS-latency hiding per resident graph
constexpr int MP_ITEMS = 4;      // float4 accumulators per thread

typedef __attribute__((address_space(3))) char* lds_ptr_t;

// LDS-DMA: 16 bytes per lane from `gsrc` (per lane) to LDS byte address `lds_dst` + lane * 16
// (`lds_dst` wave-uniform, goes through M0).  Inline asm on purpose: a DMA issued through the
// builtin is tracked by hipcc as a pending LDS write and it then waits vmcnt(0) before every
// ds_read of the stage buffers, which serialises the pipeline; issued here it is invisible to the
// compiler and ordered by the counted s_waitcnt + barrier of the stage loop instead.
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

template <int H, int ITEMS, bool HR = false>
__global__ __launch_bounds__(MP_THREADS, 6) void k_gat_mp_tiled(MpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t off_src = (size_t)a.e_cap * H * 4;
    const size_t off_row = (off_src + (size_t)a.e_cap * 4 + 15) & ~(size_t)15;
    // padded edge table (built once per graph, read by every stage): every node's in-edge list padded to whole QUADS -- 4 slots --
    // with zero-weight entries, coefficients head-major, so that a trip of the aggregation loop is two 16-byte LDS reads (4 source
    // rows, 4 coefficients of this stage's head) + 4 row reads + 16 FMAs and nothing else (before: 12 scalar LDS reads, clamps,
    // masks and address arithmetic per trip -- ~56 VALU instructions for 16 FMAs).  The last quad is an all-zero one: lanes whose
    // node has fewer quads than the wave's longest read it instead (wave-uniform trip count, no divergence).
    const int EPQ = mp_padded_quads(a.e_cap, a.n_cap);
    const size_t off_ps = (off_row + (size_t)(a.n_cap + 1) * 4 + 15) & ~(size_t)15;
    const size_t off_ap = (off_ps + (size_t)(a.n_cap + 1) * 4 + 15) & ~(size_t)15;
    const size_t off_sq = off_ap + (size_t)H * EPQ * 16;
    const size_t off_cst = off_sq + (size_t)EPQ * 16;
    const size_t off_hr = off_cst + (size_t)5 * a.C * 4;                       // [H][C] per-graph head rows (head_rows only)
    const size_t off_buf = off_hr + (a.head_rows ? (size_t)H * a.C * 4 : 0);
    float* alpha_s = reinterpret_cast<float*>(smem);
    int* src_l = reinterpret_cast<int*>(smem + off_src);
    int* rowp_l = reinterpret_cast<int*>(smem + off_row);
    int* pst_l = reinterpret_cast<int*>(smem + off_ps);         // [tn + 1] first quad of node i's padded list
    float* alpha_p = reinterpret_cast<float*>(smem + off_ap);   // [H][4 EPQ]
    int* srcq = reinterpret_cast<int*>(smem + off_sq);          // [4 EPQ] local source row of every padded slot
    float* cst = reinterpret_cast<float*>(smem + off_cst);      // [pbar | bias | scale | shift | gscale][C]
    float* hrow = reinterpret_cast<float*>(smem + off_hr);
    // stage buffers hold whole DMA rounds of MP_THREADS units
    const size_t buf_bytes = (((size_t)a.n_cap * (a.cw >> 2) + MP_THREADS - 1) / MP_THREADS) * MP_THREADS * 16;
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;

    const int g = blockIdx.x, part = blockIdx.y;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1];
    const int tn = n1 - n0;
    if (tn <= 0) return;
    const int e0 = a.rowptr[n0], ne = a.rowptr[n1] - e0;
    const int tid = threadIdx.x;
    const int wave_unit0 = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int C = a.C;
    const int nch_all = (C + a.cw - 1) / a.cw;
    // this block's share of the channel ranges (small batches: several blocks per graph, each repeating the cheap
    // alpha prologue, so that the chip is filled and the last block wave is fine-grained)
    const int r_lo = part * nch_all / a.nparts, r_hi = (part + 1) * nch_all / a.nparts;
    const int nch = r_hi - r_lo;
    if (nch <= 0) return;
    const int spc = H + (a.skip ? 1 : 0);            // stages per channel range
    const int T = nch * spc;
    const int q4cap = a.cw >> 2;
    const int dma_round_rows = MP_THREADS / q4cap, dma_round_cols = MP_THREADS - dma_round_rows * q4cap;
    const int dma_row0 = tid / q4cap, dma_col0 = tid - dma_row0 * q4cap;   // unit `tid` of a full-width range

    // DMA of one stage -- head j (or the skip rows when j == H) of the channel range starting at
    // c0 -- as a [tn x q4c] float4 tile (row segments of q4c*16 contiguous bytes) into stage buffer
    // `bufi`.  Every wave issues the same number of DMAs per stage (lanes past the end re-load the
    // last unit into the buffer's padding), so the counted vmcnt below is exact.
    // All stage bookkeeping is incremental (adds and compares): the scalar unit is shared by every
    // wave of the CU and runtime integer divisions here made it the kernel's bottleneck.
    auto prefetch = [&](int j, int c0, int bufi) {
        const int q4c = min(a.cw, C - c0) >> 2;
        const int units = tn * q4c;
        const float* base;
        int64_t row_stride;
        if (j < H) { base = a.xp + (int64_t)n0 * a.xp_ld + (int64_t)j * C + c0; row_stride = a.xp_ld; }
        else       { base = a.skip + (int64_t)n0 * a.skip_ld + c0;             row_stride = a.skip_ld; }
        unsigned dst = lds_base + (unsigned)off_buf + (unsigned)bufi * (unsigned)buf_bytes + (unsigned)wave_unit0 * 16u;
        int row, col, rstep, cstep;
        if (q4c == q4cap) { row = dma_row0; col = dma_col0; rstep = dma_round_rows; cstep = dma_round_cols; }
        else { row = tid / q4c; col = tid - row * q4c; rstep = MP_THREADS / q4c; cstep = MP_THREADS - rstep * q4c; }
        const int last_row = tn - 1, last_col = q4c - 1;
        for (int u0 = 0; u0 < units; u0 += MP_THREADS) {
            const bool in = row < tn;
            const int r = in ? row : last_row, c = in ? col : last_col;
            lds_dma16(base + r * row_stride + c * 4, __builtin_amdgcn_readfirstlane(dst));
            dst += MP_THREADS * 16;
            row += rstep; col += cstep;
            if (col >= q4c) { col -= q4c; ++row; }
        }
    };
    // advance a (j, c0, buffer) stage cursor by one stage
    auto advance = [&](int& j, int& c0, int& bufi) {
        if (++j == spc) { j = 0; c0 += a.cw; }
        if (++bufi == a.nbuf) bufi = 0;
    };

    const int depth = a.nbuf - 1;
    int pf_j = 0, pf_c0 = r_lo * a.cw, pf_buf = 0, pf_t = 0;      // cursor of the next stage to prefetch
    for (; pf_t < depth && pf_t < T; ++pf_t) { prefetch(pf_j, pf_c0, pf_buf); advance(pf_j, pf_c0, pf_buf); }

    // ---- prologue: local CSR, destination-independent logit terms, epilogue constants ----
    for (int s = tid; s < (GVQA_MP_DBG(8) ? 0 : ne); s += MP_THREADS) {
        const int src = a.csr_src[e0 + s];
        const int eid = a.csr_eid[e0 + s];
        src_l[s] = src - n0;
        const float* ae = a.a_edge + (int64_t)eid * a.a_edge_stride;
#pragma unroll
        for (int h = 0; h < H; ++h) alpha_s[s * H + h] = (a.a_node ? a.a_node[(int64_t)src * 2 * H + h] : 0.f) + ae[h];
    }
    for (int i = tid; i <= tn; i += MP_THREADS) rowp_l[i] = a.rowptr[n0 + i] - e0;
    for (int c = tid; c < (GVQA_MP_DBG(8) ? 0 : C); c += MP_THREADS) {
        cst[c] = a.graph_term ? a.graph_term[(int64_t)g * a.t_ld + c] : 0.f;
        cst[C + c] = a.bias ? a.bias[c] : 0.f;
        float sc = 1.f, sh = 0.f;
        if (a.bn_w) {   // torch's eval BatchNorm: y = x * (w * invstd) + (b - mean * w * invstd)
            const float invstd = 1.0f / sqrtf(a.bn_v[c] + a.bn_eps);
            sc = a.bn_w[c] * invstd;
            sh = a.bn_b[c] - a.bn_m[c] * sc;
        }
        cst[2 * C + c] = sc;
        cst[3 * C + c] = sh;
        cst[4 * C + c] = a.graph_scale ? a.graph_scale[(int64_t)g * a.gs_ld + c] : 1.f;
    }
    if (a.head_rows)
        for (int c = tid; c < H * C; c += MP_THREADS) hrow[c] = a.head_rows[(int64_t)g * a.hr_ld + c];
    for (int u = tid; u < (H + 1) * EPQ; u += MP_THREADS)      // padded table: all slots zero weight / row 0 first (alpha_p | srcq are contiguous)
        reinterpret_cast<float4*>(alpha_p)[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tid < 64) {     // quad starts: exclusive scan of ceil(deg / 4) over the graph's nodes (one wave: chunks per lane, then a wave scan)
        const int per = (tn + 63) >> 6, b0 = min(tid * per, tn), b1 = min(b0 + per, tn);
        int sum = 0;
        for (int i = b0; i < b1; ++i) sum += (rowp_l[i + 1] - rowp_l[i] + 3) >> 2;
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if (tid >= o) inc += v;
        }
        int run = inc - sum;
        for (int i = b0; i < b1; ++i) { pst_l[i] = run; run += (rowp_l[i + 1] - rowp_l[i] + 3) >> 2; }
        if (tid == 63) pst_l[tn] = inc;
    }
    __syncthreads();
    // ---- leaky-relu + softmax over the incoming edges of each (node, head) ----
    for (int it = tid; it < (GVQA_MP_DBG(2) ? 0 : tn * H); it += MP_THREADS) {
        const int i = it / H, h = it - i * H;
        const int lo = rowp_l[i], hi = rowp_l[i + 1];
        float ar = a.a_node ? a.a_node[(int64_t)(n0 + i) * 2 * H + H + h] : 0.f;
        if (a.graph_term) ar += a.graph_term[(int64_t)g * a.t_ld + C + h];
        float m = -INFINITY;
        for (int s = lo; s < hi; ++s) {
            const float v = leaky(alpha_s[s * H + h] + ar, a.slope);
            alpha_s[s * H + h] = v;
            m = fmaxf(m, v);
        }
        float sum = 0.f;
        for (int s = lo; s < hi; ++s) {
            const float ex = expf(alpha_s[s * H + h] - m);
            alpha_s[s * H + h] = ex;
            sum += ex;
        }
        const float den = sum + 1e-16f;
        for (int s = lo; s < hi; ++s) {
            float al = alpha_s[s * H + h] / den;
            if (a.alpha_out && part == 0) a.alpha_out[(int64_t)a.csr_eid[e0 + s] * H + h] = al;
            if (a.alpha_mask) al *= a.alpha_mask[(int64_t)a.csr_eid[e0 + s] * H + h];
            const int ps = 4 * pst_l[i] + (s - lo);            // the slot of this edge in the padded table
            alpha_p[h * 4 * EPQ + ps] = al;
            if (h == 0) srcq[ps] = src_l[s];
        }
    }

    // ---- stage loop ----
    // Work item = (node i, float4 column q).  A node's columns sit on `lpn` consecutive lanes
    // (power of two >= cw/4, so the (i, q) split is shifts only); a thread walks nodes
    // i = tid / lpn + k * (MP_THREADS / lpn).
    const int lpn_log = a.lpn_log, lpn = 1 << lpn_log;
    const int q = tid & (lpn - 1), i_base = tid >> lpn_log, i_step = MP_THREADS >> lpn_log;
    float4 acc[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_h = 1.0f / H;
    const bool relu = a.bn_w != nullptr;

    // DMA instructions per wave for a stage of a full-width / of the last (narrower) channel range
    const int dma_full = (tn * q4cap + MP_THREADS - 1) / MP_THREADS;
    const int dma_last = (tn * ((C - (nch_all - 1) * a.cw) >> 2) + MP_THREADS - 1) / MP_THREADS;
    const int last_c0 = (nch_all - 1) * a.cw;

    // Per-item row extents and the wave-uniform trip count of the 4-wide edge loop are the same in
    // every stage: computed once per graph.  (Wave-uniform trips + branch-free body: a divergent
    // loop costs ~8 scalar instructions of exec-mask handling per trip.)
    int it_q0[ITEMS], it_own[ITEMS], it_trips[ITEMS];       // first quad / own quads / the wave's longest list, per item
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // the padded table is complete before it is read below
    const int zq = EPQ - 1;                 // the all-zero quad
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int i = i_base + k * i_step;
        const bool valid = q < q4cap && i < tn;
        const int ii = min(i, tn - 1);
        it_q0[k] = valid ? pst_l[ii] : zq;
        it_own[k] = valid ? pst_l[ii + 1] - pst_l[ii] : 0;
        int trips = it_own[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) trips = max(trips, __shfl_xor(trips, o, 64));
        it_trips[k] = GVQA_MP_DBG(1) ? 0 : __builtin_amdgcn_readfirstlane(trips);
    }

    int cur_j = 0, cur_c0 = r_lo * a.cw, cur_buf = 0;     // cursor of the stage being consumed
    for (int t = 0; t < T; ++t) {
        // Wait until stage t has landed: this wave's DMAs of stage t are older than those of stages
        // t+1 .. t+depth-1, loads retire in order, so "at most N outstanding" with N = the younger
        // stages' DMA count covers it (output stores, if any, only make the wait stricter).  Then a
        // raw barrier (no compiler fence, which would drain the whole DMA queue): every wave's
        // part of stage t is in LDS, and everybody is done computing stage t-1, whose buffer the
        // prefetch of stage t+depth overwrites.
        if (depth <= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            int allow = 0, jj = cur_j, cc = cur_c0, bb = cur_buf;
            for (int k = 1; k < depth && t + k < T; ++k) {
                advance(jj, cc, bb);
                allow += (cc == last_c0) ? dma_last : dma_full;
            }
            switch (allow) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (pf_t < T) { prefetch(pf_j, pf_c0, pf_buf); advance(pf_j, pf_c0, pf_buf); ++pf_t; }

        const int j = cur_j, c0 = cur_c0;
        const int q4c = min(a.cw, C - c0) >> 2;
        const float4* buf4 = reinterpret_cast<const float4*>(smem + off_buf + (size_t)cur_buf * buf_bytes);
        const bool lane_on = q < q4c;
        const int qq = lane_on ? q : 0;
        if (j < H) {
            const int4* sq4 = reinterpret_cast<const int4*>(srcq);
            const float4* ap4 = reinterpret_cast<const float4*>(alpha_p) + j * EPQ;       // this stage's head
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                const int q0 = it_q0[k], own = it_own[k], trips = it_trips[k];
                float4 s4 = acc[k];
                float wsum = 0.f;                               // (head_rows: the node's coefficient sum of this head)
                for (int tr = 0; tr < trips; ++tr) {
                    const int pq = tr < own ? q0 + tr : zq;     // (past this node's list: the all-zero quad)
                    const int4 sl = sq4[pq];
                    const float4 al = ap4[pq];
                    if (HR) wsum += (al.x + al.y) + (al.z + al.w);
                    const float4 v0 = buf4[sl.x * q4c + qq], v1 = buf4[sl.y * q4c + qq], v2 = buf4[sl.z * q4c + qq], v3 = buf4[sl.w * q4c + qq];
                    s4.x += al.x * v0.x; s4.y += al.x * v0.y; s4.z += al.x * v0.z; s4.w += al.x * v0.w;
                    s4.x += al.y * v1.x; s4.y += al.y * v1.y; s4.z += al.y * v1.z; s4.w += al.y * v1.w;
                    s4.x += al.z * v2.x; s4.y += al.z * v2.y; s4.z += al.z * v2.z; s4.w += al.z * v2.w;
                    s4.x += al.w * v3.x; s4.y += al.w * v3.y; s4.z += al.w * v3.z; s4.w += al.w * v3.w;
                }
                if (HR) {
                    // the per-graph row of this head rides on the node's coefficient sum: MP(xp + rows[graph]) without forming the sum
                    // (gat_skip.py:133,263-264: the instruction half of lin_l).  No mask: the sum is 1 (0 without in-edges), exactly
                    if (!a.alpha_mask) wsum = own > 0 ? 1.f : 0.f;
                    const float4 hr = *reinterpret_cast<const float4*>(hrow + j * C + c0 + qq * 4);
                    s4.x += wsum * hr.x; s4.y += wsum * hr.y; s4.z += wsum * hr.z; s4.w += wsum * hr.w;
                    const int i = i_base + k * i_step;
                    if (a.head_weight_out && q == 0 && i < tn && part == 0 && c0 == 0)
                        a.head_weight_out[(int64_t)(n0 + i) * H + j] = wsum;
                }
                acc[k] = s4;
            }
        }
        if (j == spc - 1 && lane_on) {      // last stage of this channel range: epilogue + store
            // per-channel epilogue constants live in LDS: no ordinary global load inside the stage loop
            const int c = c0 + q * 4;
            const float4 pb = *reinterpret_cast<const float4*>(cst + c);
            const float4 bi = *reinterpret_cast<const float4*>(cst + C + c);
            const float4 sc = *reinterpret_cast<const float4*>(cst + 2 * C + c);
            const float4 sh = *reinterpret_cast<const float4*>(cst + 3 * C + c);
            const float4 gs = *reinterpret_cast<const float4*>(cst + 4 * C + c);
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                const int i = i_base + k * i_step;
                if (i < tn) {
                    const bool has_edges = it_own[k] > 0;
                    float4 r = make_float4(acc[k].x * inv_h, acc[k].y * inv_h, acc[k].z * inv_h, acc[k].w * inv_h);
                    if (a.graph_scale) { r.x *= gs.x; r.y *= gs.y; r.z *= gs.z; r.w *= gs.w; }
                    if (has_edges) { r.x += pb.x; r.y += pb.y; r.z += pb.z; r.w += pb.w; }
                    r.x += bi.x; r.y += bi.y; r.z += bi.z; r.w += bi.w;
                    if (a.skip) {
                        const float4 sk = buf4[i * q4c + q];
                        r.x += sk.x; r.y += sk.y; r.z += sk.z; r.w += sk.w;
                    }
                    if (relu) {
                        r.x = fmaxf(r.x * sc.x + sh.x, 0.f); r.y = fmaxf(r.y * sc.y + sh.y, 0.f);
                        r.z = fmaxf(r.z * sc.z + sh.z, 0.f); r.w = fmaxf(r.w * sc.w + sh.w, 0.f);
                    }
                    // (holding the rows back and storing them one stage later, behind the next barrier and prefetch, so that the
                    //  vmcnt(0) at the top of the next stage does not wait for them: 201 vs 201 us at config 3 -- no gain)
                    if (!GVQA_MP_DBG(4)) *reinterpret_cast<float4*>(a.out + (int64_t)(n0 + i) * a.out_ld + c) = r;
                }
            }
        }
        if (j == spc - 1) {
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        advance(cur_j, cur_c0, cur_buf);
    }
}

// ----------------------------------------------------------------------------------------------
// General CSR kernels (any graph: huge graphs, inter-graph edges, C % 4 != 0).
//   k_gat_alpha_general: one thread per (node, head): three passes over the node's in-edges.
//   k_gat_aggregate_general: one wave per node, lanes stride the channels, gathers from global.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gat_alpha_general(MpArgs a, int H) {
    const int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (it >= (int64_t)a.N * H) return;
    const int i = (int)(it / H), h = (int)(it - (int64_t)i * H);
    const int lo = a.rowptr[i], hi = a.rowptr[i + 1];
    float ar = a.a_node ? a_node_at(a, (int64_t)i * 2 * H + H + h) : 0.f;
    if (a.graph_term) ar += a.graph_term[(int64_t)a.node_graph[i] * a.t_ld + a.C + h];
    float m = -INFINITY;
    for (int s = lo; s < hi; ++s) {
        const float v = leaky((a.a_node ? a_node_at(a, (int64_t)a.csr_src[s] * 2 * H + h) : 0.f) +
                              a.a_edge[(int64_t)a.csr_eid[s] * a.a_edge_stride + h] + ar, a.slope);
        a.alpha_csr[(int64_t)s * H + h] = v;
        m = fmaxf(m, v);
    }
    float sum = 0.f;
    for (int s = lo; s < hi; ++s) {
        const float ex = expf(a.alpha_csr[(int64_t)s * H + h] - m);
        a.alpha_csr[(int64_t)s * H + h] = ex;
        sum += ex;
    }
    const float den = sum + 1e-16f;
    for (int s = lo; s < hi; ++s) {
        float al = a.alpha_csr[(int64_t)s * H + h] / den;
        if (a.alpha_out) a.alpha_out[(int64_t)a.csr_eid[s] * H + h] = al;
        if (a.alpha_mask) al *= a.alpha_mask[(int64_t)a.csr_eid[s] * H + h];
        a.alpha_csr[(int64_t)s * H + h] = al;
    }
}

// The same coefficients, one block per ROW GROUP of the fused hop (<= 128 consecutive nodes, whole graphs): phase 1 is parallel
// over the group's CSR slots -- index loads and the two gathers of every slot are independent, nothing is read back from global
// memory -- and leaves a_l[src] + a_e[eid] in LDS; phase 2, one thread per (node, head), runs the three softmax passes out of
// LDS.  Same operations in the same order as k_gat_alpha_general: bit-identical coefficients, without its three dependent
// trips to L2 per edge (26 -> 17 us per hop at config 3, 15 -> 12 on a 256-graph shard, launch gaps included; starting the
// phase-2 operand loads before phase 1 changed nothing).
__global__ __launch_bounds__(256) void k_gat_alpha_groups(MpArgs a, int H, const int32_t* __restrict__ group_ptr) {
    extern __shared__ float raw_s[];                        // [slots of the group][H]
    const int tid = threadIdx.x;
    const int ns = group_ptr[blockIdx.x], cnt = group_ptr[blockIdx.x + 1] - ns;
    const int e0 = a.rowptr[ns], ne = a.rowptr[ns + cnt] - e0;
    const bool v4 = (H & 3) == 0 && (a.a_edge_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(a.a_edge) & 15) == 0 &&
                    (!a.a_node || ((reinterpret_cast<uintptr_t>(a.a_node) & 15) == 0 && (a.a_node_part_stride & 3) == 0));
    for (int s = tid; s < ne; s += 256) {
        const int src = a.csr_src[e0 + s], eid = a.csr_eid[e0 + s];
        const bool an = a.a_node != nullptr;
        const int64_t an0 = (int64_t)src * 2 * H;
        const float* ae = a.a_edge + (int64_t)eid * a.a_edge_stride;
        if (v4) {
            for (int h = 0; h < H; h += 4) {
                const float4 x = an ? a_node_at4(a, an0 + h) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 y = *reinterpret_cast<const float4*>(ae + h);
                *reinterpret_cast<float4*>(raw_s + s * H + h) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
            }
        } else {
            for (int h = 0; h < H; ++h) raw_s[s * H + h] = (an ? a_node_at(a, an0 + h) : 0.f) + ae[h];
        }
    }
    __syncthreads();
    for (int it = tid; it < cnt * H; it += 256) {
        const int i = it / H, h = it - i * H, node = ns + i;
        const int lo = a.rowptr[node] - e0, hi = a.rowptr[node + 1] - e0;
        float ar = a.a_node ? a_node_at(a, (int64_t)node * 2 * H + H + h) : 0.f;
        if (a.graph_term) ar += a.graph_term[(int64_t)a.node_graph[node] * a.t_ld + a.C + h];
        float m = -INFINITY;
        for (int s = lo; s < hi; ++s) {
            const float v = leaky(raw_s[s * H + h] + ar, a.slope);
            raw_s[s * H + h] = v;
            m = fmaxf(m, v);
        }
        float sum = 0.f;
        for (int s = lo; s < hi; ++s) {
            const float ex = expf(raw_s[s * H + h] - m);
            raw_s[s * H + h] = ex;
            sum += ex;
        }
        const float den = sum + 1e-16f;
        for (int s = lo; s < hi; ++s) {
            float al = raw_s[s * H + h] / den;
            if (a.alpha_out) a.alpha_out[(int64_t)a.csr_eid[e0 + s] * H + h] = al;
            if (a.alpha_mask) al *= a.alpha_mask[(int64_t)a.csr_eid[e0 + s] * H + h];
            a.alpha_csr[(int64_t)(e0 + s) * H + h] = al;
        }
    }
}

// attention coefficients of every edge in CSR slot order -> a.alpha_csr (and a.alpha_out in COO order)
// (one thread per (node, head), three passes through alpha_csr.  Measured SLOWER at config 3: one thread per node with all heads
// in registers and 16-byte accesses -- 31 vs 25 us, a quarter of the threads; logits of short rows kept in registers, one
// round of 8 clamped gathers instead of three dependent passes -- 34 vs 27 us)
static int launch_alpha(const MpArgs& a, int H, hipStream_t stream, const gvqa_graph* g = nullptr) {
    // row groups planned and the largest group's logits within 64 KiB of LDS: the group kernel
    if (g && g->num_row_groups > 0 && g->row_group_ptr && (size_t)g->max_row_group_edges * H * sizeof(float) <= 64 * 1024 &&
        get_option(GVQA_OPT_COEFF_KERNEL) == 0) {
        const size_t lds = std::max<size_t>((size_t)g->max_row_group_edges * H * sizeof(float), 16);
        hipLaunchKernelGGL(k_gat_alpha_groups, dim3((unsigned)g->num_row_groups), dim3(256), lds, stream, a, H, g->row_group_ptr);
        GVQA_LAUNCH_CHECK();
        return GVQA_OK;
    }
    hipLaunchKernelGGL(k_gat_alpha_general, dim3((unsigned)cdiv((int64_t)a.N * H, 256)), dim3(256), 0, stream, a, H);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// Chained hops (hop2.hip): the previous hop left h as PACKED rows (two fp16 pieces by row-group slot) and nothing else, so this
// form computes the node logits a_node = h . [V_l | V_r] itself -- wave w takes row tile w of the group through all k blocks on
// the fp16 matrix cores against the two-piece image of the folded vectors (the arithmetic of the pack pass's logits), results
// in LDS -- and then runs the two coefficient phases of k_gat_alpha_groups with LDS logits.  One pass over the packed rows
// (4 N Dn bytes) replaces the pack pass's read of h, its write of the pieces and the a_node round trip.
struct ChainScaleArgs {      // what the hop launch that follows needs decided ahead of it (hop2.hip, chained hops); all NULL: nothing to do
    const float* PMin;       // [ncb][B] per-graph maxima of this hop's INPUT rows over each column block (left by the previous hop)
    const float* Tmax;       // NULL or [B]: largest |instruction term| of this hop per graph
    const float* bc;         // [4]: largest weight-row L1 norm | BN scale | BN shift | bias magnitude of this hop
    float* gscale;           // [B] out: power-of-two scale of the hop's OUTPUT rows, per graph
    float* a_inv_next;       // [128 G] out: its inverse per output slot
    int ncb, B;
};

template <int J>
__global__ __launch_bounds__(512) void k_gat_alpha_groups_packed(MpArgs a, const int32_t* __restrict__ group_ptr, const uint16_t* __restrict__ Apk,
                                                                 const float* __restrict__ a_inv, int KB, const uint16_t* __restrict__ vn_pk,
                                                                 const float* __restrict__ vn_inv, ChainScaleArgs cs) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x16v __attribute__((ext_vector_type(16)));
    constexpr int H = J / 2;
    extern __shared__ float raw_s[];                        // [2][128][J] node logits by k half | [slots of the group][H]
    float* an_s = raw_s;
    float* rs = raw_s + 2 * 128 * J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rtile = wave & 3, khalf = wave >> 2;           // eight waves: two per row tile, each half of the k blocks
    const int grp = blockIdx.x;
    const int ns = group_ptr[grp], cnt = group_ptr[grp + 1] - ns;
    const int e0 = a.rowptr[ns], ne = a.rowptr[ns + cnt] - e0;
    __shared__ float gs_s[128];
    // this thread's first edge slot: source row, edge id and the edge halves of the logits start their trips from HBM now, ahead of
    // the matrix-core logits (they depend on nothing computed here: two dependent loads off the critical path of phase 1)
    int pf_src = 0;
    float pf_ae[H];
#pragma unroll
    for (int h = 0; h < H; ++h) pf_ae[h] = 0.f;
    if (tid < ne) {
        pf_src = a.csr_src[e0 + tid] - ns;
        const float* ae = a.a_edge + (int64_t)a.csr_eid[e0 + tid] * a.a_edge_stride;
#pragma unroll
        for (int h = 0; h < H; ++h) pf_ae[h] = ae[h];
    }
    if (cs.gscale) {
        // Scale of the rows the hop is about to produce, one power of two per graph from an upper bound of their magnitudes (the hop
        // writes them as fp16 pieces straight from its epilogue, hop2.hip):
        //   |out| <= max|BN scale| (M (max L1 norm of a weight row + 1) + max|instruction term| + max|bias|) + max|BN shift|,
        // M = the largest input magnitude in the graph.  The bound overshoots by 2^5 .. 2^10; the two-piece split keeps 2^-22 relative
        // accuracy down to 2^-27 of the scaled maximum and degrades gracefully below.
        const int gf = a.node_graph[ns], ngl = a.node_graph[ns + cnt - 1] - gf + 1;
        if (tid < ngl) {
            const int g = gf + tid;
            float M = 0.f;
            for (int q = 0; q < cs.ncb; ++q) M = fmaxf(M, cs.PMin[(int64_t)q * cs.B + g]);
            const float tm = cs.Tmax ? cs.Tmax[g] : 0.f;
            const float bound = (cs.bc[1] * (M * (cs.bc[0] + 1.f) + tm + cs.bc[3]) + cs.bc[2]) * 1.001f;
            int ex = 0;
            if (bound > 0.f && bound <= 3.0e38f) { int e2; frexpf(bound, &e2); ex = max(-114, min(126, 14 - e2)); }
            const float sc = __uint_as_float((unsigned)(127 + ex) << 23);
            gs_s[tid] = sc;
            cs.gscale[g] = sc;
        }
        __syncthreads();
        if (tid < 128) cs.a_inv_next[grp * 128 + tid] = tid < cnt ? 1.0f / gs_s[a.node_graph[ns + tid] - gf] : 1.f;
    }
    {
        f32x16v lacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
        const uint16_t* ap = Apk + (int64_t)(grp * 4 + rtile) * KB * 1024 + lane * 8;
        const uint16_t* vp = vn_pk + lane * 8;
        const int kb0 = khalf ? (KB + 1) / 2 : 0, kb1 = khalf ? KB : (KB + 1) / 2;
#pragma unroll 8
        for (int kb = kb0; kb < kb1; ++kb) {
            const f16x8 p0 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ap + (int64_t)kb * 1024));
            const f16x8 p1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ap + (int64_t)kb * 1024 + 512));
            const f16x8 q0 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(vp + (int64_t)kb * 1024));
            const f16x8 q1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(vp + (int64_t)kb * 1024 + 512));
            lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1, p0, lacc, 0, 0, 0);      // transposed accumulators: lane (m, hh) holds logits
            lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, p1, lacc, 0, 0, 0);      // 8 q + 4 hh + 0..3 of row m in registers 4 q + 0..3
            lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, p0, lacc, 0, 0, 0);
        }
        const int m = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int q = 0; q < (J + 7) / 8; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 8 * q + 4 * hh + r;
                if (j < J) an_s[(khalf * 128 + rtile * 32 + m) * J + j] = lacc[4 * q + r];
            }
    }
    __syncthreads();
    for (int it = tid; it < 128 * J; it += 512) {           // the two k halves, then the rows' and vectors' exact power-of-two factors
        const int r = it / J, j = it - r * J;
        an_s[it] = (an_s[it] + an_s[128 * J + it]) * a_inv[grp * 128 + r] * vn_inv[j];
    }
    __syncthreads();
    if (tid < ne) {
#pragma unroll
        for (int h = 0; h < H; ++h) rs[tid * H + h] = an_s[pf_src * J + h] + pf_ae[h];
    }
    for (int s = tid + 512; s < ne; s += 512) {
        const int src = a.csr_src[e0 + s] - ns, eid = a.csr_eid[e0 + s];
        const float* ae = a.a_edge + (int64_t)eid * a.a_edge_stride;
        for (int h = 0; h < H; ++h) rs[s * H + h] = an_s[src * J + h] + ae[h];
    }
    __syncthreads();
    for (int it = tid; it < cnt * H; it += 512) {
        const int i = it / H, h = it - i * H, node = ns + i;
        const int lo = a.rowptr[node] - e0, hi = a.rowptr[node + 1] - e0;
        float ar = an_s[i * J + H + h];
        if (a.graph_term) ar += a.graph_term[(int64_t)a.node_graph[node] * a.t_ld + a.C + h];
        float m = -INFINITY;
        for (int s = lo; s < hi; ++s) {
            const float v = leaky(rs[s * H + h] + ar, a.slope);
            rs[s * H + h] = v;
            m = fmaxf(m, v);
        }
        float sum = 0.f;
        for (int s = lo; s < hi; ++s) {
            const float ex = expf(rs[s * H + h] - m);
            rs[s * H + h] = ex;
            sum += ex;
        }
        const float den = sum + 1e-16f;
        for (int s = lo; s < hi; ++s) {
            float al = rs[s * H + h] / den;
            if (a.alpha_out) a.alpha_out[(int64_t)a.csr_eid[e0 + s] * H + h] = al;
            if (a.alpha_mask) al *= a.alpha_mask[(int64_t)a.csr_eid[e0 + s] * H + h];
            a.alpha_csr[(int64_t)(e0 + s) * H + h] = al;
        }
    }
}

static int launch_alpha_packed(const MpArgs& a, int H, hipStream_t stream, const gvqa_graph* g, const void* Apk, int KB, const void* Vn_packed,
                               const ChainScaleArgs& cs) {
    GVQA_REQUIRE(g && g->num_row_groups > 0 && g->row_group_ptr && Apk && Vn_packed, GVQA_E_INVALID, "alpha_packed: null argument");
    const size_t lds = ((size_t)2 * 128 * 2 * H + (size_t)g->max_row_group_edges * H) * sizeof(float);
    GVQA_REQUIRE(lds <= 64 * 1024, GVQA_E_UNSUPPORTED, "alpha_packed: row group too large");
    const uint16_t* ap = static_cast<const uint16_t*>(Apk);
    const float* a_inv = reinterpret_cast<const float*>(static_cast<const char*>(Apk) + (size_t)g->num_row_groups * 4 * KB * 2048);
    const uint16_t* vp = static_cast<const uint16_t*>(Vn_packed);
    const float* v_inv = reinterpret_cast<const float*>(static_cast<const char*>(Vn_packed) + (size_t)KB * 2048);
    const dim3 grid((unsigned)g->num_row_groups), block(512);
#define GVQA_AP(J_) hipLaunchKernelGGL((k_gat_alpha_groups_packed<J_>), grid, block, lds, stream, a, g->row_group_ptr, ap, a_inv, KB, vp, v_inv, cs)
    switch (H) {
        case 1: GVQA_AP(2); break;
        case 2: GVQA_AP(4); break;
        case 4: GVQA_AP(8); break;
        case 8: GVQA_AP(16); break;
        default: return GVQA_E_UNSUPPORTED;
    }
#undef GVQA_AP
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

__global__ __launch_bounds__(256) void k_gat_aggregate_general(MpArgs a, int H) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= a.N) return;
    const int lo = a.rowptr[i], hi = a.rowptr[i + 1];
    const int g = a.node_graph[i];
    const float inv_h = 1.0f / H;
    for (int c = lane; c < a.C; c += 64) {
        float acc = 0.f;
        for (int s = lo; s < hi; ++s) {
            const float* row = a.xp + (int64_t)a.csr_src[s] * a.xp_ld + c;
            const float* al = a.alpha_csr + (int64_t)s * H;
            for (int h = 0; h < H; ++h) acc += al[h] * row[(int64_t)h * a.C];
        }
        float r = acc * inv_h;
        if (a.graph_scale) r *= a.graph_scale[(int64_t)g * a.gs_ld + c];
        if (a.graph_term && hi > lo) r += a.graph_term[(int64_t)g * a.t_ld + c];
        if (a.bias) r += a.bias[c];
        if (a.skip) r += a.skip[(int64_t)i * a.skip_ld + c];
        if (a.bn_w) r = fmaxf((r - a.bn_m[c]) * (1.0f / sqrtf(a.bn_v[c] + a.bn_eps)) * a.bn_w[c] + a.bn_b[c], 0.f);
        a.out[(int64_t)i * a.out_ld + c] = r;
    }
}

// ---- dispatch ---------------------------------------------------------------------------------
constexpr size_t LDS_MAX = 160 * 1024;

struct TilePlan {
    bool ok;
    int cw, e_cap, n_cap, nbuf, lpn_log;
    size_t lds_bytes;
};

// plan tunables of the stand-alone kernel: environment overrides exist in the measurement build only (scripts/bench_mp_plan.py)
static size_t env_size(const char* name, size_t dflt) {
#ifdef GVQA_PROBES
    const char* v = getenv(name);
    return (v && *v) ? (size_t)strtoull(v, nullptr, 10) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

static size_t tiled_lds_bytes(size_t e_cap, size_t n_cap, int C, int H, int cw, int nbuf, bool head_rows = false) {
    size_t off = e_cap * H * 4;
    off = align_up(off + e_cap * 4, 16);
    off = align_up(off + (n_cap + 1) * 4, 16);
    off = align_up(off + (n_cap + 1) * 4, 16);                                      // quad starts
    off += (size_t)(H + 1) * (size_t)mp_padded_quads((int)e_cap, (int)n_cap) * 16;  // padded coefficients (head-major) + source rows
    off += (size_t)5 * C * 4 + (head_rows ? (size_t)H * C * 4 : 0);
    return off + (size_t)nbuf * align_up(n_cap * (size_t)(cw / 4), MP_THREADS) * 16;
}

// Stage geometry.  HBM streaming is latency-bound per CU (Little: ~25 GB/s/CU x ~3 us loaded latency
// ~ 80 KB in flight), so the plan maximises bytes in flight per CU = blocks/CU x (nbuf-1) x stage
// bytes: per-block LDS target 160 KiB / 3, up to 4 stage buffers, channel range cw as wide as
// fits (row segments of >= 256 B preferred, >= 128 B required unless C is smaller), subject to the
// per-thread accumulator budget.  If nothing fits the 3-per-CU target, 2 then 1 per CU.
// Tunables for experiments: GVQA_MP_CW, GVQA_MP_NBUF, GVQA_MP_LDS (bytes per block).
static TilePlan plan_tiled(const gvqa_graph* g, int C, int H, bool head_rows = false) {
    TilePlan p{false, 0, 0, 0, 0, 0, 0};
    if (!g->finalized || !g->intra_graph || (C & 3) || !(H == 1 || H == 2 || H == 4 || H == 8)) return p;
    if (g->num_graphs <= 0 || g->num_graphs > 0x7fffffff) return p;
    const size_t e_cap = (size_t)(g->max_graph_edges > 0 ? g->max_graph_edges : 1);
    const size_t n_cap = (size_t)(g->max_graph_nodes > 0 ? g->max_graph_nodes : 1);
    const size_t forced_cw = env_size("GVQA_MP_CW", 0);
    const size_t forced_nbuf = env_size("GVQA_MP_NBUF", 2);   // measured: deeper prefetch does not pay
    const size_t forced_lds = env_size("GVQA_MP_LDS", 0);
    // 3 graphs per CU, else 2, else 1 (measured at config 3: 4 per CU with narrower channel ranges is
    // slower -- 228 us vs 166 us -- the per-stage overhead outweighs the removed block-wave tail)
    const size_t targets[3] = {LDS_MAX / 3, LDS_MAX / 2, LDS_MAX};
    for (int ti = 0; ti < 3 && !p.ok; ++ti) {
        const size_t target = forced_lds ? (forced_lds < LDS_MAX ? forced_lds : LDS_MAX) : targets[ti];
        size_t best_score = 0;
        for (int nbuf = 2; nbuf <= 4; ++nbuf) {
            if (forced_nbuf && (size_t)nbuf != forced_nbuf) continue;
            for (int nch = 1; nch <= C / 4; ++nch) {
                int cw = (int)align_up((size_t)cdiv(C, nch), 4);
                if (forced_cw) cw = (int)forced_cw;
                int lpn_log = 0;
                while ((1 << lpn_log) < cw / 4) ++lpn_log;
                if (cw > C || (1 << lpn_log) > MP_THREADS ||
                    n_cap > (size_t)MP_ITEMS * (MP_THREADS >> lpn_log)) { if (forced_cw) break; continue; }
                const size_t bytes = tiled_lds_bytes(e_cap, n_cap, C, H, cw, nbuf, head_rows);
                if (bytes > target) { if (forced_cw) break; continue; }
                if (cw < 32 && cw < C && ti < 2 && !forced_cw) break;        // segments < 128 B: next target
                size_t score = (size_t)(nbuf - 1) * n_cap * cw * 4;           // bytes in flight per block
                if (cw < 64 && cw < C) score /= 2;                            // < 256 B segments: penalise
                if (score > best_score) {
                    best_score = score;
                    p.ok = true; p.cw = cw; p.e_cap = (int)e_cap; p.n_cap = (int)n_cap; p.nbuf = nbuf; p.lpn_log = lpn_log;
                    p.lds_bytes = bytes;
                }
                break;      // widest cw for this nbuf found
            }
        }
        if (forced_lds) break;
    }
    return p;
}

// Blocks per graph.  Splitting a graph's channel ranges over n blocks repeats its alpha prologue n times, so it only
// pays while the batch cannot fill the chip by itself (measured at d = 512, 32-node graphs: B = 64: 39.9 us with one
// block per graph, 18.9 us with four; B = 256: 41.5 -> 32.4 us with two; B >= 1000: one block per graph is best, the
// block-wave tail costs less than the repeated prologues).  Target two blocks per CU.  GVQA_MP_PARTS overrides.
static int plan_parts(int64_t B, int C, const TilePlan& p) {
    const int64_t nch = cdiv(C, p.cw);
    const size_t forced = (size_t)std::max(get_option(GVQA_OPT_MP_PARTS), 0);
    if (forced) return (int)std::min<int64_t>((int64_t)forced, nch);
    return (int)std::max<int64_t>(1, std::min<int64_t>(nch, cdiv(2 * 256, std::max<int64_t>(B, 1))));
}

template <int H, int ITEMS, bool HR = false>
static int launch_tiled_i(const MpArgs& a, const TilePlan& p, int64_t B, hipStream_t stream) {
    // the 160 KiB dynamic-LDS opt-in is a per-device function attribute: set once per (instantiation, device)
    static std::mutex mu;
    static bool attr_set[64] = {};
    int dev = 0;
    GVQA_HIP_CHECK(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            GVQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gat_mp_tiled<H, ITEMS, HR>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL((k_gat_mp_tiled<H, ITEMS, HR>), dim3((unsigned)B, (unsigned)a.nparts), dim3(MP_THREADS), p.lds_bytes, stream, a);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

template <int H>
static int launch_tiled(const MpArgs& a, const TilePlan& p, int64_t B, hipStream_t stream) {
    // fewer accumulators (registers) when the largest graph needs only two node passes per thread
    if (a.head_rows) {
        if ((size_t)p.n_cap <= (size_t)2 * (MP_THREADS >> p.lpn_log)) return launch_tiled_i<H, 2, true>(a, p, B, stream);
        return launch_tiled_i<H, MP_ITEMS, true>(a, p, B, stream);
    }
    if ((size_t)p.n_cap <= (size_t)2 * (MP_THREADS >> p.lpn_log)) return launch_tiled_i<H, 2>(a, p, B, stream);
    return launch_tiled_i<H, MP_ITEMS>(a, p, B, stream);
}

static int launch_gat_mp(const gvqa_graph* g, const gvqa_gat_mp_desc* d, void* ws, size_t ws_bytes, hipStream_t stream) {
    GVQA_REQUIRE(g && d, GVQA_E_INVALID, "gat_mp: null argument");
    const int C = d->C, H = d->H;
    GVQA_REQUIRE(d->xp && d->a_edge && d->out, GVQA_E_INVALID, "gat_mp: null tensor");
    GVQA_REQUIRE(C > 0 && H > 0, GVQA_E_INVALID, "gat_mp: bad dims");
    const bool bn = d->bn_weight || d->bn_bias || d->bn_mean || d->bn_var;
    GVQA_REQUIRE(!bn || (d->bn_weight && d->bn_bias && d->bn_mean && d->bn_var), GVQA_E_INVALID,
                 "gat_mp: BatchNorm needs weight, bias, running_mean and running_var");
    if (g->num_nodes == 0) return GVQA_OK;
    MpArgs a;
    memset(&a, 0, sizeof(a));
    a.rowptr = g->rowptr; a.csr_src = g->csr_src; a.csr_eid = g->csr_eid;
    a.node_graph = g->node_graph; a.graph_ptr = g->graph_ptr;
    a.xp = d->xp; a.xp_ld = d->xp_ld ? d->xp_ld : (int64_t)H * C;
    a.a_node = d->a_node; a.a_edge = d->a_edge; a.a_edge_stride = d->a_edge_stride ? d->a_edge_stride : H;
    GVQA_REQUIRE(!d->graph_term || (d->graph_term_ld >= C + H && d->graph_term_ld % 4 == 0), GVQA_E_INVALID,
                 "gat_mp: graph_term_ld must be >= C+H and a multiple of 4");
    a.graph_term = d->graph_term; a.t_ld = d->graph_term_ld;
    a.graph_scale = d->graph_scale; a.gs_ld = d->graph_scale_ld ? d->graph_scale_ld : C;
    a.skip = d->skip; a.skip_ld = d->skip_ld ? d->skip_ld : C;
    a.bias = d->bias;
    a.bn_w = d->bn_weight; a.bn_b = d->bn_bias; a.bn_m = d->bn_mean; a.bn_v = d->bn_var;
    a.out = d->out; a.out_ld = d->out_ld ? d->out_ld : C; a.alpha_out = d->alpha_out; a.alpha_mask = d->alpha_mask; a.alpha_csr = nullptr;
    a.head_rows = d->head_rows; a.hr_ld = d->head_rows_ld ? d->head_rows_ld : (int64_t)H * C; a.head_weight_out = d->head_rows ? d->head_weight_out : nullptr;
    GVQA_REQUIRE(!d->head_rows || (a.hr_ld >= (int64_t)H * C && !d->graph_term && !d->graph_scale), GVQA_E_INVALID,
                 "gat_mp: head_rows needs head_rows_ld >= H*C and excludes graph_term / graph_scale");
    a.N = (int)g->num_nodes; a.C = C; a.cw = 0; a.e_cap = 0; a.n_cap = 0; a.nbuf = 2; a.nparts = 1; a.lpn_log = 0;
#ifdef GVQA_PROBES
    { const char* dv = getenv("GVQA_MP_DEBUG"); a.debug = dv ? atoi(dv) : 0; }
#endif
    a.slope = d->negative_slope; a.bn_eps = d->bn_eps;
    const int force = d->force;
    const float* graph_term = d->graph_term;
    const bool ld_vec_ok = (a.xp_ld % 4 == 0) && (a.skip_ld % 4 == 0) && (a.out_ld % 4 == 0) &&
                           ((reinterpret_cast<uintptr_t>(a.xp) | reinterpret_cast<uintptr_t>(a.out) |
                             reinterpret_cast<uintptr_t>(a.skip)) & 15) == 0;

    StageTimer timer(GVQA_STAGE_MP, stream);
    TilePlan plan = plan_tiled(g, C, H, d->head_rows != nullptr);
    if (!ld_vec_ok) plan.ok = false;
    GVQA_REQUIRE(!d->head_rows || (plan.ok && force != 2), GVQA_E_UNSUPPORTED, "gat_mp: head_rows needs the LDS-tiled kernel (gvqa_graph_head_rows_add is the general form)");
    GVQA_REQUIRE(force != 1 || plan.ok, GVQA_E_UNSUPPORTED,
                 "gat_mp: tiled kernel not applicable (needs finalized intra-graph batch, C %% 4 == 0, "
                 "H in {1,2,4,8}, largest graph fitting 160 KiB of LDS)");
    if (plan.ok && force != 2) {
        a.cw = plan.cw; a.e_cap = plan.e_cap; a.n_cap = plan.n_cap; a.nbuf = plan.nbuf; a.lpn_log = plan.lpn_log;
        a.nparts = plan_parts(g->num_graphs, C, plan);
        switch (H) {
            case 1: return launch_tiled<1>(a, plan, g->num_graphs, stream);
            case 2: return launch_tiled<2>(a, plan, g->num_graphs, stream);
            case 4: return launch_tiled<4>(a, plan, g->num_graphs, stream);
            default: return launch_tiled<8>(a, plan, g->num_graphs, stream);
        }
    }
    GVQA_REQUIRE(graph_term == nullptr || (g->finalized && g->intra_graph), GVQA_E_UNSUPPORTED,
                 "gat_mp: per-graph terms need an intra-graph batch");
    const size_t need = (size_t)g->num_edges * H * sizeof(float);
    GVQA_REQUIRE(need == 0 || (ws && ws_bytes >= need), GVQA_E_WORKSPACE,
                 "gat_mp: general kernel needs %zu workspace bytes", need);
    a.alpha_csr = static_cast<float*>(ws);
    hipLaunchKernelGGL(k_gat_alpha_general, dim3((unsigned)cdiv((int64_t)a.N * H, 256)), dim3(256), 0, stream, a, H);
    hipLaunchKernelGGL(k_gat_aggregate_general, dim3((unsigned)cdiv(a.N, 4)), dim3(256), 0, stream, a, H);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// Train-mode BatchNorm1d (batch statistics over all N rows, biased variance) + ReLU after the skip, in place:
// the column-reduction kernels of bn_train.hip (gvqa_bn_relu_train_forward), y == x.
static int bn_train_forward(int64_t N, int C, float* h, const gvqa_gat_conv_params* p, float eps, float* stats /* [2,C] */,
                            float* partial, size_t partial_bytes, hipStream_t stream) {
    StageTimer t(GVQA_STAGE_OTHER, stream);
    return gvqa_bn_relu_train_forward(N, C, h, p->bn_weight, p->bn_bias, eps, h, stats, stats + C, partial, partial_bytes, stream);
}

int launch_gat_mp_public(const gvqa_graph* g, const gvqa_gat_mp_desc* d, void* ws, size_t ws_bytes, hipStream_t stream) {
    return launch_gat_mp(g, d, ws, ws_bytes, stream);
}

// ==============================================================================================
// Drivers
// ==============================================================================================
// Projection arithmetic of the hop GEMM xp = h . W_h^T (gat_skip.py:133), GVQA_OPT_PROJECTION:
//   split2h           two scaled fp16 pieces per value, three fp16-MFMA products (split3.hip)
//   split3            three exact bf16 pieces per value, six bf16-MFMA products (split3.hip)
//   f32               f32-input MFMA (k_linear_f32*; rocBLAS only when GVQA_OPT_VENDOR_GEMM asks for it)
// Products too small to fill the chip stay on the f32 kernels (the pack passes would not pay).
// Returns the pieces per value of the split projection (2 / 3), or 0 for the f32 kernels.
// options a call may override in its dims struct (0 = the process-wide value, else value + 1)
static int opt_projection(const gvqa_gat_dims* d) { return d && d->projection > 0 ? d->projection - 1 : get_option(GVQA_OPT_PROJECTION); }
static int opt_hop_fusion(const gvqa_gat_dims* d) { return d && d->hop_fusion > 0 ? d->hop_fusion - 1 : get_option(GVQA_OPT_HOP_FUSION); }

static int proj_pieces(const gvqa_gat_dims* d, int64_t M, int64_t N, int64_t K) {
    const int mode = opt_projection(d);
    if (mode == GVQA_PROJECTION_F32 || N % 4 != 0 ||
        2.0 * (double)M * (double)N * (double)K < 1e6 * (double)get_option(GVQA_OPT_SPLIT3_MIN_MFLOP))
        return 0;
    return mode == GVQA_PROJECTION_SPLIT2H ? 2 : 3;
}
// weight-cache layout id: -1 no packed projection weights; bit 0: head-interleaved rows (fused hop) instead of plain row order;
// bit 1: two fp16 pieces instead of three bf16 pieces
// bit 2: half-interleaved rows (hop2.hip) instead of head-interleaved ones (identical for H = 4)
static int weight_layout_id(int pieces, bool heads, bool hop2 = false) {
    return pieces == 0 ? -1 : (heads ? 1 : 0) | (pieces == 2 ? 2 : 0) | (heads && hop2 && pieces == 2 ? 4 : 0);
}
static int layout_pieces(int layout) { return layout < 0 ? 0 : (layout & 2) ? 2 : 3; }
// bit 3: K'-concatenated two-piece weights with per-column scales (the aggregate-first hop, hopagg.hip); excludes bits 0 and 2
constexpr int LAYOUT_AGGFIRST = 2 | 8;

// Fused hop (projection + aggregation in one kernel, split3.hip): needs the split3 projection, a row-group plan (every
// graph <= 128 nodes, intra-graph batch), H dividing 256 and the largest row group's edges within the kernel's LDS budget.
static bool hop_fusion_applies(const gvqa_graph* g, const gvqa_gat_dims* d) {
    const int H = d->heads, C = d->out_channels;
    return opt_hop_fusion(d) != 0 && proj_pieces(d, g->num_nodes, (int64_t)H * C, d->node_dim) != 0 &&
           g->num_row_groups > 0 && g->row_group_ptr && (H == 1 || H == 2 || H == 4 || H == 8) && C % 4 == 0 &&
           (size_t)g->max_row_group_edges <= hop_fused_lds_edge_capacity(H) &&
           cdiv(g->num_row_groups, 2) <= 65535;                   // (grid.y of the 8-wave kernel: beyond it the unfused kernels run)
}

// The hop "aggregate first" (hopagg.hip, GVQA_OPT_HOP_FUSION = 4; taken by the default rule 3 when the batch fills the chip):
// H = 4, C == Dn <= 512, the two-piece projection, a row-group plan with <= 1024 edges per group.  Rows travel chunk-major
// between hops; per-hop fp32 outputs and the attention weights are served, batch-statistics BatchNorm is not.
constexpr bool kAggFirstByDefault = true;            // (since the epilogue's loads run a batch ahead: 386-400 vs 389-428 us per hop for the chained 8-wave kernel, same boxes)
// ... with a row group's output columns split over four workgroups (k_hopagg4<4, 2, 1, 2, ..., CP = 4>; GVQA_OPT_HOP_FUSION = 6: built for
// strong-scaling shards, where one workgroup per row group leaves most CUs idle).  Per-hop launches; attention weights and per-hop rows served.
constexpr int kHopaggParts = 4;
// Packed row groups (gvqa_graph::pk_*, graph.hip): when the handle carries them, the aggregate-first hops run on the packed numbering --
// fewer, fuller groups = fewer workgroups of the one-per-CU launch
static bool agg_packed(const gvqa_graph* g) { return g->pk_num_row_groups > 0 && g->pk_row_group_ptr && get_option(GVQA_OPT_PACKED_GROUPS) != 0; }
static int agg_groups(const gvqa_graph* g) { return agg_packed(g) ? g->pk_num_row_groups : g->num_row_groups; }
static int agg_max_group_edges(const gvqa_graph* g) { return agg_packed(g) ? g->pk_max_row_group_edges : g->max_row_group_edges; }
static bool hopagg_parts_shape(const gvqa_graph* g, const gvqa_gat_dims* d) {
    const int H = d->heads, C = d->out_channels;
    return H == 4 && C == d->node_dim && C > 384 && C <= 512 && g->num_row_groups > 0 && g->row_group_ptr && g->intra_graph &&
           proj_pieces(d, g->num_nodes, (int64_t)H * C, d->node_dim) == 2 && hopagg_supported(H, C, d->node_dim, agg_max_group_edges(g));
}
static bool hopagg_parts_applies(const gvqa_graph* g, const gvqa_gat_dims* d) {
    const int mode = opt_hop_fusion(d);
    // (explicit only.  Measured on 256- / 128- / 64-graph shards of config 3, round 5: 89 / 82 / 81 us per hop against 54 / 35 / 31 for the
    //  8-wave kernel + its two small launches (0.54 vs 0.43 ms per step at 256 graphs) -- a 128 x 128 part is 6 MFMAs in two dependent
    //  chains per wave and step beside the full producer, and 64 us of the hop are outside the loop; profiles/r05_hop_coeffs_ab.txt)
    return mode == 6 && hopagg_parts_shape(g, d);
}
static bool hopagg_applies(const gvqa_graph* g, const gvqa_gat_dims* d) {
    const int mode = opt_hop_fusion(d);
    if (hopagg_parts_applies(g, d)) return true;
    if (mode != 4 && mode != 5 && mode != 3) return false;
    const int H = d->heads, C = d->out_channels;
    if (!(proj_pieces(d, g->num_nodes, (int64_t)H * C, d->node_dim) == 2 && g->num_row_groups > 0 && g->row_group_ptr && g->intra_graph &&
          hopagg_supported(H, C, d->node_dim, agg_max_group_edges(g))))
        return false;
    if (mode == 3 && !kAggFirstByDefault) return false;
    if (mode == 3) {
        // one workgroup per row group, one resident per CU: the default rule takes it when the row groups fill the CUs' last round
        // well enough (config 3: 512 groups on 256 CUs); smaller / ragged batches keep the item-granular kernels.  Thresholds from a
        // sweep of batch sizes at config-3 widths (profiles/r04d_sweep_batch_sizes.jsonl, forward wall ms, one launch vs chained
        // 8-wave): 425 groups (0.83 of two rounds) 2.13 vs 2.29; 550 (0.72 of three) 3.18 vs 2.90; 725 (0.94 of three) 3.32 vs 3.82 --
        // a workgroup's hops run faster when the last round is thin (fewer CUs draw power), so two rounds pay from ~0.72 on, three
        // from ~0.80; one round and four or more keep the earlier 0.85.
        const int64_t cus = device_cu_count();         // (per device: the thresholds below were tuned on the 256-CU part)
        const int64_t G = agg_groups(g), rounds = cdiv(G, cus);
        const int64_t fill_pct = rounds == 2 ? 76 : rounds == 3 ? 82 : 85;
        if (G * 100 < rounds * cus * fill_pct) return false;
        // (round 5 excluded C <= 320 here: "d = 300, 149 vs 110 us per hop".  That comparison was config 2's 262 in-order row groups = TWO
        //  rounds on 256 CUs; inside a workgroup a d = 300 hop is 88 us against 113 + 17 + 4 for the 8-wave kernel and its two small launches
        //  (profiles/r06a_cfg2_seq_stamps.json), and with the packed row groups (246 groups: one round) the forward is 0.56 vs 0.72 ms
        //  (profiles/r06b_cfg2_packed_ab.jsonl).  Narrower rows are unmeasured: they keep the 8-wave kernels)
        if (C <= 256) return false;
    }
    return true;
}

// ... and its K hops as ONE launch (GVQA_OPT_HOP_FUSION = 5; k_hopagg4<..., SEQ>): plain outputs only -- the attention weights and
// per-hop fp32 rows are served by the per-hop launches
// Default rule (mode 3): wherever it takes the aggregate-first kernel, a plain-output forward of K >= 2 hops is the one launch (since the
// overflow edges sit in registers the workgroups' hops are even, and the one launch is the faster form: 2.29 vs 2.35 ms per step, same box).
static bool hopagg_seq_applies(const gvqa_graph* g, const gvqa_gat_dims* d) {
    const int mode = opt_hop_fusion(d);
    return (mode == 5 || mode == 3) && d->num_hops >= 2 && d->num_hops <= HA_MAXHOPS && hopagg_applies(g, d) && !hopagg_parts_applies(g, d);
}

// Chained hops on the 8-WAVE kernel (launch_hop_fused_split with a chain descriptor): a hop writes the next hop's packed operand
// from its epilogue, as the persistent kernel's chained form does, on the kernel whose launches are the shorter ones (round 4, same
// box: 391 vs 415 us per hop at config 3, forward 2.40 vs 2.51 ms; config 2's d = 300 batch 0.72 -> 0.69 ms; a 256-graph shard
// loses its 12 us pack pass per hop).  H = 4 -- the head-interleaved weight rows then ARE the half-interleaved ones of the persistent
// kernel: one cache layout, id 7, serves both.  Shape conditions (the batch-level ones are hop_fusion_applies'):
// In-kernel attention coefficients for the chained 8-wave hops (k_linear_split3<..., CHN = 2>, GVQA_OPT_HOP_COEFFS): both row groups'
// CSR slices resident in LDS at once
static bool chain8_in_kernel_coeffs(const gvqa_graph* g, const gvqa_gat_dims* d) {
    // GVQA_OPT_HOP_COEFFS: 1 always, 0 never, 2 (default, round 6) by size -- below 128 row groups a hop's launches are latency-bound and
    // the in-kernel phase replaces two of them per hop (256-graph shard, same box: 0.413 -> 0.394 ms, 16 launches -> 7); above that the
    // coefficient kernels win (config 2 on the 8-wave kernel, round 5: 0.705 vs 0.725)
    const int opt = get_option(GVQA_OPT_HOP_COEFFS);
    const bool want = opt == 1 || (opt == 2 && g->num_row_groups < 128);
    return want && d->heads == 4 && (size_t)g->max_row_group_edges <= hop_fused_ic_lds_edge_capacity(4);
}
static bool chain8_shape_ok(const gvqa_graph* g, const gvqa_gat_dims* d) {
    // (below ~128 row groups every launch of a hop is latency-bound and the chained coefficient kernel -- it reads the packed rows
    //  through the matrix cores -- costs more than the pack pass it replaces: 0.428 vs 0.413 ms on a 256-graph shard; GVQA_OPT_HOP_FUSION
    //  = 1 / 2 ask for a kernel explicitly and chain regardless.  With the coefficients computed inside the hop kernel there is no
    //  coefficient kernel to pay for: chained at every size)
    if (opt_hop_fusion(d) == 3 && g->num_row_groups < 128 && !chain8_in_kernel_coeffs(g, d)) return false;
    return d->heads == 4 && d->node_dim == d->out_channels &&
           proj_pieces(d, g->num_nodes, (int64_t)d->heads * d->out_channels, d->node_dim) == 2 &&
           (size_t)g->max_row_group_edges <= hop_fused_chain_lds_edge_capacity(d->heads) &&
           split_pack_groups_logits_supported(2, 2 * d->heads, d->node_dim);
}

// The hop as the persistent two-workgroups-per-CU kernel of hop2.hip (GVQA_OPT_HOP_FUSION = 2): two-piece operands and the
// largest row group's CSR slice within that kernel's 16 KiB region; otherwise the 8-wave fused kernel runs.
static bool hop2_applies(const gvqa_graph* g, const gvqa_gat_dims* d) {
    const int mode = opt_hop_fusion(d);
    if (mode != 2 && mode != 3) return false;
    if (mode == 3) {
        // by shape: the persistent kernel overlaps one item's epilogue with its CU partner's matrix-core loop, which pays from about
        // six (row group, column block) items per workgroup slot on (config 3: 8); below that the 8-wave kernel's single large
        // tile per CU is faster (measured: 256 .. 1024-graph shards and the d = 300 batch of config 2, profiles/r03_*)
        const int64_t slots = (int64_t)2 * device_cu_count();
        if ((int64_t)g->num_row_groups * cdiv(d->out_channels, 256 / d->heads) < 6 * slots) return false;
        if (chain8_shape_ok(g, d)) return false;          // the chained 8-wave kernel is the faster of the two wherever it applies
    }
    return hop_fusion_applies(g, d) &&
           proj_pieces(d, g->num_nodes, (int64_t)d->heads * d->out_channels, d->node_dim) == 2 &&
           (size_t)g->max_row_group_edges <= hop2_lds_edge_capacity(d->heads, false);
}
// ... with the hops chained: a hop's output leaves as the next hop's packed operand, partial logits and per-graph maxima, and
// the pack pass between hops disappears (eval forward without per-hop fp32 outputs; H >= 4)
static bool hop2_chain_capable(const gvqa_graph* g, const gvqa_gat_dims* d) {
#ifdef GVQA_PROBES
    static const bool off = []() { const char* v = getenv("GVQA_HOP2_CHAIN"); return v && v[0] == '0'; }();     // (A/B switch of the measurement build)
    if (off) return false;
#endif
    return hop2_applies(g, d) && d->node_dim == d->out_channels &&
           (size_t)g->max_row_group_edges <= hop2_lds_edge_capacity(d->heads, true);
}

static bool fused_chain8_capable(const gvqa_graph* g, const gvqa_gat_dims* d) {
    return hop_fusion_applies(g, d) && !hop2_applies(g, d) && chain8_shape_ok(g, d);
}

// ---- weight cache: everything a forward derives from the PARAMETERS alone (folded attention vectors Vn / Ve, per-graph
// term weights Gw, split3-packed projection weights of every hop) can be prepared once and reused while the weights do
// not change.  layout: see weight_layout_id.
// The per-graph instruction terms [K] x ([B, Di] x [Di, C + H]) as ONE batched two-piece product (weights packed in the cache,
// the instruction vectors packed per call): dims it takes (the f32-input MFMA kernel serves the rest)
static bool graph_term_split_dims(const gvqa_gat_dims* d) {
    return d->ins_dim > 0 && d->ins_dim % 4 == 0 && (d->out_channels + d->heads) % 4 == 0;
}
struct WeightCacheLayout {
    size_t Vn, Ve, Gw, w6, w6_hop, vn2h, vn2h_hop, epc, epc_hop, bc, gw2h, gw2h_hop, total;   // gw2h: two-piece images of every hop's Gw (the per-graph instruction terms as a batched split product)   // bc: bound constants of the chained hops   // epc: per-channel epilogue constants of every hop (hop2.hip)     // vn2h: two-piece images of every hop's Vn (fused hop on split2h: the
};                                                             // pack pass computes the attention logits on the matrix cores)
static WeightCacheLayout weight_cache_layout(const gvqa_gat_dims* d, int layout) {
    WeightCacheLayout W;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += align_up(bytes, 256); return r; };
    const size_t K = d->num_hops, H = d->heads, C = d->out_channels;
    W.Vn = take(K * 2 * H * d->node_dim * sizeof(float));
    W.Ve = take(K * H * d->edge_dim * sizeof(float));
    W.Gw = take(K * (C + H) * (size_t)d->ins_dim * sizeof(float));
    const int np = layout_pieces(layout);
    const bool aggw = layout >= 0 && (layout & 8);
    W.w6_hop = layout < 0 ? 0 : aggw ? align_up(hopagg_packed_w_bytes((int)C, d->node_dim, (int)H), 256)
                          : (layout & 1) ? split_packed_rows_bytes(np, cdiv((int64_t)C, 256 / (int64_t)H) * 8, d->node_dim)
                                         : split_packed_bytes(np, (int64_t)(H * C), d->node_dim);
    W.w6 = take(K * W.w6_hop);
    W.vn2h_hop = (layout >= 0 && (layout & 3) == 3 && !aggw) ? align_up(split_packed_bytes(2, 2 * (int64_t)H, d->node_dim), 256) : 0;
    W.vn2h = take(K * W.vn2h_hop);
    W.epc_hop = (layout >= 0 && (layout & (4 | 8))) ? align_up(3 * (size_t)hop2_consts_ld((int)H, (int)C) * sizeof(float), 256) : 0;
    W.epc = take(K * W.epc_hop);
    const bool chainw = layout >= 0 && (layout & 4) && d->node_dim == d->out_channels;
    W.bc = take(chainw ? K * 4 * sizeof(float) : 0);
    W.gw2h_hop = (layout_pieces(layout) == 2 && graph_term_split_dims(d)) ? align_up(split_packed_bytes(2, (int64_t)(C + H), d->ins_dim), 256) : 0;
    W.gw2h = take(K * W.gw2h_hop);
    W.total = off;
    return W;
}

struct SeqLayout {
    size_t Vn, Ve, Gw, T, a_edge, a_node, xp, h0, h1, alpha_csr, bn_partial, bn_stats, a6, w6, a6b, PM, Tmax, gscale, ipk, ipk_hop, x4a, x4b, gma, gmb, lp0, lp1, total;   // a6b ..: chained hops; x4a ..: aggregate-first hops
};

static SeqLayout seq_layout(int64_t N, int64_t E, int64_t B, const gvqa_gat_dims* d, const gvqa_graph* g = nullptr) {
    SeqLayout L;
    size_t off = 0;
    auto take = [&](size_t count) {
        size_t r = off;
        off += align_up(count * sizeof(float), 256);
        return r;
    };
    const size_t K = d->num_hops, H = d->heads, C = d->out_channels;
    const bool fused = g && hop_fusion_applies(g, d);
    const int np = proj_pieces(d, N, (int64_t)(H * C), d->node_dim);
    const bool chain8 = fused && fused_chain8_capable(g, d);
    const int w_layout = weight_layout_id(np, fused, fused && (hop2_applies(g, d) || chain8));
    const bool aggf = g && hopagg_applies(g, d);
    // Vn | Ve | Gw | packed projection weights (aggregate-first batches: the larger of the two forms -- a batch-statistics forward of
    // the same batch takes the other hop kernels)
    L.Vn = take(std::max(weight_cache_layout(d, w_layout).total, aggf ? weight_cache_layout(d, LAYOUT_AGGFIRST).total : (size_t)0) / sizeof(float));
    L.Ve = L.Gw = L.Vn;
    L.T = take(K * B * align_up(C + H, 4));
    L.a_edge = take((size_t)E * K * H);
    const bool agg_parts = g && hopagg_parts_applies(g, d);
    L.a_node = take((size_t)N * 2 * H * (agg_parts ? 2 * kHopaggParts : 1));      // (column parts: two buffers of one set per part)
    L.xp = take(fused ? 0 : (size_t)N * H * C);                // the fused hop never materialises xp
    L.h0 = take((size_t)N * C);
    L.h1 = take((size_t)N * C);
    L.alpha_csr = take((size_t)E * H);
    L.bn_partial = take(gvqa_bn_train_workspace_bytes(N > 0 ? N : 1, C) / sizeof(float) + 1);
    L.bn_stats = take(2 * C);
    if (fused) L.a6 = take(split_packed_rows_bytes(np, (int64_t)g->num_row_groups * 4, d->node_dim) / sizeof(float));   // row-group slots
    else if (np) L.a6 = take(split_packed_bytes(np, N, d->node_dim) / sizeof(float));
    else L.a6 = off;
    L.w6 = L.Vn;
    const bool chain = fused && (hop2_chain_capable(g, d) || chain8);
    const size_t ncb = chain ? (size_t)cdiv((int64_t)C, 256 / (int64_t)H) : 0;
    L.a6b = take(chain ? split_packed_rows_bytes(2, (int64_t)g->num_row_groups * 4, d->node_dim) / sizeof(float) : 0);
    L.PM = take(2 * ncb * (size_t)B);
    {   // aggregate-first hops: the rows chunk-major (two buffers: a hop reads one, writes the other), per-graph maxima likewise
        const size_t x4 = aggf ? (size_t)g->num_row_groups * (size_t)cdiv((int64_t)d->node_dim, 4) * 128 * 4 : 0;
        L.x4a = take(x4); L.x4b = take(x4);
        L.gma = take(aggf ? (size_t)B * (agg_parts ? kHopaggParts : 1) : 0); L.gmb = take(aggf ? (size_t)B * (agg_parts ? kHopaggParts : 1) : 0);
    }
    // in-kernel coefficients: the next hop's partial node logits by column block, two buffers (a hop reads one, writes the other)
    const bool ic_lp = chain && chain8 && chain8_in_kernel_coeffs(g, d);
    L.lp0 = take(ic_lp ? ncb * (size_t)N * 2 * H : 0);
    L.lp1 = take(ic_lp ? ncb * (size_t)N * 2 * H : 0);
    L.Tmax = take(chain ? K * (size_t)B : 0);
    L.gscale = take(chain ? (size_t)B : 0);
    // packed instruction vectors of the K hops (two-piece graph-term product): one image of K B rows, a hop = B / 32 whole
    // row tiles of it (other B: the f32-input kernel; K separate images measured slower than it at config 2, 55 vs 30 us)
    L.ipk_hop = (np == 2 && graph_term_split_dims(d) && B > 0 && B % 32 == 0) ? (size_t)(B / 32) * (size_t)cdiv((int64_t)d->ins_dim, 16) * 2048 : 0;
    L.ipk = take(L.ipk_hop ? split_packed_bytes(2, (int64_t)K * (int64_t)B, d->ins_dim) / sizeof(float) + 1 : 0);
    L.total = off;
    return L;
}

// Side stream for work that is independent of the projection GEMM of the same hop (HBM-bound
// logit / fold kernels under the MFMA-bound projection).  One non-blocking stream and a pair of
// events per host thread; fork = side waits for everything already enqueued on the caller's
// stream, join = caller's stream waits for the side stream.  OFF by default (GVQA_OVERLAP=1 turns it on).  Round 1 forked and
// joined around every hop's projection: the projection's blocks fill every CU, the side kernels waited for it anyway (+1.2 %).
// Round 4 re-cut it to what a fused forward does BEFORE its first hop -- the all-hops edge-logit pass, the per-graph instruction
// terms and the chained hops' term maxima beside hop 0's pack pass, ONE fork and ONE join per forward -- and measured it again,
// alternating on one box: 2.76 / 2.81 ms with, 2.63 / 2.63 ms without.  The kernels that would overlap are HBM-bound alike (they
// share the bandwidth) and the two cross-stream dependencies cost this runtime more than the small GEMM hides.  It stays off.
// Round 6: mode 2 -- ONLY the latency-bound per-graph instruction terms (a 50 us chain of two small launches) on the side stream, the HBM-bound
// edge-logit pass stays in front of the layout pass on the caller's stream.  On a forward with a PREBUILT graph handle it gained 0.9 % at config 3
// (2.122 / 2.136 -> 2.107 / 2.110 ms same box, scripts/bench_hopagg.py; mode 1: 2.133 / 2.129), nothing at config 2, and lost 3 % on a 256-graph
// shard -- and inside bench.py's step, where the CSR build (pinned upload + launch) precedes every forward, it LOST 5-10 % (2.12 / 2.25 -> 2.36 /
// 2.32 ms, alternating on one box; the side stream's two launches then run 2.5x longer and the join stalls the hop launch):
// `profiles/r06_overlap_modes_ab.jsonl`.  Off by default, as after rounds 1 and 4; GVQA_OVERLAP = 1 / 2 asks for a mode.
static int side_stream_mode(int64_t edges = 0) {
    static const int mode = []() { const char* v = getenv("GVQA_OVERLAP"); return v ? atoi(v) : 0; }();
    (void)edges;
    return mode;
}
static SideStream* side_stream(int64_t edges) { return side_stream_mode(edges) > 0 ? side_stream_get() : nullptr; }

static gvqa_gat_mp_desc mp_desc_from(const gvqa_gat_dims* d, const gvqa_gat_conv_params* p) {
    gvqa_gat_mp_desc m;
    memset(&m, 0, sizeof(m));
    m.C = d->out_channels; m.H = d->heads; m.negative_slope = d->negative_slope; m.bn_eps = d->bn_eps;
    m.bias = p->bias; m.bn_weight = p->bn_weight; m.bn_bias = p->bn_bias; m.bn_mean = p->bn_mean; m.bn_var = p->bn_var;
    return m;
}

static int check_dims(const gvqa_gat_dims* d, bool seq) {
    GVQA_REQUIRE(d, GVQA_E_INVALID, "gat: null dims");
    GVQA_REQUIRE(d->node_dim > 0 && d->edge_dim > 0 && d->out_channels > 0 && d->heads > 0 && d->ins_dim >= 0,
                 GVQA_E_INVALID, "gat: non-positive dimension");
    GVQA_REQUIRE(d->num_hops >= 1 && d->num_hops <= MAX_HOPS, GVQA_E_INVALID, "gat: num_hops must be in [1,%d]", MAX_HOPS);
    if (seq) {
        GVQA_REQUIRE(d->node_dim == d->out_channels, GVQA_E_INVALID,
                     "gat_seq: skip connection needs in_channels == out_channels (gat_skip.py:270)");
    } else {
        GVQA_REQUIRE(d->ins_dim == 0 && d->num_hops == 1, GVQA_E_INVALID, "gat_conv: ins_dim must be 0 and num_hops 1");
    }
    return GVQA_OK;
}

static int run_fold(const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops, float* Vn, float* Ve, float* Gw,
                    hipStream_t stream) {
    FoldArgs fa;
    memset(&fa, 0, sizeof(fa));
    for (int i = 0; i < d->num_hops; ++i) {
        GVQA_REQUIRE(hops[i].lin_l_weight && hops[i].lin_e_weight && hops[i].att_l && hops[i].att_r && hops[i].att_e,
                     GVQA_E_INVALID, "gat: hop %d has a null weight", i);
        fa.W_l[i] = hops[i].lin_l_weight; fa.W_e[i] = hops[i].lin_e_weight;
        fa.att_l[i] = hops[i].att_l; fa.att_r[i] = hops[i].att_r; fa.att_e[i] = hops[i].att_e;
    }
    fa.Vn = Vn; fa.Ve = Ve; fa.Gw = Gw;
    fa.Dn = d->node_dim; fa.De = d->edge_dim; fa.Di = d->ins_dim; fa.C = d->out_channels; fa.H = d->heads;
    fa.K = d->num_hops;
    StageTimer timer(GVQA_STAGE_FOLD, stream);
    int maxk = fa.Dn > fa.De ? fa.Dn : fa.De;
    if (fa.Di > maxk) maxk = fa.Di;
    hipLaunchKernelGGL(k_fold_att, dim3((unsigned)cdiv(maxk, 64), (unsigned)fa.H, (unsigned)fa.K * 4), dim3(256), 0,
                       stream, fa);
    if (fa.Di > 0)
        hipLaunchKernelGGL(k_fold_headmean, dim3((unsigned)cdiv((int64_t)fa.C * fa.Di, 256), 1, (unsigned)fa.K),
                           dim3(256), 0, stream, fa);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // namespace gvqa

extern "C" {

using namespace gvqa;

int gvqa_gat_mp_plan(const gvqa_graph* g, int32_t C, int32_t H, gvqa_mp_plan* out) {
    GVQA_REQUIRE(g && out && C > 0 && H > 0, GVQA_E_INVALID, "gat_mp_plan: bad argument");
    TilePlan p = plan_tiled(g, C, H);
    memset(out, 0, sizeof(*out));
    out->tiled = p.ok ? 1 : 0;
    if (p.ok) {
        out->channel_range = p.cw;
        out->stage_buffers = p.nbuf;
        out->lds_bytes = (int64_t)p.lds_bytes;
        out->blocks_per_graph = plan_parts(g->num_graphs, C, p);
        out->blocks_per_cu = (int32_t)(LDS_MAX / p.lds_bytes);
        out->stages_per_graph = (int32_t)(cdiv(C, p.cw) * (H + 1));
        out->accumulators = ((size_t)p.n_cap <= (size_t)2 * (MP_THREADS >> p.lpn_log)) ? 2 : MP_ITEMS;
    }
    return GVQA_OK;
}

int gvqa_gat_message_passing(const gvqa_graph* g, const gvqa_gat_mp_desc* d, void* ws, size_t ws_bytes, void* stream) {
    return launch_gat_mp(g, d, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

size_t gvqa_gat_seq_workspace_bytes(const gvqa_graph* g, const gvqa_gat_dims* d) {
    if (!g || !d) return 0;
    return seq_layout(g->num_nodes, g->num_edges, g->num_graphs, d, d->ins_dim >= 0 && d->num_hops >= 1 ? g : nullptr).total;
}

size_t gvqa_gat_conv_workspace_bytes(const gvqa_graph* g, const gvqa_gat_dims* d) {
    return gvqa_gat_seq_workspace_bytes(g, d);
}

int gvqa_gat_conv_forward(const gvqa_graph* g, const gvqa_gat_dims* d, const gvqa_gat_conv_params* p, const float* x,
                          const float* edge_attr, float* out, float* alpha_out, void* ws, size_t ws_bytes,
                          void* stream_) {
    GVQA_REQUIRE(g && p, GVQA_E_INVALID, "gat_conv: null argument");
    int rc = check_dims(d, false);
    if (rc) return rc;
    const int64_t N = g->num_nodes, E = g->num_edges, B = g->num_graphs;
    SeqLayout L = seq_layout(N, E, B, d);
    GVQA_REQUIRE(ws && ws_bytes >= L.total, GVQA_E_WORKSPACE, "gat_conv: workspace %zu < required %zu", ws_bytes, L.total);
    GVQA_REQUIRE((N == 0 || (x && out)) && (E == 0 || edge_attr), GVQA_E_INVALID, "gat_conv: null tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* base = static_cast<char*>(ws);
    auto P = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    const int H = d->heads, C = d->out_channels;

    const WeightCacheLayout WL = weight_cache_layout(d, weight_layout_id(proj_pieces(d, N, (int64_t)d->heads * d->out_channels, d->node_dim), false));
    float* Vn1 = reinterpret_cast<float*>(base + L.Vn + WL.Vn);
    float* Ve1 = reinterpret_cast<float*>(base + L.Vn + WL.Ve);
    rc = run_fold(d, p, Vn1, Ve1, nullptr, stream);
    if (rc) return rc;
    {   // a_e = edge_attr . V_e^T                       (gat_skip.py:150-151, folded)
        StageTimer t(GVQA_STAGE_EDGE_LOGIT, stream);
        rc = launch_linear(E, H, d->edge_dim, edge_attr, d->edge_dim, Ve1, d->edge_dim, nullptr, 0, P(L.a_edge), H, 1,
                           0, 0, 0, stream);
        if (rc) return rc;
    }
    {   // xp = x . W_l^T                                (gat_skip.py:133)
        StageTimer t(GVQA_STAGE_PROJ, stream);
        rc = launch_linear(N, (int64_t)H * C, d->node_dim, x, d->node_dim, p->lin_l_weight, d->node_dim, nullptr, 0,
                           P(L.xp), (int64_t)H * C, 1, 0, 0, 0, stream);
        if (rc) return rc;
    }
    {   // (a_l | a_r) = x . [V_l | V_r]                 (gat_skip.py:134-135, folded)
        StageTimer t(GVQA_STAGE_NODE_LOGIT, stream);
        rc = launch_linear(N, 2 * H, d->node_dim, x, d->node_dim, Vn1, d->node_dim, nullptr, 0, P(L.a_node), 2 * H, 1,
                           0, 0, 0, stream);
        if (rc) return rc;
    }
    gvqa_gat_mp_desc m = mp_desc_from(d, p);
    m.xp = P(L.xp); m.a_node = P(L.a_node); m.a_edge = P(L.a_edge); m.a_edge_stride = H;
    m.out = out; m.alpha_out = alpha_out;
    return launch_gat_mp(g, &m, P(L.alpha_csr), (size_t)E * H * sizeof(float), stream);
}

static int prepare_weights(const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops, int layout, char* cache, hipStream_t fold_stream,
                           hipStream_t stream) {
    const WeightCacheLayout W = weight_cache_layout(d, layout);
    const int H = d->heads, C = d->out_channels, K = d->num_hops, Dn = d->node_dim, Di = d->ins_dim;
    int rc = run_fold(d, hops, reinterpret_cast<float*>(cache + W.Vn), reinterpret_cast<float*>(cache + W.Ve),
                      Di > 0 ? reinterpret_cast<float*>(cache + W.Gw) : nullptr, fold_stream);
    if (rc) return rc;
    if (layout >= 0) {     // node-column weights of all hops -> packed pieces
        StageTimer t(GVQA_STAGE_PACK, stream);
        for (int i = 0; i < K; ++i) {
            GVQA_REQUIRE(hops[i].lin_l_weight, GVQA_E_INVALID, "gat: hop %d has a null weight", i);
            const int np = layout_pieces(layout);
            rc = (layout & 8) ? launch_hopagg_pack_w(H, C, Dn, hops[i].lin_l_weight, Dn + Di, cache + W.w6 + (size_t)i * W.w6_hop, stream)
                 : (layout & 4) ? launch_split_pack_heads2(H, C, 256 / H, Dn, hops[i].lin_l_weight, Dn + Di, cache + W.w6 + (size_t)i * W.w6_hop, stream)
                 : (layout & 1) ? launch_split_pack_heads(np, H, C, 256 / H, Dn, hops[i].lin_l_weight, Dn + Di, cache + W.w6 + (size_t)i * W.w6_hop, stream)
                              : launch_split_pack(np, (int64_t)H * C, Dn, hops[i].lin_l_weight, Dn + Di, cache + W.w6 + (size_t)i * W.w6_hop, stream);
            if (rc) return rc;
        }
    }
    if (W.epc_hop) {       // bias | BatchNorm scale | shift per output channel of every hop
        StageTimer t(GVQA_STAGE_PACK, stream);
        for (int i = 0; i < K; ++i) {
            rc = launch_hop2_consts(H, C, hops[i].bias, hops[i].bn_weight, hops[i].bn_bias, hops[i].bn_mean, hops[i].bn_var, d->bn_eps,
                                    reinterpret_cast<float*>(cache + W.epc + (size_t)i * W.epc_hop), stream);
            if (rc) return rc;
        }
    }
    if (W.epc_hop && (layout & 4) && d->node_dim == d->out_channels) {      // chained hops: bound constants (after the epilogue constants)
        StageTimer t(GVQA_STAGE_PACK, stream);
        for (int i = 0; i < K; ++i) {
            rc = launch_hop2_bound_consts(H, C, Dn, hops[i].lin_l_weight, Dn + Di, reinterpret_cast<const float*>(cache + W.epc + (size_t)i * W.epc_hop),
                                          reinterpret_cast<float*>(cache + W.bc) + 4 * i, stream);
            if (rc) return rc;
        }
    }
    if (W.gw2h_hop) {      // two-piece images of the per-graph term weights (after the fold, on its stream)
        StageTimer t(GVQA_STAGE_PACK, fold_stream);
        for (int i = 0; i < K; ++i) {
            rc = launch_split_pack(2, (int64_t)(C + H), Di, reinterpret_cast<const float*>(cache + W.Gw) + (int64_t)i * (C + H) * Di, Di,
                                   cache + W.gw2h + (size_t)i * W.gw2h_hop, fold_stream);
            if (rc) return rc;
        }
    }
    if (W.vn2h_hop) {      // two-piece images of the folded attention vectors (after the fold, on its stream)
        StageTimer t(GVQA_STAGE_PACK, fold_stream);
        for (int i = 0; i < K; ++i) {
            rc = launch_split_pack(2, 2 * (int64_t)H, Dn, reinterpret_cast<const float*>(cache + W.Vn) + (int64_t)i * 2 * H * Dn, Dn,
                                   cache + W.vn2h + (size_t)i * W.vn2h_hop, fold_stream);
            if (rc) return rc;
        }
    }
    return GVQA_OK;
}

static int gat_seq_forward_impl(const gvqa_graph* g, const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops, const float* x,
                                const float* edge_attr, const float* instr, float* out, float* alpha_out, float* hop_out,
                                float* bn_stats_out, void* ws, size_t ws_bytes, void* stream_, const void* wcache = nullptr,
                                size_t wcache_bytes = 0, int wcache_layout = -2) {
    GVQA_REQUIRE(g && hops, GVQA_E_INVALID, "gat_seq: null argument");
    int rc = check_dims(d, true);
    if (rc) return rc;
    GVQA_REQUIRE(g->finalized, GVQA_E_INVALID, "gat_seq: call gvqa_graph_finalize first");
    GVQA_REQUIRE(g->intra_graph, GVQA_E_UNSUPPORTED,
                 "gat_seq: an edge joins two graphs of the batch; use gvqa_gat_conv_forward on concatenated inputs");
    const int64_t N = g->num_nodes, E = g->num_edges, B = g->num_graphs;
    SeqLayout L = seq_layout(N, E, B, d, g);
    GVQA_REQUIRE(ws && ws_bytes >= L.total, GVQA_E_WORKSPACE, "gat_seq: workspace %zu < required %zu", ws_bytes, L.total);
    GVQA_REQUIRE((N == 0 || (x && out)) && (E == 0 || edge_attr) && (d->ins_dim == 0 || B == 0 || instr), GVQA_E_INVALID,
                 "gat_seq: null tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* base = static_cast<char*>(ws);
    auto P = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    const int H = d->heads, C = d->out_channels, K = d->num_hops, Dn = d->node_dim, De = d->edge_dim, Di = d->ins_dim;
    const int64_t Tld = (int64_t)align_up((size_t)(C + H), 4);
    if (N == 0) return GVQA_OK;

    // Everything that does not depend on the current hop's projection runs on a side stream, under
    // the MFMA-bound projection GEMM on the caller's stream: weight folding, the all-hops edge-logit
    // pass and the graph terms (hop 0), and each hop's node-logit product.
    SideStream* ss = side_stream(g->num_edges);
    hipStream_t aux = ss ? ss->stream : stream;
    if (ss) { rc = side_fork(ss, stream); if (rc) return rc; }
    const int np = proj_pieces(d, N, (int64_t)H * C, Dn);
    const bool split = np != 0;
    const bool fused = hop_fusion_applies(g, d);
    const int fcw = 256 / H;                                   // channels of every head per column block of the fused hop
    const bool hop2 = fused && hop2_applies(g, d);
    const bool chain8_ok = fused && fused_chain8_capable(g, d);        // (decides the cache layout: must not depend on the outputs asked for)
    const bool chain = ((hop2 && hop2_chain_capable(g, d)) || chain8_ok) && !hop_out && !bn_stats_out &&    // (per-hop fp32 outputs / batch statistics need fp32 rows)
                       split_pack_groups_logits_supported(2, 2 * H, Dn);
    // (batch-statistics BatchNorm needs fp32 rows between the passes; rows that are not 16-byte aligned -- the layout pass reads float4 --
    //  take the other hop kernels instead of an error return: ADVICE r04)
    const bool aggf = hopagg_applies(g, d) && !bn_stats_out && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const int need_layout = aggf ? LAYOUT_AGGFIRST : weight_layout_id(np, fused, hop2 || chain8_ok);
    // parameter-only products: from the caller's cache when it was prepared for the layout this batch needs, else computed
    // now into the workspace (the workspace slices have exactly the cache's sub-layout)
    const WeightCacheLayout WL = weight_cache_layout(d, need_layout);
    const bool cached = wcache && wcache_layout == need_layout && wcache_bytes >= WL.total;
    const char* wbase;
    if (cached) {
        wbase = static_cast<const char*>(wcache);
    } else {
        char* wtmp = base + L.Vn;                              // Vn | Ve | Gw | w6 are carved contiguously below
        rc = prepare_weights(d, hops, need_layout, wtmp, aux, stream);
        if (rc) return rc;
        wbase = wtmp;
    }
    const float* Vn_all = reinterpret_cast<const float*>(wbase + WL.Vn);
    const float* Ve_all = reinterpret_cast<const float*>(wbase + WL.Ve);
    const float* Gw_all = reinterpret_cast<const float*>(wbase + WL.Gw);
    const char* w6 = wbase + WL.w6;
    const size_t w6_hop = WL.w6_hop;
    {   // edge logit terms of ALL hops in one pass over edge_attr: [E, De] x [De, K*H]
        // (GVQA_OVERLAP = 2: this HBM-bound pass stays on the caller's stream, in front of the equally HBM-bound layout pass; only the
        //  latency-bound per-graph terms go to the side stream)
        hipStream_t es = (side_stream_mode(g->num_edges) == 2 && cached) ? stream : aux;
        StageTimer t(GVQA_STAGE_EDGE_LOGIT, es);
        rc = launch_linear(E, (int64_t)K * H, De, edge_attr, De, Ve_all, De, nullptr, 0, P(L.a_edge), (int64_t)K * H, 1, 0,
                           0, 0, es);
        if (rc) return rc;
    }
    if (Di > 0) {   // per-graph instruction terms of all hops: [K] x ([B, Di] x [Di, C+H])
        StageTimer t(GVQA_STAGE_GRAPH_TERM, aux);
        const bool term_split = np == 2 && WL.gw2h_hop && L.ipk_hop && B <= 65535 * 128 &&
                                2.0 * (double)K * (double)B * (double)(C + H) * (double)Di >= 1e6 * (double)get_option(GVQA_OPT_SPLIT3_MIN_MFLOP);
        if (term_split) {
            // ONE batched two-piece product on the fp16 matrix cores (config 3: the stage 85 -> 58 us): instruction rows packed
            // now (row scales), weights from the cache
            char* ipk = base + L.ipk;
            const int KBi = (int)cdiv(Di, 16);
            LinearEpilogue ep{};
            rc = launch_split_pack(2, (int64_t)K * B, Di, instr, Di, ipk, aux);
            if (rc) return rc;
            const float* a_inv = reinterpret_cast<const float*>(ipk + (size_t)(K * B / 32) * KBi * 2048);
            ep.zs_ia = B;
            ep.zs_a = (int64_t)(L.ipk_hop / sizeof(uint16_t));
            ep.zs_b = (int64_t)(WL.gw2h_hop / sizeof(uint16_t));
            ep.zs_ib = (int64_t)(WL.gw2h_hop / sizeof(float));
            ep.zs_c = (int64_t)B * Tld;
            const char* gw2h = wbase + WL.gw2h;
            const float* b_inv = reinterpret_cast<const float*>(gw2h + (size_t)cdiv((int64_t)(C + H), 32) * KBi * 2048);
            rc = launch_linear_split(2, B, C + H, Di, ipk, gw2h, ep, P(L.T), Tld, aux, K, a_inv, b_inv);
        } else {
            rc = launch_linear(B, C + H, Di, instr, Di, Gw_all, Di, nullptr, 0, P(L.T), Tld, K, (int64_t)B * Di,
                               (int64_t)(C + H) * Di, (int64_t)B * Tld, aux);
        }
        if (rc) return rc;
    }
    if (chain && Di > 0 && !aggf) {   // largest instruction-term magnitude per (hop, graph): part of the chained hops' output bounds (the aggregate-first kernel scales by exact maxima)
        StageTimer t(GVQA_STAGE_GRAPH_TERM, aux);
        rc = launch_rows_absmax((int64_t)K * B, C, P(L.T), Tld, P(L.Tmax), aux);
        if (rc) return rc;
    }
    if (aggf) {
        // ---- aggregate-first hops (hopagg.hip): x -> chunk-major rows + per-graph maxima once, then per hop TWO launches --
        // coefficient kernel (node logits from the chunks + segment softmax) and the hop kernel -- rows staying chunk-major.
        float* X4[2] = {P(L.x4a), P(L.x4b)};
        float* GM[2] = {P(L.gma), P(L.gmb)};
        const bool pk = agg_packed(g);            // packed row groups: every array of the batch below in the packed numbering, rows in / out through the map
        const int ngroups = agg_groups(g);
        const int NQ = (int)cdiv(Dn, 4);
        {
            StageTimer tp(GVQA_STAGE_PACK, stream);
            // (+ hop 0's node logits, out of the slab it moves the rows through)
            rc = launch_rows_to_x4(g, Dn, x, Dn, X4[0], GM[0], stream, Vn_all, P(L.a_node), pk);
            if (rc) return rc;
        }
        if (ss) { rc = side_join(ss, stream); if (rc) return rc; }
        const bool aggseq = hopagg_seq_applies(g, d) && !alpha_out && !hop_out;
        const int cp = hopagg_parts_applies(g, d) ? kHopaggParts : 1;         // column parts: node logits / maxima as one set per part, two buffers
        for (int i = 0; i < K; ++i) {
            const float* gterm = Di > 0 ? P(L.T) + (int64_t)i * B * Tld : nullptr;
            if (aggseq && i > 0) break;               // (one launch: hops 1 .. K - 1 compute their coefficients inside it)
            // every hop computes its attention coefficients in its own prologue: hop 0's node logits came with the layout pass, hop i's
            // with hop i - 1's rows (the one-launch form: between two hops, inside the workgroup) -- no coefficient kernel
            const bool in_prologue = true;
            {
                StageTimer t(GVQA_STAGE_PROJ, stream);
                HopAggArgs ha;
                memset(&ha, 0, sizeof(ha));
                ha.group_ptr = pk ? g->pk_row_group_ptr : g->row_group_ptr; ha.rowptr = pk ? g->pk_rowptr : g->rowptr;
                ha.csr_src = pk ? g->pk_csr_src : g->csr_src; ha.node_graph = pk ? g->pk_node_graph : g->node_graph;
                ha.row_map = pk ? g->pk_node_old : nullptr; ha.graph_old = pk ? g->pk_graph_old : nullptr;
                const int32_t* agg_csr_eid = pk ? g->pk_csr_eid : g->csr_eid;
                ha.alpha_csr = P(L.alpha_csr); ha.X4in = X4[i & 1];
                const char* wk = w6 + (size_t)i * w6_hop;
                ha.NCT = (int)cdiv(C, 32); ha.NQ = NQ; ha.C = C;
                ha.Wk = reinterpret_cast<const uint16_t*>(wk);
                ha.binv = reinterpret_cast<const float*>(wk + (size_t)ha.NCT * NQ * 2048);
                ha.epc = reinterpret_cast<const float*>(wbase + WL.epc + (size_t)i * WL.epc_hop); ha.epc_ld = hop2_consts_ld(H, C);
                ha.graph_term = gterm; ha.t_ld = Tld;
                ha.gmax_in = GM[i & 1];
                const bool last = i == K - 1;
                ha.X4out = last ? nullptr : X4[(i + 1) & 1];
                ha.gmax_out = last ? nullptr : GM[(i + 1) & 1];
                ha.out = hop_out ? hop_out + (int64_t)i * N * C : (last ? out : nullptr);
                ha.out_ld = C;
                ha.relu = hops[i].bn_weight != nullptr;
                // (column parts: hop i reads buffer i & 1 -- hop 0: the layout pass's single set --, writes buffer (i + 1) & 1: the workgroups
                //  of a row group read each other's sets, so the in-place hand-over of the one-workgroup form does not apply)
                const int64_t an_set = (int64_t)N * 2 * H;
                float* an_in = cp > 1 ? P(L.a_node) + (int64_t)(i & 1) * cp * an_set : P(L.a_node);
                float* an_out = cp > 1 ? P(L.a_node) + (int64_t)((i + 1) & 1) * cp * an_set : P(L.a_node);
                ha.parts_in = (cp > 1 && i > 0) ? cp : 1;
                ha.an_part_stride = an_set; ha.gm_part_stride = B;
                if (!last && !aggseq) {               // the next hop's node logits leave with the rows
                    ha.Vn_next = Vn_all + (int64_t)(i + 1) * 2 * H * Dn;
                    ha.a_node_out = an_out;
                }
                if (in_prologue) {                    // (in place: a workgroup reads its rows' logits before it writes the next hop's)
                    ha.alpha_csr = nullptr;
                    ha.a_node_in = an_in;
                    ha.a_edge = P(L.a_edge) + (int64_t)i * H; ha.a_edge_stride = (int64_t)K * H; ha.csr_eid = agg_csr_eid;
                    ha.alpha_out = alpha_out ? alpha_out + (int64_t)i * E * H : nullptr;
                    ha.slope = d->negative_slope;
                }
                GVQA_REQUIRE(!hops[i].bn_weight || (hops[i].bn_bias && hops[i].bn_mean && hops[i].bn_var), GVQA_E_INVALID,
                             "gat_seq: BatchNorm needs weight, bias, running_mean and running_var");
                if (aggseq) {
                    // ---- the K hops as ONE launch: a workgroup walks all hops of its row group (rows ping-pong between the two
                    // chunk-major buffers, coefficients of hops 1 .. K - 1 computed in the workgroup)
                    HopAggSeq hs;
                    memset(&hs, 0, sizeof(hs));
                    hs.Wk = reinterpret_cast<const uint16_t*>(w6); hs.w_hop_bytes = (int64_t)w6_hop; hs.binv_off_bytes = (int64_t)ha.NCT * NQ * 2048;
                    hs.epc = reinterpret_cast<const float*>(wbase + WL.epc); hs.epc_hop = (int64_t)(WL.epc_hop / sizeof(float));
                    hs.graph_term = Di > 0 ? P(L.T) : nullptr; hs.t_hop = (int64_t)B * Tld;
                    hs.Vn = Vn_all;
                    hs.csr_eid = agg_csr_eid; hs.a_edge = P(L.a_edge); hs.a_edge_stride = (int64_t)K * H;
                    hs.X4a = X4[0]; hs.X4b = X4[1];
                    hs.slope = d->negative_slope; hs.K = K;
                    for (int j = 0; j < K; ++j) {
                        GVQA_REQUIRE(!hops[j].bn_weight || (hops[j].bn_bias && hops[j].bn_mean && hops[j].bn_var), GVQA_E_INVALID,
                                     "gat_seq: BatchNorm needs weight, bias, running_mean and running_var");
                        if (hops[j].bn_weight) hs.relu_mask |= 1u << j;
                    }
                    ha.out = out; ha.X4out = nullptr; ha.gmax_out = nullptr;
                    ha.alpha_csr = nullptr; ha.a_node_in = P(L.a_node); ha.a_node_out = nullptr; ha.Vn_next = nullptr;
                    rc = launch_hopagg_seq(H, ha, hs, ngroups, stream);
                    if (rc) return rc;
                    continue;
                }
                rc = launch_hopagg(H, ha, ngroups, stream, cp);
                if (rc) return rc;
            }
        }
        if (hop_out) {
            StageTimer t(GVQA_STAGE_OTHER, stream);
            GVQA_HIP_CHECK(hipMemcpyAsync(out, hop_out + (int64_t)(K - 1) * N * C, (size_t)N * C * sizeof(float), hipMemcpyDeviceToDevice, stream));
        }
        return GVQA_OK;
    }
    char* a6 = base + L.a6;
    const int ncb_chain = (int)cdiv(C, fcw);
    // chained 8-wave hops with the attention coefficients computed inside the hop kernel (CHN = 2): no coefficient launch at all
    const bool ic = chain && chain8_ok && !hop2 && chain8_in_kernel_coeffs(g, d);
    const float* h = x;
    bool aux_pending = ss != nullptr;                 // the side stream's pre-hop work has not been joined yet
    for (int i = 0; i < K; ++i) {
        if (chain) a6 = base + ((i & 1) ? L.a6b : L.a6);          // hop i reads the operand hop i - 1 (or the pack pass) left
        float* h_next;
        if (hop_out) h_next = hop_out + (int64_t)i * N * C;
        else if (i == K - 1) h_next = out;
        else h_next = (i & 1) ? P(L.h1) : P(L.h0);
        if (ss && i > 0 && !fused) { rc = side_fork(ss, stream); if (rc) return rc; }     // (unfused hops: the node-logit product runs beside the projection) h of this hop is ready on `stream`
        const bool logits_in_pack = fused && split_pack_groups_logits_supported(np, 2 * H, Dn);
        if (chain && i > 0) {
            // nothing to prepare: packed rows, partial logits and maxima of h came out of the previous hop's launch
        } else if (logits_in_pack) {
            // row-group slots of h for the fused hop AND (a_l | a_r) = h . [V_l | V_r] in one pass over h
            if (ss && aux_pending && !cached) { rc = side_join(ss, stream); if (rc) return rc; aux_pending = false; }      // Vn comes from the fold on the side stream (hop 0, uncached weights)
            StageTimer tp(GVQA_STAGE_PACK, stream);
            rc = launch_split_pack_groups(np, g->num_row_groups, g->row_group_ptr, Dn, h, Dn, a6, Vn_all + (int64_t)i * 2 * H * Dn, 2 * H,
                                          P(L.a_node), stream, WL.vn2h_hop ? wbase + WL.vn2h + (size_t)i * WL.vn2h_hop : nullptr);
            if (rc) return rc;
        } else {   // (a_l | a_r) node halves = h . [V_l | V_r]     (side stream)
            if (ss && fused && i > 0) { rc = side_fork(ss, stream); if (rc) return rc; aux_pending = true; }     // h of this hop is ready on `stream`
            StageTimer t(GVQA_STAGE_NODE_LOGIT, aux);
            rc = launch_linear(N, 2 * H, Dn, h, Dn, Vn_all + (int64_t)i * 2 * H * Dn, Dn, nullptr, 0, P(L.a_node), 2 * H,
                               1, 0, 0, 0, aux);
            if (rc) return rc;
        }
        if (fused) {
            // attention coefficients (CSR order) -> row-group slots of h -> projection + aggregation + epilogue in one kernel
            if (ss && aux_pending) { rc = side_join(ss, stream); if (rc) return rc; aux_pending = false; }      // edge logits / graph terms of the side stream: before the first coefficient kernel
            const bool train_bn = bn_stats_out && hops[i].bn_weight;
            MpArgs a;
            memset(&a, 0, sizeof(a));
            a.rowptr = g->rowptr; a.csr_src = g->csr_src; a.csr_eid = g->csr_eid; a.node_graph = g->node_graph; a.graph_ptr = g->graph_ptr;
            a.a_node = P(L.a_node); a.a_edge = P(L.a_edge) + (int64_t)i * H; a.a_edge_stride = (int64_t)K * H;
            a.graph_term = Di > 0 ? P(L.T) + (int64_t)i * B * Tld : nullptr; a.t_ld = Tld;
            a.alpha_out = alpha_out ? alpha_out + (int64_t)i * E * H : nullptr;
            a.alpha_csr = P(L.alpha_csr);
            a.N = (int)N; a.C = C; a.slope = d->negative_slope;
            if (!ic) {   // (when the logits ride on the pack pass, it ran above, before the coefficients; chained hops: the coefficient kernel
                //  computes the logits itself from the packed rows the previous hop left)
                StageTimer t(GVQA_STAGE_ALPHA, stream);
                ChainScaleArgs cs;
                memset(&cs, 0, sizeof(cs));
                if (chain && i > 0 && i < K - 1) {        // the hop that follows leaves packed rows: their scales are decided here
                    cs.PMin = P(L.PM) + (size_t)(i & 1) * ncb_chain * B;
                    cs.Tmax = Di > 0 ? P(L.Tmax) + (size_t)i * B : nullptr;
                    cs.bc = reinterpret_cast<const float*>(wbase + WL.bc) + 4 * i;
                    cs.gscale = P(L.gscale);
                    cs.a_inv_next = reinterpret_cast<float*>(base + ((i & 1) ? L.a6 : L.a6b) + (size_t)g->num_row_groups * 4 * cdiv(Dn, 16) * 2048);
                    cs.ncb = ncb_chain; cs.B = (int)B;
                }
                rc = (chain && i > 0) ? launch_alpha_packed(a, H, stream, g, a6, (int)cdiv(Dn, 16), wbase + WL.vn2h + (size_t)i * WL.vn2h_hop, cs)
                                      : launch_alpha(a, H, stream, g);
                if (rc) return rc;
            }
            if (!logits_in_pack && !(chain && i > 0)) {
                StageTimer tp(GVQA_STAGE_PACK, stream);
                rc = launch_split_pack_groups(np, g->num_row_groups, g->row_group_ptr, Dn, h, Dn, a6, nullptr, 0, nullptr, stream);
                if (rc) return rc;
            }
            FusedHopArgs f;
            memset(&f, 0, sizeof(f));
            f.group_ptr = g->row_group_ptr; f.num_groups = g->num_row_groups; f.row_order = g->row_group_order;
            f.rowptr = g->rowptr; f.csr_src = g->csr_src; f.alpha_csr = P(L.alpha_csr); f.node_graph = g->node_graph;
            f.graph_term = a.graph_term; f.t_ld = Tld;
            f.bias = hops[i].bias;
            if (!train_bn) { f.bn_w = hops[i].bn_weight; f.bn_b = hops[i].bn_bias; f.bn_m = hops[i].bn_mean; f.bn_v = hops[i].bn_var; }
            GVQA_REQUIRE(!f.bn_w || (f.bn_b && f.bn_m && f.bn_v), GVQA_E_INVALID,
                         "gat_seq: BatchNorm needs weight, bias, running_mean and running_var");
            f.skip = h; f.skip_ld = C; f.out = h_next; f.out_ld = C;
            f.H = H; f.C = C; f.cw = fcw; f.e_cap = g->max_row_group_edges; f.bn_eps = d->bn_eps;
            {
                StageTimer t(GVQA_STAGE_PROJ, stream);
                Hop2ChainDesc cd;
                memset(&cd, 0, sizeof(cd));
                if (chain) {
                    if (i < K - 1) {
                        cd.Pnext = base + ((i & 1) ? L.a6 : L.a6b);
                        cd.PMout = P(L.PM) + (size_t)((i + 1) & 1) * ncb_chain * B;
                    }
                    cd.PMin = i > 0 ? P(L.PM) + (size_t)(i & 1) * ncb_chain * B : nullptr;
                    cd.gscale = (i > 0 && i < K - 1) ? P(L.gscale) : nullptr;
                    cd.Tmax = Di > 0 ? P(L.Tmax) + (size_t)i * B : nullptr;
                    cd.bc = reinterpret_cast<const float*>(wbase + WL.bc) + 4 * i;
                    cd.graph_ptr = g->graph_ptr; cd.B = (int)B; cd.N = (int)N;
                    if (ic) {
                        // hop 0: ONE set of node logits, from the pack pass; hop i > 0: one set per column block of hop i - 1.  Output
                        // scales of hops 1 .. K - 2 from the per-graph maxima hop i - 1 left (PMin), inside the kernel.
                        cd.gscale = nullptr;
                        f.alpha_csr = nullptr;
                        f.ic_csr_eid = g->csr_eid;
                        f.ic_a_edge = a.a_edge; f.ic_a_edge_stride = a.a_edge_stride;
                        f.ic_lp_in = i == 0 ? P(L.a_node) : P((i & 1) ? L.lp1 : L.lp0);
                        f.ic_parts_in = i == 0 ? 1 : ncb_chain;
                        f.ic_lp_stride = (int64_t)N * 2 * H;
                        f.ic_lp_out = i < K - 1 ? P((i & 1) ? L.lp0 : L.lp1) : nullptr;
                        f.ic_vn_next = i < K - 1 ? Vn_all + (int64_t)(i + 1) * 2 * H * Dn : nullptr;
                        f.ic_pmin = (i > 0 && i < K - 1) ? cd.PMin : nullptr;
                        f.ic_alpha_out = a.alpha_out;
                        f.ic_slope = d->negative_slope;
                    }
                }
                rc = hop2 ? launch_hop2(Dn, a6, w6 + (size_t)i * w6_hop, f, reinterpret_cast<const float*>(wbase + WL.epc + (size_t)i * WL.epc_hop),
                                        chain ? &cd : nullptr, stream)
                          : launch_hop_fused_split(np, Dn, a6, w6 + (size_t)i * w6_hop, f, stream, chain ? &cd : nullptr);
                if (rc) return rc;
            }
            if (train_bn) {
                GVQA_REQUIRE(hops[i].bn_bias, GVQA_E_INVALID, "gat_seq: BatchNorm needs weight and bias");
                rc = bn_train_forward(N, C, h_next, &hops[i], d->bn_eps, bn_stats_out + (int64_t)i * 2 * C, P(L.bn_partial),
                                      gvqa_bn_train_workspace_bytes(N, C), stream);
                if (rc) return rc;
            }
            h = h_next;
            continue;
        }
        {   // xp = h . W_l[:, :Dn]^T   (node half of gat_skip.py:133; instruction half is in T)
            StageTimer t(GVQA_STAGE_PROJ, stream);
            if (split) {
                {
                    StageTimer tp(GVQA_STAGE_PACK, stream);
                    rc = launch_split_pack(np, N, Dn, h, Dn, a6, stream);
                    if (rc) return rc;
                }
                LinearEpilogue ep0{nullptr, nullptr, 0, nullptr, 0, 0};
                rc = launch_linear_split(np, N, (int64_t)H * C, Dn, a6, w6 + (size_t)i * w6_hop, ep0, P(L.xp), (int64_t)H * C, stream);
                if (rc) return rc;
            } else {
                rc = launch_linear(N, (int64_t)H * C, Dn, h, Dn, hops[i].lin_l_weight, Dn + Di, nullptr, 0, P(L.xp),
                                   (int64_t)H * C, 1, 0, 0, 0, stream);
                if (rc) return rc;
            }
        }
        if (ss) { rc = side_join(ss, stream); if (rc) return rc; }
        gvqa_gat_mp_desc m = mp_desc_from(d, &hops[i]);
        const bool train_bn = bn_stats_out && hops[i].bn_weight;
        if (train_bn) { m.bn_weight = m.bn_bias = m.bn_mean = m.bn_var = nullptr; }   // BN applied after the batch statistics
        m.xp = P(L.xp); m.a_node = P(L.a_node);
        m.a_edge = P(L.a_edge) + (int64_t)i * H; m.a_edge_stride = (int64_t)K * H;
        m.graph_term = Di > 0 ? P(L.T) + (int64_t)i * B * Tld : nullptr; m.graph_term_ld = Tld;
        m.skip = h; m.out = h_next;
        m.alpha_out = alpha_out ? alpha_out + (int64_t)i * E * H : nullptr;
        rc = launch_gat_mp(g, &m, P(L.alpha_csr), (size_t)E * H * sizeof(float), stream);
        if (rc) return rc;
        if (train_bn) {
            GVQA_REQUIRE(hops[i].bn_bias, GVQA_E_INVALID, "gat_seq: BatchNorm needs weight and bias");
            float* st = bn_stats_out + (int64_t)i * 2 * C;
            rc = bn_train_forward(N, C, h_next, &hops[i], d->bn_eps, st, P(L.bn_partial),
                                  gvqa_bn_train_workspace_bytes(N, C), stream);
            if (rc) return rc;
        }
        h = h_next;
    }
    if (hop_out) {
        StageTimer t(GVQA_STAGE_OTHER, stream);
        GVQA_HIP_CHECK(hipMemcpyAsync(out, hop_out + (int64_t)(K - 1) * N * C, (size_t)N * C * sizeof(float),
                                      hipMemcpyDeviceToDevice, stream));
    }
    return GVQA_OK;
}

int gvqa_gat_seq_forward(const gvqa_graph* g, const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops, const float* x,
                         const float* edge_attr, const float* instr, float* out, float* alpha_out, float* hop_out,
                         void* ws, size_t ws_bytes, void* stream) {
    return gat_seq_forward_impl(g, d, hops, x, edge_attr, instr, out, alpha_out, hop_out, nullptr, ws, ws_bytes, stream);
}

size_t gvqa_gat_seq_weight_cache_bytes(const gvqa_gat_dims* d, int32_t layout) {
    if (!d || check_dims(d, true) || layout < -1 || layout > 15) return 0;
    return weight_cache_layout(d, layout).total;
}

int gvqa_gat_seq_weight_layout(const gvqa_graph* g, const gvqa_gat_dims* d) {
    if (!g || !d || check_dims(d, true)) return -1;
    if (hopagg_applies(g, d)) return LAYOUT_AGGFIRST;
    const bool fused = hop_fusion_applies(g, d);
    return weight_layout_id(proj_pieces(d, g->num_nodes, (int64_t)d->heads * d->out_channels, d->node_dim), fused,
                            fused && (hop2_applies(g, d) || fused_chain8_capable(g, d)));
}

int gvqa_gat_seq_hop_kernel(const gvqa_graph* g, const gvqa_gat_dims* d) {
    if (!g || !d || check_dims(d, true) || !g->finalized) return GVQA_E_INVALID;
    if (!g->intra_graph) return GVQA_E_UNSUPPORTED;
    if (hopagg_seq_applies(g, d)) return GVQA_HOP_AGGREGATE_FIRST_SEQ;
    if (hopagg_parts_applies(g, d)) return GVQA_HOP_AGGREGATE_FIRST_PARTS;
    if (hopagg_applies(g, d)) return GVQA_HOP_AGGREGATE_FIRST;
    if (!hop_fusion_applies(g, d)) return GVQA_HOP_UNFUSED;
    const bool logits = split_pack_groups_logits_supported(2, 2 * d->heads, d->node_dim);
    if (hop2_applies(g, d)) return (hop2_chain_capable(g, d) && logits) ? GVQA_HOP_PERSISTENT_CHAINED : GVQA_HOP_PERSISTENT;
    return (fused_chain8_capable(g, d) && logits) ? GVQA_HOP_FUSED8_CHAINED : GVQA_HOP_FUSED8;
}

int gvqa_gat_seq_prepare_weights(const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops, int32_t layout, void* cache,
                                 size_t cache_bytes, void* stream) {
    GVQA_REQUIRE(hops && cache, GVQA_E_INVALID, "gat_seq_prepare_weights: null argument");
    int rc = check_dims(d, true);
    if (rc) return rc;
    GVQA_REQUIRE(layout >= -1 && layout <= 15, GVQA_E_INVALID, "gat_seq_prepare_weights: layout must be -1 .. 15");
    GVQA_REQUIRE(cache_bytes >= weight_cache_layout(d, layout).total, GVQA_E_WORKSPACE, "gat_seq_prepare_weights: cache too small");
    GVQA_REQUIRE((reinterpret_cast<uintptr_t>(cache) & 255) == 0, GVQA_E_INVALID, "gat_seq_prepare_weights: cache must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return prepare_weights(d, hops, layout, static_cast<char*>(cache), s, s);
}

int gvqa_gat_seq_forward_cached(const gvqa_graph* g, const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops, const float* x,
                                const float* edge_attr, const float* instr, float* out, float* alpha_out, float* hop_out,
                                const void* weight_cache, size_t weight_cache_bytes, int32_t weight_cache_layout_id, void* ws,
                                size_t ws_bytes, void* stream) {
    return gat_seq_forward_impl(g, d, hops, x, edge_attr, instr, out, alpha_out, hop_out, nullptr, ws, ws_bytes, stream,
                                weight_cache, weight_cache_bytes, weight_cache_layout_id);
}

int gvqa_gat_seq_forward_trainbn(const gvqa_graph* g, const gvqa_gat_dims* d, const gvqa_gat_conv_params* hops, const float* x,
                                 const float* edge_attr, const float* instr, float* out, float* bn_stats_out, void* ws,
                                 size_t ws_bytes, void* stream) {
    GVQA_REQUIRE(bn_stats_out, GVQA_E_INVALID, "gat_seq_trainbn: null statistics buffer");
    return gat_seq_forward_impl(g, d, hops, x, edge_attr, instr, out, nullptr, nullptr, bn_stats_out, ws, ws_bytes, stream);
}

}  // extern "C"
