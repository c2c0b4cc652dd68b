// The GAT hop "aggregate first": heads concatenated along K (gfx950).
//
// Same math as the fused hops of split3.hip / hop2.hip (/root/reference gat_skip.py:133,155-168,270-275), re-associated:
//
//     out[i, c] = (1/H) sum_h sum_{e -> i} alpha[e, h] (h W_h^T)[src_e, c]                     (project, then aggregate)
//               = (1/H) sum_h sum_k ( sum_{e -> i} alpha[e, h] x[src_e, k] ) W[h C + c, k]     (aggregate, then project)
//               = (1/H) sum_{k'} A'[i, k'] W'[c, k'],   k' = (k, h),  A'[i, (k, h)] = sum_{e -> i} alpha[e, h] x[src_e, k]
//
// -- ONE GEMM with K' = H Dn whose A operand is the attention-weighted neighbour sum, formed on the fly.  What that buys on this
// chip: the projected features xp ([rows, H C]: a 128 KiB fp32 LDS image per row group in the other two kernels) never exist,
// not even in LDS; the aggregation is no longer an epilogue behind the matrix-core loop (27 % of the 8-wave kernel's time,
// run with the matrix cores idle) but ~60 VALU instructions per lane and K step INSIDE the loop, beside the other waves' MFMAs;
// the epilogue is register -> global (scale, per-graph term, bias, skip, BatchNorm, ReLU); and a workgroup owns ALL C output
// columns of its 128 rows, so the aggregation is computed once per row group, every weight tile is shared by all eight waves,
// and a row group's result (per-graph maxima, rows) is complete when its workgroup ends.
//
//   * Workgroup = one row group (<= 128 rows, whole graphs: every neighbour row is in the tile) x all C <= 512 columns; 8 waves
//     as 2 (64-row halves) x 4 (128-column slices), 8 accumulators of 32 x 32 per wave.  One workgroup per CU (LDS).
//   * K step = 16 k' = KPH = 16 / H node columns x H heads.  Per step the workgroup DMAs the step's weight tiles (16 column tiles
//     x two fp16 pieces = 32 KiB, three-stage ring) and the next x chunk (128 rows x KPH floats = 2 KiB, four-stage ring:
//     x lives in HBM "chunk-major" -- [row group][k / KPH][128 rows][KPH] -- so a chunk is one contiguous 2 KiB read).
//   * Producer: lane (node i, head h) of the 512 threads keeps its node's first 8 in-edges (source slot, alpha) in REGISTERS for
//     the whole tile (they do not change from step to step; further edges come from the CSR slice in LDS) and per step forms
//     sum_e alpha_e x[src_e, 4 columns] from the x chunk in LDS, scales it by the graph's power-of-two scale, splits it into two
//     fp16 pieces and writes its 2 x 8 bytes straight into the MFMA A-fragment image of the NEXT step.
//     Scale: |A'| <= max |x| over the graph (the softmax weights sum to <= 1), so the EXACT per-graph maximum of the input rows --
//     left by the previous hop's epilogue (first hop: by the layout pass) -- anchors the two pieces; no a-priori output bound.
//   * Weights: packed K'-concatenated, W'[c, 16 s + KPH h + kk] = W[h C + c, KPH s + kk], one power-of-two scale per OUTPUT
//     COLUMN (not per block: a weight row far below its neighbours keeps its 22 bits).
//
// H = 4 only (KPH = 4: a chunk row is one float4); other head counts take the other hop kernels.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#include "gemm_tile.h"

namespace gvqa {

typedef _Float16 ha_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ha_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ha_f16x2 __attribute__((ext_vector_type(2)));
typedef float ha_f32x2 __attribute__((ext_vector_type(2)));

constexpr int HA_ROWS = 128;          // rows of a row group / tile
constexpr int HA_DMAX = 8;            // in-edges of a node kept in registers, fetched by the node's quad (one row per lane and batch of four)
constexpr int HA_NOV = 8;             // ... and the next ones ("overflow": source slot + coefficient in registers too, one row read per lane and edge)
constexpr int HA_ECAP = 1024;         // in-edges of a row group held in LDS (beyond the registers' 8 per node)

// ------------------------------------------------------------------------------------------------------------------------------
// Weights -> K'-concatenated packed fragments.  One block per 32-channel column tile: Wk[ct][s][piece][lane][8] with lane
// (m, khalf) holding W'[32 ct + m, 16 s + 8 khalf + e], e = 0..7, scaled by the channel's power of two; binv[c] = its inverse.
__global__ __launch_bounds__(256) void k_hopagg_pack_w(int H, int C, int Dn, int NQ, const float* __restrict__ W, int64_t ldw,
                                                       uint16_t* __restrict__ out, float* __restrict__ binv) {
    __shared__ float mx_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ct = blockIdx.x, KPH = 16 / H;
    if (tid < 32) mx_s[tid] = 0.f;
    __syncthreads();
    // largest magnitude of every output channel over its H weight rows: a wave per (channel, head) row, lanes over k
    for (int rr = wave; rr < 32 * H; rr += 4) {
        const int m = rr / H, h = rr - m * H, c = ct * 32 + m;
        float v = 0.f;
        if (c < C) {
            const float* wrow = W + ((int64_t)h * C + c) * ldw;
            for (int k = lane; k < Dn; k += 64) v = fmaxf(v, fabsf(wrow[k]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(&mx_s[m]), __float_as_uint(v));
    }
    __syncthreads();
    const int m = lane & 31, khalf = lane >> 5, c = ct * 32 + m;
    const int ex = split2h_exponent(mx_s[m]);
    const float scale = pow2i(ex);
    if (tid < 32) binv[ct * 32 + tid] = pow2i(-split2h_exponent(mx_s[tid]));
    for (int s = wave; s < NQ; s += 4) {
        ha_f16x8 p0, p1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kp = 8 * khalf + e, h = kp / KPH, kk = kp - h * KPH, k = s * KPH + kk;
            const float w = (c < C && k < Dn) ? W[((int64_t)h * C + c) * ldw + k] * scale : 0.f;
            const _Float16 hi = (_Float16)w;
            p0[e] = hi;
            p1[e] = (_Float16)(w - (float)hi);
        }
        uint16_t* dst = out + ((int64_t)(ct * NQ + s) * 2) * 512 + lane * 8;
        *reinterpret_cast<uint4*>(dst) = __builtin_bit_cast(uint4, p0);
        *reinterpret_cast<uint4*>(dst + 512) = __builtin_bit_cast(uint4, p1);
    }
}

size_t hopagg_packed_w_bytes(int C, int Dn, int H) {
    const size_t nct = (size_t)cdiv(C, 32), nq = (size_t)cdiv(Dn, 16 / H);
    return nct * nq * 2048 + align_up(nct * 32 * sizeof(float), 256);
}

int launch_hopagg_pack_w(int H, int C, int Dn, const float* W, int64_t ldw, void* packed, hipStream_t stream) {
    GVQA_REQUIRE(W && packed && (H == 1 || H == 2 || H == 4 || H == 8) && C > 0 && Dn > 0, GVQA_E_INVALID, "hopagg_pack_w: bad argument");
    const int nct = (int)cdiv(C, 32), nq = (int)cdiv(Dn, 16 / H);
    float* binv = reinterpret_cast<float*>(static_cast<char*>(packed) + (size_t)nct * nq * 2048);
    hipLaunchKernelGGL(k_hopagg_pack_w, dim3((unsigned)nct), dim3(256), 0, stream, H, C, Dn, nq, W, ldw, static_cast<uint16_t*>(packed), binv);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Row-major fp32 rows -> the chunk-major layout X4[group][k / 4][128 slots][4] (slots past the group's rows: zeros), and the largest
// magnitude of every graph's rows -> gmax[B].  One block per row group; the rows go through LDS 128 columns at a time so that
// both sides move whole lines (reads: 512 B per row; writes: a chunk = 2 KiB contiguous).
// Vn != NULL: the node logits of hop 0, a_node[node, j] = sum_k x[node, k] Vn[j, k], j < 8 (folded a_l | a_r vectors, gat_skip.py:134-135),
// leave with the rows (out of the slab in LDS: thread (row, quarter) takes every fourth chunk) -- hop 0 needs no pass of its own over x.
__global__ __launch_bounds__(512) void k_rows_to_x4(const int32_t* __restrict__ group_ptr, const int32_t* __restrict__ node_graph, int D, int NQ,
                                                    const float* __restrict__ X, int64_t ld, float* __restrict__ X4, float* __restrict__ gmax,
                                                    const float* __restrict__ Vn, float* __restrict__ a_node, const int32_t* __restrict__ row_map) {
    __shared__ float4 slab[HA_ROWS][33];         // 32 chunks (+1: the column walk of the write phase is conflict-free)
    __shared__ unsigned gm_s[HA_ROWS];
    __shared__ float4 vn_s[8][32];               // Vn columns of the current 128-column block
    ha_f32x2 lg[4][8];                           // thread (r4 = tid & 31, pj = tid >> 5): rows r4 + 32 k, chunks pj and pj + 16 of every block
#pragma unroll                                   // (even / odd columns in separate packed chains: v_pk_fma_f32, half the VALU issue of the phase)
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) lg[k][j] = ha_f32x2{0.f, 0.f};
    const int tid = threadIdx.x, t = blockIdx.x;
    const int ns = group_ptr[t], cnt = group_ptr[t + 1] - ns;
    const int gf = node_graph[ns];
    if (tid < HA_ROWS) gm_s[tid] = 0u;
    __syncthreads();
    float rowmax[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rowmax[u] = 0.f;
    // read: thread (r, p) = 16 bytes of row r; 32 threads cover 512 contiguous bytes of a row.  The loads of block q0 + 32 are issued as soon
    // as block q0 sits in the slab -- they travel under its node-logit FMAs, its chunk writes and both barriers (issued at the top of the
    // iteration, every block's read latency was exposed: 82 us for 268 MB)
    float4 pre[8], prev = make_float4(0.f, 0.f, 0.f, 0.f);
    // (packed row groups: row r of the group is row row_map[ns + r] of X -- whole rows still move as 512-byte segments)
    int xrow[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = (u * 512 + tid) >> 5;
        xrow[u] = r < cnt ? (row_map ? row_map[ns + r] : ns + r) : 0;
    }
    auto load_block = [&](int q0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = u * 512 + tid, r = idx >> 5, p = idx & 31, q = q0 + p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < cnt && q < NQ) {
                const float* src = X + (int64_t)xrow[u] * ld + q * 4;
                if (q * 4 + 4 <= D) v = *reinterpret_cast<const float4*>(src);
                else { v.x = src[0]; if (q * 4 + 1 < D) v.y = src[1]; if (q * 4 + 2 < D) v.z = src[2]; }
            }
            pre[u] = v;
        }
        if (Vn && tid < 256) {
            const int j = tid >> 5, p = tid & 31, k = (q0 + p) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k + 4 <= D) v = *reinterpret_cast<const float4*>(Vn + (int64_t)j * D + k);
            else if (k < D) { v.x = Vn[(int64_t)j * D + k]; if (k + 1 < D) v.y = Vn[(int64_t)j * D + k + 1]; if (k + 2 < D) v.z = Vn[(int64_t)j * D + k + 2]; }
            prev = v;
        }
    };
    load_block(0);
    for (int q0 = 0; q0 < NQ; q0 += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = u * 512 + tid, r = idx >> 5, p = idx & 31;
            const float4 v = pre[u];
            slab[r][p] = v;
            rowmax[u] = fmaxf(rowmax[u], fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        if (Vn && tid < 256) vn_s[tid >> 5][tid & 31] = prev;
        if (q0 + 32 < NQ) load_block(q0 + 32);
        __syncthreads();
        if (Vn) {                                // (four rows per Vn read: the phase is LDS-bound)
            const int r4 = tid & 31, pj = tid >> 5;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int p = pj + 16 * pp;
                float4 x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = slab[r4 + 32 * k][p];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 w = vn_s[j][p];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lg[k][j] = ha_f32x2{x[k].x, x[k].y} * ha_f32x2{w.x, w.y} + lg[k][j];
                        lg[k][j] = ha_f32x2{x[k].z, x[k].w} * ha_f32x2{w.z, w.w} + lg[k][j];
                    }
                }
            }
        }
        // write: thread (p, r): chunk q0 + p, slot r -- 128 consecutive threads cover a chunk's 2 KiB
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = u * 512 + tid, p = idx >> 7, r = idx & 127, q = q0 + p;
            if (q < NQ) *reinterpret_cast<float4*>(X4 + (((int64_t)t * NQ + q) * HA_ROWS + r) * 4) = slab[r][p];
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = (u * 512 + tid) >> 5;
        if (r < cnt) atomicMax(&gm_s[node_graph[ns + r] - gf], __float_as_uint(rowmax[u]));
    }
    __syncthreads();
    const int ngl = node_graph[ns + cnt - 1] - gf + 1;
    if (tid < ngl) gmax[gf + tid] = __uint_as_float(gm_s[tid]);
    if (Vn) {                                    // the sixteen chunk classes of a row meet through the slab's first 64 KiB
        float* part = reinterpret_cast<float*>(&slab[0][0]);           // [16][128][8]
        const int r4 = tid & 31, pj = tid >> 5;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float* dst = part + (pj * HA_ROWS + r4 + 32 * k) * 8;
            *reinterpret_cast<float4*>(dst) = make_float4(lg[k][0][0] + lg[k][0][1], lg[k][1][0] + lg[k][1][1], lg[k][2][0] + lg[k][2][1], lg[k][3][0] + lg[k][3][1]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(lg[k][4][0] + lg[k][4][1], lg[k][5][0] + lg[k][5][1], lg[k][6][0] + lg[k][6][1], lg[k][7][0] + lg[k][7][1]);
        }
        __syncthreads();
        for (int it = tid; it < cnt * 8; it += 512) {
            float sacc = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) sacc += part[c * HA_ROWS * 8 + it];
            a_node[(int64_t)ns * 8 + it] = sacc;
        }
    }
}

int launch_rows_to_x4(const gvqa_graph* g, int D, const float* X, int64_t ld, float* X4, float* gmax, hipStream_t stream, const float* Vn, float* a_node,
                      bool packed) {
    GVQA_REQUIRE(g && g->num_row_groups > 0 && X && X4 && gmax && D > 0 && ld >= D && (ld % 4) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0,
                 GVQA_E_INVALID, "rows_to_x4: bad argument");
    GVQA_REQUIRE(!Vn == !a_node, GVQA_E_INVALID, "rows_to_x4: node logits need both the folded vectors and their destination");
    GVQA_REQUIRE(!packed || g->pk_num_row_groups > 0, GVQA_E_INVALID, "rows_to_x4: the handle has no packed row groups");
    hipLaunchKernelGGL(k_rows_to_x4, dim3((unsigned)(packed ? g->pk_num_row_groups : g->num_row_groups)), dim3(512), 0, stream,
                       packed ? g->pk_row_group_ptr : g->row_group_ptr, packed ? g->pk_node_graph : g->node_graph, D, (int)cdiv(D, 4), X, ld,
                       X4, gmax, Vn, a_node, packed ? g->pk_node_old : nullptr);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// (Round 4, first half: a stand-alone coefficient kernel for chunk-major rows, k_gat_alpha_x4, stood here -- fp32 node logits from the
//  group's chunks + the two softmax phases, 38 us per hop at config 3.  Its work moved into the hop kernel: node logits leave with the
//  rows (LGT / the layout pass), the softmax runs in the hop kernel's prologue (ALP) or between two hops of the one launch (SEQ).)

// ------------------------------------------------------------------------------------------------------------------------------

// LDS-DMA with the global address as SGPR base + 32-bit VGPR offset (no 64-bit address registers per lane): 2 x 16 bytes per lane,
// 1 KiB apart on both sides / 4 bytes per lane
#ifdef GVQA_HA_M0_SAVE       /* (A/B build switch: round 4's form, M0 saved and restored around every DMA set-up) */
__device__ __forceinline__ void ha_dma16_x2(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
#else
__device__ __forceinline__ void ha_dma16_x2(const void* sbase, unsigned voff, unsigned lds_dst) {
    // (M0 is declared clobbered instead of saved and restored around the pair: two SALU instructions less per DMA set-up -- nothing else
    //  in this kernel lives in M0, and the compiler re-materialises it where it needs it)
    asm volatile(
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:1024"
        :
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory", "m0");
}
#endif
// GVQA_HA_NT (A/B build switch; bits: 1 x-chunk DMAs, 2 skip-row loads, 4 row stores): the ROW traffic of the hop kernel -- x chunks in, skip rows in, result rows out: 16 + 8 MiB per XCD and
// hop, each line touched once or re-used only ~100 us later -- carries the non-temporal hint, so that it does not push the hop's 4 MiB of
// weight tiles (re-used by all 32 workgroups of an XCD) out of the XCD's 4 MiB L2.
#ifndef GVQA_HA_NT
#define GVQA_HA_NT 0
#endif
#if GVQA_HA_NT & 1
#define GVQA_HA_NT_STR " nt"
#else
#define GVQA_HA_NT_STR ""
#endif
#ifdef GVQA_HA_M0_SAVE
__device__ __forceinline__ void ha_dma4(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" GVQA_HA_NT_STR "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
#else
__device__ __forceinline__ void ha_dma4(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" GVQA_HA_NT_STR
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory", "m0");
}
#endif
typedef float ha_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ha_row_load(const float* p) {
#if GVQA_HA_NT & 2
    const ha_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ha_f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void ha_row_store(float* p, const float4& v) {
#if GVQA_HA_NT & 4
    __builtin_nontemporal_store(ha_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<ha_f32x4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

// a wave-uniform pointer the compiler has lost track of (re-pointed inside the hop loop) back into SGPRs: the DMA helpers above take
// their base as an "s" operand, which inline asm does not legalise
template <typename T>
__device__ __forceinline__ const T* ha_uniform(const T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int ha_wave_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// LDS map (bytes): weight ring 3 x 32 KiB | A' ring 2 x 8 KiB | x ring 4 x 2 KiB | CSR slice: src [ECAP] ints, alpha [ECAP][4] floats |
// per-row: 1 / (scale H), graph id, in-degree, row of graph_term, row of `out` | per-graph output maxima
constexpr unsigned HA_BSTAGE = 16 * 2048, HA_B0 = 0, HA_A0 = 3 * HA_BSTAGE, HA_X0 = HA_A0 + 2 * 8192, HA_SRC0 = HA_X0 + 4 * 2048,
                   HA_AL0 = HA_SRC0 + HA_ECAP * 4, HA_ROW0 = HA_AL0 + HA_ECAP * 16, HA_GM0 = HA_ROW0 + 5 * HA_ROWS * 4, HA_LDS = HA_GM0 + HA_ROWS * 4;
static_assert(HA_LDS <= 160 * 1024, "hopagg: LDS");
// One-launch form (SEQ): + the second set of per-graph maxima | the slice's COO edge ids.  BETWEEN two hops of a tile the idle rings
// hold the next hop's coefficient phase: weight-ring stage 2 = the next hop's folded attention vectors Vn [8][C] | edge halves of the
// logits, head-major [4][ECAP]; A' ring = partial node logits [WC][128][8] (summed in place into wave column 0's slots).  The
// priming DMAs of the next hop go to weight-ring stages 0 / 1 and the x ring: no overlap with these.
constexpr unsigned HA_CC0 = HA_LDS, HA_LDS1 = HA_CC0 + 4 * 512 * 4;       // + per-column constants of the epilogue [4][512]: inverse weight scale | bias | BN scale | BN shift
constexpr unsigned HA_GM1 = HA_LDS1, HA_EID0 = HA_GM1 + HA_ROWS * 4, HA_LDS_SEQ = HA_EID0 + HA_ECAP * 4;
constexpr unsigned HA_VN0 = HA_B0 + 2 * HA_BSTAGE, HA_ST0 = HA_VN0 + 16384;
static_assert(HA_LDS_SEQ <= 160 * 1024 && HA_ST0 + HA_ECAP * 16 <= HA_A0, "hopagg (one launch): LDS");

// WR x WC waves, RT row tiles x TN column tiles of 32 x 32 per wave: <2, 4, 2, 4> covers 512 columns (config 3: d = 512), <4, 2, 1, 5>
// 320 (the reference's real width, d = 300: ten column tiles, no empty MFMA columns beyond the 20 that pad 300 to 320).
// SEQ: the K hops of gat_seq as ONE launch.  A workgroup owns all output columns of its rows and its rows are whole graphs, so hop
// i + 1 of a row group needs nothing from another workgroup: rows (chunk-major, through HBM / L2: written by the epilogue, DMA'd
// back by the next hop), per-graph maxima and attention coefficients are group-local.  Between two hops the workgroup runs the
// coefficient phase itself (gat_skip.py:134-135,180-208): node logits a_node = h . [V_l | V_r]^T of the rows it has just finished
// (out of the accumulator registers, against Vn in LDS), then leaky-relu + segment softmax per (node, head) -- the lane that owns
// the producer item (node, head) of the main loop computes exactly the coefficients it will use.  Hop 0's coefficients are computed in
// the first hop's prologue from the node logits the layout pass left (as ALP does in the per-hop form).
// LGT (per-hop launches): the node logits of the NEXT hop, a_node = h . [V_l | V_r]^T of the rows this launch produces, leave with them
// (accumulated as the finished values leave the epilogue: 8 packed FMAs per float4 against the next hop's Vn in LDS) -- the next launch's
// prologue (ALP) turns them into coefficients; nothing passes over the rows (134 MB at config 3) for the logits.
// ALP (per-hop launches): the hop's attention coefficients are computed in this launch's PROLOGUE -- node logits left by the previous
// launch (a_node_in), edge halves gathered through the slice's COO edge ids, leaky-relu + segment softmax by the lane that owns the
// producer item (node, head) -- under the latency of the priming DMAs: no coefficient kernel, no alpha_csr round trip.
// CP > 1 (per-hop launches only): the output columns of a row group split over CP workgroups (blockIdx.y), WC x TN column tiles each -- a
// 256-graph shard of config 3 is 64 row groups for 256 CUs; as 64 x 4 workgroups of 128 x 128 every CU has one.  Each part forms the full
// A' (the producer's work is duplicated: ~46 VALU per lane and step beside 6 MFMAs per wave instead of 24) but streams only its own
// quarter of the weights; the next hop's node logits and the per-graph maxima leave as one set per part and are combined by the reader.
template <int WR, int WC, int RT, int TN, bool SEQ, bool LGT = false, bool ALP = !SEQ, int CP = 1>
__global__ __launch_bounds__(512, 2) void k_hopagg4(HopAggArgs a, HopAggSeq hs) {
    static_assert(SEQ || ALP, "hopagg: the per-hop form computes its coefficients in its prologue (the form fed by a coefficient kernel was removed with that kernel)");
    static_assert(!(SEQ && (LGT || ALP)), "hopagg: the one-launch form computes its logits and coefficients itself (hop 0's in its first prologue, as ALP does)");
    static_assert(WR * WC == 8 && WR * RT == 4 && WC * TN <= 16, "hopagg: eight waves over 128 rows and at most 16 column tiles");
    static_assert(CP == 1 || (!SEQ && WC * TN <= 8), "hopagg: column parts are a per-hop form of at most 8 column tiles per part");
    constexpr int NTP = WC * TN;                      // column tiles of this workgroup
    constexpr int NBU = CP > 1 ? 1 : 2;               // weight DMA units per wave and step (a unit = one column tile, both pieces)
    const int cpart = CP > 1 ? (int)blockIdx.y : 0, ct0 = cpart * NTP, cbase = ct0 * 32;
    // Rings.  One workgroup per row group: weight stages of 16 tiles (32 KiB), three of them, DMAs two steps ahead; x chunks three ahead in
    // four slots.  Column parts: a step is 6 MFMAs per wave (~0.25 us), shorter than an L2 -> LDS DMA's trip (~0.5 us: with the rings above
    // the part measured 0.6 us per step, latency-bound) -- so weight stages of NTP tiles (8 KiB), SIX of them, DMAs FOUR steps ahead (only the
    // waves that own a tile issue them), x chunks five ahead in eight slots, both inside the first 64 KiB of the (otherwise idle) weight area.
    constexpr int PD = CP > 1 ? 4 : 2;                // weight DMAs run PD steps ahead
    constexpr int NBST = CP > 1 ? 6 : 3;              // weight ring stages
    constexpr unsigned BST = CP > 1 ? (unsigned)NTP * 2048u : HA_BSTAGE;      // bytes per weight stage
    constexpr int NXS = CP > 1 ? 8 : 4;               // x ring slots (x DMAs run PD + 1 steps ahead)
    constexpr unsigned XR0 = CP > 1 ? 48u * 1024u : HA_X0;
    static_assert(CP == 1 || (NBST * BST <= 48 * 1024 && XR0 + NXS * 2048 <= HA_VN0), "hopagg: the column part's rings sit below the coefficient phase's scratch");
    const bool b_owner = CP == 1 || __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) < NTP;       // this wave issues weight DMAs
    constexpr int H = 4;
    constexpr int NM = RT * TN;                       // MFMAs of one piece product per wave
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SEQ ? HA_LDS_SEQ : HA_LDS1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;
    const int t = blockIdx.x;
    const int ns = a.group_ptr[t], cnt = a.group_ptr[t + 1] - ns;
    const int e0 = a.rowptr[ns], ne = min(a.rowptr[ns + cnt] - e0, HA_ECAP);
    const int NQ = a.NQ, NCT = a.NCT;

    // ---- DMA duties of this wave: weight units (column tile, K step) = 2 KiB (both pieces), tiles wave and wave + 8; one
    // 256-byte slice of every x chunk
    // (every wave issues the same five DMA instructions in every K step -- column tiles past the last one re-load the last tile into
    //  their own, unused ring slot; steps past the last one re-load the last step -- so that the step is branch-free and the counted
    //  waits are the same everywhere)
    const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
    // operands of the current hop (SEQ: re-pointed at the top of every hop; wave-uniform: SGPRs)
    const uint16_t* Wk_h = a.Wk;
    const float *binv_h = a.binv, *epc_h = a.epc, *gterm_h = a.graph_term, *X4in_h = a.X4in;
    float *X4out_h = a.X4out, *out_h = a.out;
    int relu_h = a.relu;
    const uint16_t *wbase0 = nullptr, *wbase1 = nullptr;
    const float* xbase = nullptr;
    // (narrow layout, NTP = 10 column tiles: only waves 0 and 1 own a second tile -- the other six used to re-load tile 9 into unused ring
    //  slots to keep the counted waits uniform: 12 of 34 KiB per K step and CU of L2 -> LDS traffic for nothing.  Now a wave without a
    //  second tile issues three DMA instructions per step and waits on its own count)
    const bool b_second = NBU > 1 && (NTP >= 16 || __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + 8 < NTP);
    auto issue_b_unit = [&](int st, int u) {           // weight unit u of step st (clamped) -> ring slot st % NBST
        if (CP > 1 && !b_owner) return;                // (wave-uniform)
        if (u && !b_second) return;                    // (wave-uniform)
        ha_dma16_x2(ha_uniform((u ? wbase1 : wbase0) + (int64_t)min(st, NQ - 1) * 1024), lane16,
                    __builtin_amdgcn_readfirstlane(lds_base + HA_B0 + (unsigned)(st % NBST) * BST + (unsigned)(wave + 8 * u) * 2048u));
    };
    auto issue_b = [&](int st) { issue_b_unit(st, 0); if (NBU > 1) issue_b_unit(st, 1); };
    // ... and their forms for the K loop (round 6).  The general forms above cost every step ~25 scalar instructions of pure bookkeeping -- three 64-bit
    // pointer products with their clamps, two `st % 3` by multiply-high, three pairs of v_readfirstlane -- in a loop that PMC counters and the ISA put at
    // ~145 issued instructions per wave and step for 24 (wide) or 15 (narrow layout) MFMAs: the narrow layout's step IS its issue time.  In the loop the
    // bases stay put (SGPR pairs, set per hop), the step is a 32-bit byte offset per lane that advances by 2 KiB and saturates at the last step (two VALU),
    // and the ring slots are byte offsets that rotate by compare-and-select.
    // (round 6, late: the step's byte offset was first kept per lane -- lane part included -- in two VGPRs; with the register file full hipcc spilled exactly
    //  those two, re-loaded each in front of its DMA with `scratch_load_dword` + `s_waitcnt vmcnt(0)` -- its own counter knows nothing of the asm DMAs, so the
    //  wait drained every DMA in flight, twice per K step, the second time a few dozen instructions behind the step's weight DMAs.  Now the step offset is a
    //  scalar added to the base; the lane part is the constant lane16 / lane4)
    unsigned sb_off = 0, sx_off = 0;                  // scalar byte offsets of the NEXT weight step / x chunk to request
    unsigned bld_off = 0, bcur_off = 0;               // ring stage (byte offset) the next weight DMAs go to / the current step reads
    const uint16_t *wb0u = nullptr, *wb1u = nullptr;  // wbase0 / wbase1 / xbase as scalar-register pairs
    const float* xbu = nullptr;
    auto loop_state_init = [&]() {                     // after the priming DMAs of a hop: the loop's step 0 requests weight step PD and x chunk PD + 1
        wb0u = ha_uniform(wbase0); wb1u = ha_uniform(wbase1); xbu = ha_uniform(xbase);
        sb_off = (unsigned)min(PD, NQ - 1) * 2048u;
        sx_off = (unsigned)min(PD + 1, NQ - 1) * 2048u;
        bld_off = (unsigned)(PD % NBST) * BST;
        bcur_off = 0u;
    };
    const unsigned s_last = (unsigned)(NQ - 1) * 2048u;
    auto loop_issue_b = [&](int u) {
        if (CP > 1 && !b_owner) return;
        if (u && !b_second) return;
        ha_dma16_x2(reinterpret_cast<const char*>(u ? wb1u : wb0u) + sb_off, lane16, __builtin_amdgcn_readfirstlane(lds_base + HA_B0 + bld_off + (unsigned)(wave + 8 * u) * 2048u));
    };
    auto loop_issue_x = [&](int q) {
        ha_dma4(reinterpret_cast<const char*>(xbu) + sx_off, lane4, __builtin_amdgcn_readfirstlane(lds_base + XR0 + (unsigned)(q & (NXS - 1)) * 2048u + (unsigned)wave * 256u));
    };
    auto loop_advance = [&]() {                        // end of a step
        sb_off = min(sb_off + 2048u, s_last);
        sx_off = min(sx_off + 2048u, s_last);
        bld_off = bld_off == (unsigned)(NBST - 1) * BST ? 0u : bld_off + BST;
        bcur_off = bcur_off == (unsigned)(NBST - 1) * BST ? 0u : bcur_off + BST;
    };
    auto issue_x = [&](int q) {                        // this wave's 256 bytes of x chunk q (clamped) -> ring slot q & 3
        ha_dma4(ha_uniform(xbase + (int64_t)min(q, NQ - 1) * (HA_ROWS * 4)), lane4,
                __builtin_amdgcn_readfirstlane(lds_base + XR0 + (unsigned)(q & (NXS - 1)) * 2048u + (unsigned)wave * 256u));
    };
    // ---- this lane's producer item: node i = 16 wave + a, head h = 2 hhi + hlo; its first 8 in-edges in registers
    // (the four heads of a node are the four lanes of a quad: every lane fetches ONE of the node's source rows per batch of four
    //  edges and the quad shares them through DPP -- two row reads per lane and step instead of eight; the x gathers were a quarter
    //  of the step's LDS cycles)
    const int pa = lane >> 2, ph = lane & 3, hlo = ph & 1, hhi = ph >> 1;
    const int pi = wave * 16 + pa;
    const bool p_on = pi < cnt;
    int plo = 0, pdeg = 0;                            // (hop-invariant; loaded in the first hop's prologue, behind its priming DMAs)
    float al[HA_DMAX];                                // this head's coefficients of the node's first 8 in-edges (0 past the last)
    unsigned sep = 0u;                                // byte offsets (slot x 16) of the chunk rows this lane fetches: edge ph | edge 4 + ph << 16
    // acc += w * (x of quad lane E_): ONE instruction, v_fmac_f32 with a DPP quad broadcast on its first source (the compiler does not
    // fold its own v_mov_b32_dpp into the FMA here: 32 extra VALU operations per lane and K step, an eighth of the step's issue slots)
#if defined(__HIP_DEVICE_COMPILE__)
#define GVQA_HA_QF1(acc_, w_, x_, E_) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #E_ "," #E_ "," #E_ "," #E_ "] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc_) : "v"(x_), "v"(w_))
#else
#define GVQA_HA_QF1(acc_, w_, x_, E_) do { (acc_) += (w_) * (x_); } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define GVQA_HA_QM1(acc_, w_, x_, E_) asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[" #E_ "," #E_ "," #E_ "," #E_ "] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(acc_) : "v"(x_), "v"(w_))
#else
#define GVQA_HA_QM1(acc_, w_, x_, E_) do { (acc_) = (w_) * (x_); } while (0)
#endif
#define GVQA_HA_QMUL(acc_, w_, x_, E_) do { GVQA_HA_QM1((acc_).x, w_, (x_).x, E_); GVQA_HA_QM1((acc_).y, w_, (x_).y, E_); \
                                            GVQA_HA_QM1((acc_).z, w_, (x_).z, E_); GVQA_HA_QM1((acc_).w, w_, (x_).w, E_); } while (0)
#define GVQA_HA_QFMA(acc_, w_, x_, E_) do { GVQA_HA_QF1((acc_).x, w_, (x_).x, E_); GVQA_HA_QF1((acc_).y, w_, (x_).y, E_); \
                                            GVQA_HA_QF1((acc_).z, w_, (x_).z, E_); GVQA_HA_QF1((acc_).w, w_, (x_).w, E_); } while (0)
    int ovtrips = 0;                                  // wave-uniform trips through the LDS slice (edges past the HA_DMAX + HA_NOV a node keeps in registers)
    // Edges 9 .. 16 of a node ("overflow").  A K step walked them through the slice in LDS: per edge two DEPENDENT reads (source slot,
    // then the row) plus the coefficient -- ~140 cycles per edge and step for the whole wave (the trip count is the wave's largest
    // in-degree), 128 steps per tile.  At config 3 (in-degree 1 + Poisson(3), largest 15) three of four row groups have such a node:
    // phase stamps put 9 us per trip and tile on the main loop, 165 (no overflow) .. 229 us (7 trips) -- and the slowest workgroup is
    // what a launch waits for.  Now source offsets and coefficients of these edges sit in registers as well: one independent row
    // read + 4 FMAs per edge and step.
    int ovn = 0;                                      // wave-uniform: how many of them this wave uses
    float al_o[HA_NOV];
    unsigned so2[HA_NOV / 2];                         // their rows' byte offsets in a chunk (slot x 16), two per register
    float pscale = 1.f;
    int pg = 0;                                       // graph of this lane's node

    // the rows' power-of-two scale from their graph's largest input magnitude; per-row arrays of the epilogue
    auto set_row_scale = [&](float gmax_of_graph, bool first, int pgt_ = 0) {      // pgt_ (first call): the graph's row of graph_term (packed row groups: its id in the batch; otherwise pg)
        const int ex = split2h_exponent(gmax_of_graph);
        pscale = pow2i(ex);
        float* row_l = reinterpret_cast<float*>(smem + HA_ROW0);
        if (hlo == 0 && hhi == 0) {
            row_l[pi] = p_on ? pow2i(-ex) * (1.0f / H) : 0.f;
            if (first) {
                reinterpret_cast<int*>(row_l)[HA_ROWS + pi] = pg;
                reinterpret_cast<int*>(row_l)[2 * HA_ROWS + pi] = pdeg;
                reinterpret_cast<int*>(row_l)[3 * HA_ROWS + pi] = pgt_;
                reinterpret_cast<int*>(row_l)[4 * HA_ROWS + pi] = p_on ? (a.row_map ? a.row_map[ns + pi] : ns + pi) : 0;
            }
        }
    };
    const unsigned a_wr_off = (unsigned)((pi >> 5) * 2048 + ((pi & 31) + 32 * hhi) * 16 + hlo * 8);
    const int* src_l = reinterpret_cast<const int*>(smem + HA_SRC0);
    const float* al_l = reinterpret_cast<const float*>(smem + HA_AL0);
    // leaky-relu + segment softmax of this lane's (node pi, head ph) over its in-edges (gat_skip.py:183-190; the denominator's + 1e-16 as
    // in torch_geometric.utils.softmax) from LDS: node logits an_s [128][8] (a_l | a_r halves), this head's edge halves st[e] of the node's
    // in-edges (overwritten), the per-graph logit offset; the coefficients of the first 8 in-edges go to registers, further ones to the
    // slice in LDS
    auto coeffs_from_lds = [&](const float* an_s, float* st, float tlog, float slope_) {
        float* al_w = reinterpret_cast<float*>(smem + HA_AL0);
        const float ar = an_s[pi * 8 + H + ph] + tlog;
        // the first 8 in-edges in registers, branch-free (clamped slots, masked afterwards: all reads of a stage issued together);
        // edges beyond them (wave-uniform trip count) through the slice in LDS
        float lg[HA_DMAX];
        int srs[HA_DMAX];
#pragma unroll
        for (int e = 0; e < HA_DMAX; ++e) srs[e] = min(max(src_l[max(min(plo + e, ne - 1), 0)] - ns, 0), HA_ROWS - 1);
#pragma unroll
        for (int e = 0; e < HA_DMAX; ++e) lg[e] = st[min(e, max(pdeg - 1, 0))];
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < HA_DMAX; ++e) {
            float v = lg[e] + an_s[srs[e] * 8 + ph] + ar;
            v = v > 0.f ? v : v * slope_;
            lg[e] = e < pdeg ? v : -INFINITY;
            mx = fmaxf(mx, lg[e]);
        }
        for (int e = HA_DMAX; e < pdeg; ++e) {
            const int sr = min(max(src_l[plo + e] - ns, 0), HA_ROWS - 1);
            float v = st[e] + an_s[sr * 8 + ph] + ar;
            v = v > 0.f ? v : v * slope_;
            st[e] = v;
            mx = fmaxf(mx, v);
        }
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < HA_DMAX; ++e) {
            lg[e] = e < pdeg ? expf(lg[e] - mx) : 0.f;
            sum += lg[e];
        }
        for (int e = HA_DMAX; e < pdeg; ++e) {
            const float ex = expf(st[e] - mx);
            st[e] = ex;
            sum += ex;
        }
        const float den = sum + 1e-16f;
#pragma unroll
        for (int e = 0; e < HA_DMAX; ++e) al[e] = lg[e] / den;            // (0 past the node's last edge)
        for (int e = HA_DMAX; e < pdeg; ++e) al_w[(plo + e) * H + ph] = st[e] / den;     // (the loop reads the slice from a node's ninth edge on)
    };

    // A'(q) from x chunk q -> A' slot q & 1 (the first chunk, ahead of the loop; inside the loop the same operations are spread
    // between the MFMAs of a step)
    auto produce = [&](int q) {
        const float4* xs = reinterpret_cast<const float4*>(smem + XR0 + (q & (NXS - 1)) * 2048);
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        const float4 xr0 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (sep & 0xFFFFu));
        const float4 xr1 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (sep >> 16));
        GVQA_HA_QFMA(v0, al[0], xr0, 0); GVQA_HA_QFMA(v0, al[1], xr0, 1); GVQA_HA_QFMA(v0, al[2], xr0, 2); GVQA_HA_QFMA(v0, al[3], xr0, 3);
        GVQA_HA_QFMA(v1, al[4], xr1, 0); GVQA_HA_QFMA(v1, al[5], xr1, 1); GVQA_HA_QFMA(v1, al[6], xr1, 2); GVQA_HA_QFMA(v1, al[7], xr1, 3);
        if (ovn > 0) {
#pragma unroll
            for (int e = 0; e < HA_NOV; ++e)
                if (e < ovn) {
                    const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + ((so2[e >> 1] >> (16 * (e & 1))) & 0xFFFFu));
                    v1.x += al_o[e] * x.x; v1.y += al_o[e] * x.y; v1.z += al_o[e] * x.z; v1.w += al_o[e] * x.w;
                }
        }
        for (int e = 0; e < ovtrips; ++e) {
            const int k = HA_DMAX + HA_NOV + e;
            const int idx = max(min(plo + k, plo + pdeg - 1), 0);
            const int sr = min(max(src_l[idx] - ns, 0), HA_ROWS - 1);
            const float w = al_l[idx * H + ph];
            const float av = k < pdeg ? w : 0.f;
            const float4 x = xs[sr];
            v0.x += av * x.x; v0.y += av * x.y; v0.z += av * x.z; v0.w += av * x.w;
        }
        const float tx = (v0.x + v1.x) * pscale, ty = (v0.y + v1.y) * pscale, tz = (v0.z + v1.z) * pscale, tw = (v0.w + v1.w) * pscale;
        ha_f16x4 hi, lo;
        hi[0] = (_Float16)tx; hi[1] = (_Float16)ty; hi[2] = (_Float16)tz; hi[3] = (_Float16)tw;
        lo[0] = (_Float16)(tx - (float)hi[0]); lo[1] = (_Float16)(ty - (float)hi[1]);
        lo[2] = (_Float16)(tz - (float)hi[2]); lo[3] = (_Float16)(tw - (float)hi[3]);
        unsigned char* dst = smem + HA_A0 + (q & 1) * 8192 + a_wr_off;
        *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, hi);
        *reinterpret_cast<uint2*>(dst + 1024) = __builtin_bit_cast(uint2, lo);
    };

    const unsigned a_off = (unsigned)(wr * RT * 2048 + lane * 16);
    const unsigned b_off = (unsigned)(wc * TN * 2048 + lane * 16);
#ifdef GVQA_PROBES
#define GVQA_HA_DBG(bit_) (a.dbg & (bit_))
#else
#define GVQA_HA_DBG(bit_) false
#endif
    // One K step, laid out by hand.  A wave issues in order and an MFMA occupies the SIMD's matrix pipe for 32 cycles, so everything
    // else a step needs -- 5 DMA instructions, 12 fragment reads, the producer's 8 row reads, ~70 VALU operations and 2 LDS writes --
    // has to sit IN the gaps between this wave's MFMAs (about 4 issue slots each), not in front of or behind them.  History (us per
    // launch at config 3): DMAs at the top of the step, producer as a block between the piece products 468; branch-free step with
    // the DMAs between MFMA groups and the producer interleaved (sched_group_barrier) 452; fragment reads issued just ahead of the
    // product that needs them instead of 20 reads in front of the first MFMA 436; then the ROTATION below.
    //   All eight waves leave the step's barrier together and every one needs LDS data before its first MFMA: ~300 cycles of idle
    // pipe per step, 20 % of it.  So the third piece product of step s - 1 (a hi x b hi: registers only) is issued AFTER the barrier
    // that opens step s, under the reads of step s -- which costs a second register set for the b-hi fragments (16 registers; the
    // loop body is written for two steps with the sets swapped).
    ha_f16x8 afh[RT], afl[RT], bh0[TN], bh1[TN], bl[TN];      // a hi / a lo fragments, b hi (two sets), b lo
    // products n = lo .. hi - 1 of a piece product (n -> row tile n / TN, column tile n % TN)
#ifndef GVQA_HA_SNAKE
#define GVQA_HA_SNAKE 0       /* 1: odd row tiles walk their column tiles backwards, so that every two consecutive products share an operand: + 4 % for a bare MFMA
                                 stream (gvqa_mfma_stream), 0.2-0.5 % here -- inside the noise (`profiles/r06_snake_order_ab.jsonl`, same box, bit-identical results); off */
#endif
#define GVQA_HA_COL(n_) ((GVQA_HA_SNAKE && (((n_) / TN) & 1)) ? TN - 1 - (n_) % TN : (n_) % TN)
#define GVQA_HA_MFR(lo_, hi_, a_, b_) do { if (!GVQA_HA_DBG(4)) { _Pragma("unroll") for (int n_ = (lo_); n_ < (hi_); ++n_)                  \
        acc[n_ / TN][GVQA_HA_COL(n_)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_[GVQA_HA_COL(n_)], a_[n_ / TN], acc[n_ / TN][GVQA_HA_COL(n_)], 0, 0, 0); } } while (0)
    constexpr int NH = NM / 2;                        // the (a hi, b lo) product is issued in two parts around the x DMA
    // two values scaled by a power of two and split into fp16 pieces, packed two to a register: hi = f16(p v), lo = f16(p v - hi), each ONE
    // v_fma_mix instruction (the multiply by the scale rides in the FMA; hipcc's own lowering of the same expressions: v_mul + v_cvt_pk for
    // the packed hi AND a second v_fma_mixlo for the hi the subtraction reads -- 14 VALU per four values against 8).  Same bits: p v is
    // exact (p a power of two), and p v - hi is exact in fp32.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GVQA_HA_SPLIT_C)
#define GVQA_HA_SPLIT2(hi_, lo_, p_, a_, b_)                                                                                          \
    asm("v_fma_mixlo_f16 %0, %2, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %4, 0\n\t"                                                          \
        "v_fma_mixlo_f16 %1, %2, %3, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %2, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"       \
        : "=&v"(hi_), "=&v"(lo_) : "v"(p_), "v"(a_), "v"(b_))
#else           /* (host pass; GVQA_HA_SPLIT_C: A/B build with the compiler's lowering) */
#define GVQA_HA_SPLIT2(hi_, lo_, p_, a_, b_) do { const float ta_ = (a_) * (p_), tb_ = (b_) * (p_); ha_f16x2 h_, l_; h_[0] = (_Float16)ta_; h_[1] = (_Float16)tb_; \
        l_[0] = (_Float16)(ta_ - (float)h_[0]); l_[1] = (_Float16)(tb_ - (float)h_[1]); (hi_) = __builtin_bit_cast(unsigned, h_); (lo_) = __builtin_bit_cast(unsigned, l_); } while (0)
#endif
#define GVQA_HA_FMA4(acc_, w_, x_) do { acc_.x += (w_) * (x_).x; acc_.y += (w_) * (x_).y; acc_.z += (w_) * (x_).z; acc_.w += (w_) * (x_).w; } while (0)
    auto rd = [&](const unsigned char* p_) { return __builtin_bit_cast(ha_f16x8, *reinterpret_cast<const uint4*>(p_)); };
    // step s with its b-hi fragments read into BN_ while BO_ still feeds the previous step's last product
    // OV_ = 0: the wave has no node with more than 8 in-edges (five waves of six at config 3: in-degrees 1 + Poisson(3)) -- no overflow rows,
    // no zero-weight FMAs for them, and the step's first edge STARTS the sum (v_mul) instead of adding to a cleared accumulator: 20 of
    // ~114 VALU instructions per lane and step less (round 5; the ISA of the round-4 body: 64 DPP FMAs, 48 plain FMAs, 32 v_mov per two steps)
#define GVQA_HA_STEP(s_, BO_, BN_, OV_)                                                                                     \
    {                                                                                                                       \
        const int s = (s_);                                                                                                 \
        const float4* xs = reinterpret_cast<const float4*>(smem + XR0 + ((s + 1) & (NXS - 1)) * 2048);                      \
        float4 v0;                                                                                                          \
        if ((OV_) == 2) v0 = make_float4(0.f, 0.f, 0.f, 0.f);                                                               \
        /* a node's 9th and 10th in-edge (most overflow is one or two edges): their rows are only FETCHED here -- the FMAs sit  \
           further down, unconditional (zero rows, zero coefficients without overflow), so that the LDS round trip does not hold \
           back the wave's first MFMAs of the step; edges 11 .. 16 (rare) are still consumed here */                         \
        float4 xo0, xo1;                                                                                                    \
        if ((OV_) == 1) {       /* one or two overflow edges in the wave (the common overflow): both rows fetched unconditionally (slot 0, weight 0 when absent) */ \
            xo0 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (so2[0] & 0xFFFFu));        \
            xo1 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (so2[0] >> 16));            \
        }                                                                                                                   \
        if ((OV_) == 2) {                                                                                                   \
            xo0 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (so2[0] & 0xFFFFu));        \
            xo1 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (so2[0] >> 16));            \
            if (ovn > 2) {                                                                                                  \
                _Pragma("unroll") for (int e = 2; e < HA_NOV; ++e)                                                          \
                    if (e < ovn) {                                                                                          \
                        const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + ((so2[e >> 1] >> (16 * (e & 1))) & 0xFFFFu)); \
                        GVQA_HA_FMA4(v0, al_o[e], x);                                                                       \
                    }                                                                                                       \
            }                                                                                                               \
        }                                                                                                                   \
        if ((OV_) == 2 && ovtrips > 0) {                                                                                    \
            for (int e = 0; e < ovtrips; ++e) {                                                                             \
                const int k = HA_DMAX + HA_NOV + e;                                                                         \
                const int idx = max(min(plo + k, plo + pdeg - 1), 0);                                                       \
                const int sr = min(max(src_l[idx] - ns, 0), HA_ROWS - 1);                                                   \
                const float w = al_l[idx * H + ph];                                                                         \
                const float av = k < pdeg ? w : 0.f;                                                                        \
                const float4 x = xs[sr];                                                                                    \
                GVQA_HA_FMA4(v0, av, x);                                                                                    \
            }                                                                                                               \
        }                                                                                                                   \
        /* ---- from here to the barrier ONE basic block ---- */                                                            \
        const unsigned char* sa = smem + HA_A0 + (s & 1) * 8192 + a_off;                                                    \
        const unsigned char* sb = smem + HA_B0 + bcur_off + b_off;                                                          \
        float4 xr;                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < RT; ++i) afl[i] = rd(sa + i * 2048 + 1024);                                   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) BN_[j] = rd(sb + j * 2048);                                          \
        /* (a hi, b hi) of the PREVIOUS step: registers only, under the reads above */                                      \
        if (s > 0) GVQA_HA_MFR(0, NM, afh, BO_);                                                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, RT + TN, 0);                                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (!GVQA_HA_DBG(2)) loop_issue_b(0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        /* (a lo, b hi); a hi, b lo and the first four x rows travel under it */                                            \
        GVQA_HA_MFR(0, NM, afl, BN_);                                                                                       \
        _Pragma("unroll") for (int i = 0; i < RT; ++i) afh[i] = rd(sa + i * 2048);                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) bl[j] = rd(sb + j * 2048 + 1024);                                    \
        xr = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (sep & 0xFFFFu));                \
        _Pragma("unroll") for (int z = 0; z < (RT + TN + 2) / 2; ++z) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); } \
        __builtin_amdgcn_sched_group_barrier(0x008, NM - (RT + TN + 2) / 2, 0);                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (NBU > 1 && !GVQA_HA_DBG(2)) loop_issue_b(1);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        /* (a hi, b lo) with the producer: 8 x (4 FMAs), the second four x rows re-using xa */                              \
        GVQA_HA_MFR(0, NH, afh, bl);                                                                                        \
        if (!GVQA_HA_DBG(1)) { if ((OV_) == 2) GVQA_HA_QFMA(v0, al[0], xr, 0); else GVQA_HA_QMUL(v0, al[0], xr, 0);           \
                               GVQA_HA_QFMA(v0, al[1], xr, 1); GVQA_HA_QFMA(v0, al[2], xr, 2); GVQA_HA_QFMA(v0, al[3], xr, 3); \
                               if (OV_) { GVQA_HA_FMA4(v0, al_o[0], xo0); GVQA_HA_FMA4(v0, al_o[1], xo1); } }                 \
        else if ((OV_) != 2) v0 = make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
        _Pragma("unroll") for (int z = 0; z < NH; ++z) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, (40 + NH - 1) / NH, 0); } \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        loop_issue_x(s + PD + 1);                                                                                           \
        xr = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (sep >> 16));                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        GVQA_HA_MFR(NH, NM, afh, bl);                                                                                       \
        if (!GVQA_HA_DBG(1)) {                                                                                              \
            GVQA_HA_QFMA(v0, al[4], xr, 0); GVQA_HA_QFMA(v0, al[5], xr, 1); GVQA_HA_QFMA(v0, al[6], xr, 2); GVQA_HA_QFMA(v0, al[7], xr, 3); \
            /* scale by the graph's power of two, split into two fp16 pieces, 2 x 8 bytes into the A-fragment image of step s + 1 */ \
            const float psc = s + 1 < NQ ? pscale : 0.f;                                                                    \
            uint2 hi, lo;                                                                                                   \
            GVQA_HA_SPLIT2(hi.x, lo.x, psc, v0.x, v0.y); GVQA_HA_SPLIT2(hi.y, lo.y, psc, v0.z, v0.w);                       \
            unsigned char* dst = smem + HA_A0 + ((s + 1) & 1) * 8192 + a_wr_off;      /* (last step: a free slot, never read) */ \
            *reinterpret_cast<uint2*>(dst) = hi;                                                                            \
            *reinterpret_cast<uint2*>(dst + 1024) = lo;                                                                     \
        }                                                                                                                   \
        _Pragma("unroll") for (int z = 0; z < NM - NH; ++z) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, (52 + NM - NH - 1) / (NM - NH), 0); } \
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                                                                  \
        loop_advance();                                                                                                     \
        /* this wave's DMAs of step s + 1 (issued one step ago) have landed, its A' writes are out; then everybody's */     \
        if (GVQA_HA_DBG(64)) {            /* (measurement: the barrier without the counted DMA wait / 128: the waits without the barrier) */ \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
            __builtin_amdgcn_s_barrier();                                                                                   \
        } else if (!GVQA_HA_DBG(16)) {                                                                                      \
            if (CP == 1 && (NTP >= 16 || b_second)) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");   /* (this step's five DMA instructions may stay in flight) */ \
            else if (CP == 1) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");   /* (a wave with one weight tile: three) */ \
            else if (b_owner) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");   /* (column parts: the last PD - 1 = 3 steps' DMAs -- 3 per step, */ \
            else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                /*  or the x DMA alone for a wave without a weight tile) */   \
            if (!GVQA_HA_DBG(128)) __builtin_amdgcn_s_barrier();                                                            \
        }                                                                                                                   \
    }
    // ---- ping-pong form of the step (GVQA_HA_PP; round 6, late) ------------------------------------------------------------------------
    // The step above costs 1.43 us at config 3 for 0.64-0.81 us of MFMA time per SIMD, and `r06_cfg3_sync_parts_stamps.jsonl` says where the rest
    // goes: without the barrier that closes it the main loop runs 27 % shorter (without the counted DMA wait: 1-4 %).  Eight waves that leave a
    // barrier together run the same phase at the same time -- everybody reads LDS, then everybody multiplies -- so the LDS pipe and the matrix pipe
    // take turns idling.  Here the two waves of a SIMD (w and w + 4: row halves 0 and 1, each of which produces exactly the A' rows it consumes)
    // run HALF A STEP APART: while one is in its X phase (all 12 fragment reads of its step into registers, the producer's gather / FMAs / split /
    // A' writes for the next step, the five DMAs), the other is in its Y phase (the step's 24 MFMAs back to back, raised priority); two barriers
    // per step keep them there.  Row half 1 enters the loop one barrier late and row half 0 leaves it one barrier late.  Hazards: a weight stage
    // and an A' image are only read in X (into registers), the stage refilled in X(s) -- step s + 2's, slot (s - 1) % 3 -- was last read in
    // X(s - 1) by the lagging half, one barrier earlier; a wave's own DMAs of step s + 1 (issued in X(s - 1)) are waited for at the end of X(s),
    // two barriers before the leading half reads them.
    // MEASURED (`profiles/r06_pingpong_step_ab.jsonl`, same box, results bit-identical to the shipped step's): 384-390 us per hop against 371-377 -- 2-4 % SLOWER,
    // whatever the Y phase's priority.  What the barrier costs is not that the waves run the same phase together but that a step then takes as long as its
    // slowest wave, 128 times per hop (a sum of maxima, not a maximum of sums); two barriers per step pay that twice.  Kept as a build switch, off.
#ifndef GVQA_HA_PP_PRIO
#define GVQA_HA_PP_PRIO 3
#endif
#define GVQA_HA_STEP_PP(s_, OV_)                                                                                            \
    {                                                                                                                       \
        const int s = (s_);                                                                                                 \
        const float4* xs = reinterpret_cast<const float4*>(smem + XR0 + ((s + 1) & (NXS - 1)) * 2048);                      \
        float4 v0;                                                                                                          \
        if ((OV_) == 2) v0 = make_float4(0.f, 0.f, 0.f, 0.f);                                                               \
        float4 xo0, xo1;                                                                                                    \
        if ((OV_) == 1) {                                                                                                   \
            xo0 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (so2[0] & 0xFFFFu));        \
            xo1 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (so2[0] >> 16));            \
        }                                                                                                                   \
        if ((OV_) == 2) {                                                                                                   \
            xo0 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (so2[0] & 0xFFFFu));        \
            xo1 = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (so2[0] >> 16));            \
            if (ovn > 2) {                                                                                                  \
                _Pragma("unroll") for (int e = 2; e < HA_NOV; ++e)                                                          \
                    if (e < ovn) {                                                                                          \
                        const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + ((so2[e >> 1] >> (16 * (e & 1))) & 0xFFFFu)); \
                        GVQA_HA_FMA4(v0, al_o[e], x);                                                                       \
                    }                                                                                                       \
            }                                                                                                               \
        }                                                                                                                   \
        if ((OV_) == 2 && ovtrips > 0) {                                                                                    \
            for (int e = 0; e < ovtrips; ++e) {                                                                             \
                const int k = HA_DMAX + HA_NOV + e;                                                                         \
                const int idx = max(min(plo + k, plo + pdeg - 1), 0);                                                       \
                const int sr = min(max(src_l[idx] - ns, 0), HA_ROWS - 1);                                                   \
                const float w = al_l[idx * H + ph];                                                                         \
                const float av = k < pdeg ? w : 0.f;                                                                        \
                const float4 x = xs[sr];                                                                                    \
                GVQA_HA_FMA4(v0, av, x);                                                                                    \
            }                                                                                                               \
        }                                                                                                                   \
        /* ---- X: the step's fragments, the next step's A', the DMAs ---- */                                               \
        const unsigned char* sa = smem + HA_A0 + (s & 1) * 8192 + a_off;                                                    \
        const unsigned char* sb = smem + HA_B0 + bcur_off + b_off;                                                          \
        float4 xr = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (sep & 0xFFFFu));         \
        _Pragma("unroll") for (int i = 0; i < RT; ++i) { afl[i] = rd(sa + i * 2048 + 1024); afh[i] = rd(sa + i * 2048); }   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) { bh0[j] = rd(sb + j * 2048); bl[j] = rd(sb + j * 2048 + 1024); }     \
        if (!GVQA_HA_DBG(2)) { loop_issue_b(0); if (NBU > 1) loop_issue_b(1); }                                             \
        if (!GVQA_HA_DBG(1)) { if ((OV_) == 2) GVQA_HA_QFMA(v0, al[0], xr, 0); else GVQA_HA_QMUL(v0, al[0], xr, 0);           \
                               GVQA_HA_QFMA(v0, al[1], xr, 1); GVQA_HA_QFMA(v0, al[2], xr, 2); GVQA_HA_QFMA(v0, al[3], xr, 3); \
                               if (OV_) { GVQA_HA_FMA4(v0, al_o[0], xo0); GVQA_HA_FMA4(v0, al_o[1], xo1); } }                 \
        else if ((OV_) != 2) v0 = make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
        loop_issue_x(s + PD + 1);                                                                                           \
        xr = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(xs) + (sep >> 16));                    \
        if (!GVQA_HA_DBG(1)) {                                                                                              \
            GVQA_HA_QFMA(v0, al[4], xr, 0); GVQA_HA_QFMA(v0, al[5], xr, 1); GVQA_HA_QFMA(v0, al[6], xr, 2); GVQA_HA_QFMA(v0, al[7], xr, 3); \
            const float psc = s + 1 < NQ ? pscale : 0.f;                                                                    \
            uint2 hi, lo;                                                                                                   \
            GVQA_HA_SPLIT2(hi.x, lo.x, psc, v0.x, v0.y); GVQA_HA_SPLIT2(hi.y, lo.y, psc, v0.z, v0.w);                       \
            unsigned char* dst = smem + HA_A0 + ((s + 1) & 1) * 8192 + a_wr_off;                                            \
            *reinterpret_cast<uint2*>(dst) = hi;                                                                            \
            *reinterpret_cast<uint2*>(dst + 1024) = lo;                                                                     \
        }                                                                                                                   \
        loop_advance();                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);            /* (the MFMAs below only read registers: nothing else keeps them behind the barrier) */ \
        if (NTP >= 16 || b_second) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");                              \
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                                                    \
        __builtin_amdgcn_s_barrier();                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        /* ---- Y: the step's products, nothing else ---- */                                                                \
        if (GVQA_HA_PP_PRIO) __builtin_amdgcn_s_setprio(GVQA_HA_PP_PRIO);                                                   \
        GVQA_HA_MFR(0, NM, afl, bh0);                                                                                       \
        GVQA_HA_MFR(0, NM, afh, bl);                                                                                        \
        GVQA_HA_MFR(0, NM, afh, bh0);                                                                                       \
        if (GVQA_HA_PP_PRIO) __builtin_amdgcn_s_setprio(0);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
    }
    // (an odd number of steps runs one more with an all-zero A operand -- the producer's scale is 0 past the last chunk, the weight
    //  DMA re-loads the last step's tiles -- so that the two-step body needs no tail variant)
    f32x16 acc[RT][TN];
    const int nhops = SEQ ? hs.K : 1;
#ifdef GVQA_PROBES
#define GVQA_HA_STAMP(hop_, k_) do { if (SEQ && hs.stamps && tid == 0) hs.stamps[((int64_t)t * hs.K + (hop_)) * 8 + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GVQA_HA_STAMP(hop_, k_) do { } while (0)
#endif
    // (Tried, one launch: workgroups of the first dispatch round started up to 8 x 5 .. 25 us apart, so that the CUs' epilogues -- 0.5 MB
    //  of row reads and writes per tile, all CUs within the same ~35 us of a hop -- spread over the hop period: 2.34 / 2.32 / 2.30 /
    //  2.50 ms per launch against 2.26-2.30 without, measurement build.  The epilogues are not limited by each other.)
    for (int hop = 0; hop < nhops; ++hop) {
    GVQA_HA_STAMP(hop, 0);                            // hop begins
    [[maybe_unused]] const bool more = hop + 1 < nhops;           // (SEQ) another hop follows: rows leave chunk-major, coefficient phase behind the epilogue
    if constexpr (SEQ) {
        const char* wk_ = reinterpret_cast<const char*>(hs.Wk) + hop * hs.w_hop_bytes;
        Wk_h = reinterpret_cast<const uint16_t*>(wk_);
        binv_h = reinterpret_cast<const float*>(wk_ + hs.binv_off_bytes);
        epc_h = hs.epc + hop * hs.epc_hop;
        gterm_h = hs.graph_term ? hs.graph_term + hop * hs.t_hop : nullptr;
        relu_h = (int)((hs.relu_mask >> hop) & 1u);
        X4in_h = (hop & 1) ? hs.X4b : hs.X4a;
        X4out_h = more ? ((hop & 1) ? hs.X4a : hs.X4b) : nullptr;
        out_h = more ? nullptr : a.out;
    }
    wbase0 = Wk_h + (int64_t)min(CP > 1 ? ct0 + wave % NTP : wave, NCT - 1) * NQ * 1024;            // (wave-uniform: SGPRs)
    wbase1 = Wk_h + (int64_t)min(wave + 8, NCT - 1) * NQ * 1024;
    xbase = X4in_h + (int64_t)t * NQ * (HA_ROWS * 4) + wave * 64;
    {   // the epilogue's per-column constants -> LDS [4][512] (one 1 KiB unit per wave; lanes past C re-read the row's last 16 bytes into the padding)
        const int npr = (a.C + 255) >> 8;             // units per row
        if (wave < 4 * npr) {
            const int k = wave / npr, hf = wave - k * npr;
            const float* rowp = k == 0 ? binv_h : epc_h + (int64_t)(k - 1) * a.epc_ld;
            lds_dma16_b(rowp + min(hf * 256 + lane * 4, a.C - 4), __builtin_amdgcn_readfirstlane(lds_base + HA_CC0 + (unsigned)(k * 512 + hf * 256) * 4u));
        }
    }
    if (!SEQ || hop == 0) {            // (SEQ, later hops: the previous hop's coefficient phase primed the weight ring and wrote x chunks 0 .. 2 into the x ring)
#pragma unroll
        for (int q = 0; q < PD; ++q) issue_b(q);
#pragma unroll
        for (int q = 0; q < PD + 1; ++q) issue_x(q);
    }
    [[maybe_unused]] float tlog0 = 0.f;
    // (the one-launch form computes hop 0's coefficients the same way, in the prologue of its first hop: its operands ride in `hs`)
    [[maybe_unused]] const float* ae0_ = SEQ ? hs.a_edge : a.a_edge;
    [[maybe_unused]] const int32_t* eid0_ = SEQ ? hs.csr_eid : a.csr_eid;
    [[maybe_unused]] const int64_t aes0_ = SEQ ? hs.a_edge_stride : a.a_edge_stride;
    if ((ALP && !SEQ) || (SEQ && hop == 0)) {
        // everything the coefficients need starts its trip from HBM now: the slice's sources (LDS), this group's node logits (LDS:
        // weight-ring stage 2, idle until step 0), the edge halves of the logits gathered through the COO edge ids (LDS, head-major)
        const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
        for (int u = wbase; u < ne; u += 512)
            lds_dma4_b(a.csr_src + e0 + min(u + lane, ne - 1), __builtin_amdgcn_readfirstlane(lds_base + HA_SRC0 + (unsigned)u * 4u));
        const int nan = cnt * 8;
        const int nparts = SEQ ? 1 : max(a.parts_in, 1);              // (sets of node logits / maxima left by a column-split launch)
        for (int pp = 0; pp < nparts; ++pp)
            for (int u = wave * 256; u < nan; u += 2048)
                lds_dma16_b(a.a_node_in + (int64_t)pp * a.an_part_stride + (int64_t)ns * 8 + min(u + lane * 4, nan - 4),
                            __builtin_amdgcn_readfirstlane(lds_base + HA_VN0 + (unsigned)(pp * 1024 + u) * 4u));
        for (int u = wbase; u < ne; u += 512) {
            const int eid = eid0_[e0 + min(u + lane, ne - 1)];
            const float* src = ae0_ + (int64_t)eid * aes0_;
#pragma unroll
            for (int h = 0; h < H; ++h)
                lds_dma4_b(src + h, __builtin_amdgcn_readfirstlane(lds_base + HA_ST0 + (unsigned)(h * HA_ECAP + u) * 4u));
            if constexpr (SEQ) reinterpret_cast<int*>(smem + HA_EID0)[min(u + lane, ne - 1)] = eid;     // (the later hops gather through the ids in LDS)
        }
        plo = p_on ? a.rowptr[ns + pi] - e0 : 0;
        pdeg = p_on ? min(a.rowptr[ns + pi + 1] - e0, HA_ECAP) - plo : 0;
        ovn = min(max(ha_wave_max(pdeg) - HA_DMAX, 0), HA_NOV);
        ovtrips = max(ha_wave_max(pdeg) - HA_DMAX - HA_NOV, 0);
        pg = a.node_graph[ns + min(pi, cnt - 1)];
        const int pgt = a.graph_old ? a.graph_old[pg] : pg;
        if (gterm_h && p_on) tlog0 = gterm_h[(int64_t)pgt * a.t_ld + a.C + ph];
        float gmx = a.gmax_in[pg];
        for (int pp = 1; pp < nparts; ++pp) gmx = fmaxf(gmx, a.gmax_in[(int64_t)pp * a.gm_part_stride + pg]);
        set_row_scale(gmx, true, pgt);
        if (tid < HA_ROWS) reinterpret_cast<unsigned*>(smem + HA_GM0)[tid] = 0u;
        if constexpr (SEQ) { if (tid < HA_ROWS) reinterpret_cast<unsigned*>(smem + HA_GM1)[tid] = 0u; }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // rings' first stages, CSR slice and per-row arrays are in place
    if constexpr (!SEQ) {
        if (a.parts_in > 1) {                         // (block-uniform) the parts' node-logit sets summed in place, in a fixed order
            float* an = reinterpret_cast<float*>(smem + HA_VN0);
            for (int it = tid; it < cnt * 8; it += 512) {
                float sv = an[it];
                for (int pp = 1; pp < a.parts_in; ++pp) sv += an[pp * 1024 + it];
                an[it] = sv;
            }
            __syncthreads();
        }
    }
    if ((ALP && !SEQ) || (SEQ && hop == 0)) {
        coeffs_from_lds(reinterpret_cast<const float*>(smem + HA_VN0), reinterpret_cast<float*>(smem + HA_ST0) + ph * HA_ECAP + plo, tlog0, SEQ ? hs.slope : a.slope);
        const int i0 = max(min(plo + ph, ne - 1), 0), i1 = max(min(plo + 4 + ph, ne - 1), 0);
        const unsigned s0 = ne > 0 ? (unsigned)min(max(src_l[i0] - ns, 0), HA_ROWS - 1) * 16u : 0u;
        const unsigned s1 = ne > 0 ? (unsigned)min(max(src_l[i1] - ns, 0), HA_ROWS - 1) * 16u : 0u;
        sep = s0 | (s1 << 16);
        if (!SEQ && a.alpha_out && cpart == 0) {      // (block-uniform) the attention weights are asked for: COO order, [E, H]
            const float* al_w = reinterpret_cast<const float*>(smem + HA_AL0);
#pragma unroll
            for (int e = 0; e < HA_DMAX; ++e)
                if (e < pdeg) a.alpha_out[(int64_t)a.csr_eid[e0 + plo + e] * H + ph] = al[e];
            for (int e = HA_DMAX; e < pdeg; ++e) a.alpha_out[(int64_t)a.csr_eid[e0 + plo + e] * H + ph] = al_w[(plo + e) * H + ph];
        }
    }
    if (ovn > 0) {                                    // (wave-uniform) the overflow edges' coefficients and row offsets out of the slice in LDS
#pragma unroll
        for (int e = 0; e < HA_NOV; ++e) {
            const int k = HA_DMAX + e;
            const int idx = max(min(plo + k, ne - 1), 0);
            const float w = al_l[idx * H + ph];
            al_o[e] = k < pdeg ? w : 0.f;
            const unsigned off = (unsigned)min(max(src_l[idx] - ns, 0), HA_ROWS - 1) * 16u;
            if (e & 1) so2[e >> 1] |= off << 16;
            else so2[e >> 1] = off;
        }
    } else {                                          // (defined on every path: coefficients 0 and 1 are used unconditionally in the K step; SEQ: otherwise they stay live -- and spill -- across the previous hop's epilogue)
#pragma unroll
        for (int e = 0; e < HA_NOV; ++e) al_o[e] = 0.f;
#pragma unroll
        for (int e = 0; e < HA_NOV / 2; ++e) so2[e] = 0u;
    }
    produce(0);
    loop_state_init();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    GVQA_HA_STAMP(hop, 1);                            // rings primed, first A' chunk produced

#ifndef GVQA_HA_OV_ALWAYS
#define GVQA_HA_OV_ALWAYS 0                           // (A/B build switch: 1 = every wave takes the overflow-capable body, as in round 4)
#endif
#ifndef GVQA_HA_PP
#define GVQA_HA_PP 0                                  // 1: the ping-pong form of the step (one workgroup per row group only)
#endif
    constexpr bool PP = GVQA_HA_PP != 0 && CP == 1;
    if constexpr (PP) {
        const int NQe = (NQ + 1) & ~1;
        if (wave >= 4) __builtin_amdgcn_s_barrier();                      // row half 1 runs half a step behind row half 0
        if (GVQA_HA_OV_ALWAYS || ovn > 2 || ovtrips > 0) {
            for (int sq = 0; sq < NQe; ++sq) GVQA_HA_STEP_PP(sq, 2)
        } else if (ovn > 0) {
            for (int sq = 0; sq < NQe; ++sq) GVQA_HA_STEP_PP(sq, 1)
        } else {
            for (int sq = 0; sq < NQe; ++sq) GVQA_HA_STEP_PP(sq, 0)
        }
        if (wave < 4) __builtin_amdgcn_s_barrier();
    } else {
    if (GVQA_HA_OV_ALWAYS || ovn > 2 || ovtrips > 0) {                    // (wave-uniform; all three bodies meet the same barriers)
        for (int sq = 0; sq < NQ; sq += 2) {
            GVQA_HA_STEP(sq, bh1, bh0, 2)
            GVQA_HA_STEP(sq + 1, bh0, bh1, 2)
        }
    } else if (ovn > 0) {
        for (int sq = 0; sq < NQ; sq += 2) {
            GVQA_HA_STEP(sq, bh1, bh0, 1)
            GVQA_HA_STEP(sq + 1, bh0, bh1, 1)
        }
    } else {
        for (int sq = 0; sq < NQ; sq += 2) {
            GVQA_HA_STEP(sq, bh1, bh0, 0)
            GVQA_HA_STEP(sq + 1, bh0, bh1, 0)
        }
    }
    GVQA_HA_MFR(0, NM, afh, bh1);                    // the last step's (a hi, b hi)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // (the clamped re-loads of the last steps)
    GVQA_HA_STAMP(hop, 2);                            // main loop done
#ifdef GVQA_PROBES
    if (SEQ && hs.stamps && tid == 0 && hop + 1 == nhops)      // (the last hop uses stamps 0 .. 3 only: slot 7 = where the workgroup ran)
        hs.stamps[((int64_t)t * hs.K + hop) * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 20) << 32) | __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
#endif

#ifdef GVQA_PROBES
    if (a.dbg & 32) {                                 // (measurement: no epilogue; the accumulators kept live)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)          /* (host pass: a "v" constraint on a value of dependent type silently voids the kernel's stub) */
                asm volatile("" ::"v"(acc[i][j]));
#endif
            }
        return;
    }
#endif
    // ---- epilogue, register -> global: lane (m, hh) owns columns 8 q + 4 hh + 0..3 of row m of tile (i, j)
    int m = lane & 31, hh = lane >> 5;
#if defined(__HIP_DEVICE_COMPILE__)
    // (SEQ: opaque per hop -- otherwise the epilogue's 64-bit row / column offsets, invariant across hops, are all hoisted out of the
    //  hop loop and spilled: 690 bytes of scratch per lane)
    if constexpr (SEQ) asm volatile("" : "+v"(m), "+v"(hh));
#endif
    const float* row_l = reinterpret_cast<const float*>(smem + HA_ROW0);
    unsigned* gm_l = reinterpret_cast<unsigned*>(smem + ((SEQ && (hop & 1)) ? HA_GM1 : HA_GM0));
    const int gf = a.node_graph[ns];
    const int C = a.C;
    const int CQ = C >> 2;                            // chunks of the output rows (C % 4 == 0)
    [[maybe_unused]] float tlog = 0.f;                // (SEQ) the next hop's per-graph logit offset of this lane's (node, head)
    if constexpr (LGT) {
        __syncthreads();                              // every wave's ring DMAs have landed: weight-ring stage 2 takes the next hop's Vn
        const int nvn = 8 * C;
        for (int u = wave * 256; u < nvn; u += 2048)
            lds_dma16_b(a.Vn_next + min(u + lane * 4, nvn - 4), __builtin_amdgcn_readfirstlane(lds_base + HA_VN0 + (unsigned)u * 4u));
    }
    if constexpr (SEQ) {
        if (more) {                                   // (block-uniform)
            // every wave's ring DMAs have landed: weight-ring stage 2 takes the next hop's folded attention vectors and the edge
            // halves of its logits (gathered through the slice's COO edge ids, head-major), stages 0 / 1 the next hop's first two
            // weight steps -- all on their way during the epilogue's loads below
            __syncthreads();
            const float* vn_next = hs.Vn + (int64_t)(hop + 1) * 8 * C;
            const float* gterm_next = hs.graph_term ? hs.graph_term + (hop + 1) * hs.t_hop : nullptr;
            const int nvn = 8 * C;                    // floats of Vn [2 H][C] (C == Dn)
            for (int u = wave * 256; u < nvn; u += 2048)
                lds_dma16_b(vn_next + min(u + lane * 4, nvn - 4), __builtin_amdgcn_readfirstlane(lds_base + HA_VN0 + (unsigned)u * 4u));
            const int* eid_l = reinterpret_cast<const int*>(smem + HA_EID0);
            const float* ae = hs.a_edge + (int64_t)(hop + 1) * H;
            for (int u = wave * 64; u < ne; u += 512) {
                const float* src = ae + (int64_t)eid_l[min(u + lane, ne - 1)] * hs.a_edge_stride;
#pragma unroll
                for (int h = 0; h < H; ++h)
                    lds_dma4_b(src + h, __builtin_amdgcn_readfirstlane(lds_base + HA_ST0 + (unsigned)(h * HA_ECAP + u) * 4u));
            }
            if (gterm_next && p_on) tlog = gterm_next[(int64_t)reinterpret_cast<const int*>(smem + HA_ROW0)[3 * HA_ROWS + pi] * a.t_ld + C + ph];     // (the graph's row of graph_term: out of LDS, not carried across the hop)
            const uint16_t* wkn = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(hs.Wk) + (hop + 1) * hs.w_hop_bytes);
            wbase0 = wkn + (int64_t)min(wave, NCT - 1) * NQ * 1024;
            wbase1 = wkn + (int64_t)min(wave + 8, NCT - 1) * NQ * 1024;
            issue_b(0);
            issue_b(1);
        }
    }
    GVQA_HA_STAMP(hop, 3);                            // coefficient-phase DMAs and the next hop's weight priming issued
    // node logits a_node[r, jj] = sum_c h[r, c] Vn[jj, c], jj < 2 H (a_l | a_r halves, gat_skip.py:134-135) of the finished rows in
    // the accumulator registers against Vn in LDS: this lane's columns (packed fp32 FMAs: two columns per instruction, even and odd
    // columns in separate chains), the wave's partial sums -> part[wc][row][0..8) in the idle A' ring
    auto node_logits_to_part = [&]() {
            const float* vn_l = reinterpret_cast<const float*>(smem + HA_VN0);
            ha_f32x2 an2[RT][8];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) an2[i][jj] = ha_f32x2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = cbase + (wc * TN + j) * 32 + 8 * q + 4 * hh;
                    if (c0 >= C) continue;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const float4 vn = *reinterpret_cast<const float4*>(vn_l + jj * C + c0);
                        const ha_f32x2 v01{vn.x, vn.y}, v23{vn.z, vn.w};
#pragma unroll
                        for (int i = 0; i < RT; ++i) {
                            const ha_f32x2 a01{acc[i][j][4 * q], acc[i][j][4 * q + 1]}, a23{acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                            an2[i][jj] = a01 * v01 + an2[i][jj];
                            an2[i][jj] = a23 * v23 + an2[i][jj];
                        }
                    }
                }
            GVQA_HA_STAMP(hop, 5);                            // rows stored, node-logit FMAs done
            // the two column halves of a row (lanes m and m + 32) meet; lane m leaves the wave's partial at part[wc][r][0..8)
            float* part = reinterpret_cast<float*>(smem + HA_A0);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                float an[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    an[jj] = an2[i][jj][0] + an2[i][jj][1];
                    an[jj] += __shfl_xor(an[jj], 32, 64);
                }
                if (hh == 0) {
                    float* dst = part + ((wc * HA_ROWS) + (wr * RT + i) * 32 + m) * 8;
                    *reinterpret_cast<float4*>(dst) = make_float4(an[0], an[1], an[2], an[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(an[4], an[5], an[6], an[7]);
                }
            }
        };
    [[maybe_unused]] const bool defer = SEQ && more;   // (SEQ, another hop follows) the rows are stored behind the barrier below, from the accumulator registers
    // The loads of a batch -- row i, half of column tile j: two skip-row chunks and two per-graph term chunks per lane -- are issued one
    // batch AHEAD of the arithmetic and stores that consume them (two register sets; whole column tiles per batch spill).  Written as a plain loop the epilogue came out
    // as 32 x (loads, wait for all of them, arithmetic, store): one 16-byte skip read in flight per lane, 29 us per tile of pure
    // latency.  The per-column constants come from LDS.  (Round 6: two and three batches ahead -- three and four register sets -- measured SLOWER, same box: 374.7-377.6 us per
    // hop with one batch ahead, 376.7-378.1 with two, 382.3-384.7 with three: the extra sets spill, and the loads are no longer what the epilogue waits for.)
    {
        const float* cc_l = reinterpret_cast<const float*>(smem + HA_CC0);
        constexpr int NB = RT * TN * 2;
        float rf_[RT], tmask_[RT], vmax_[RT];
        int g_[RT];
        bool on_[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int r = (wr * RT + i) * 32 + m;
            on_[i] = r < cnt;
            rf_[i] = row_l[r];
            g_[i] = reinterpret_cast<const int*>(row_l)[HA_ROWS + r];
            tmask_[i] = (gterm_h && reinterpret_cast<const int*>(row_l)[2 * HA_ROWS + r] > 0) ? 1.f : 0.f;     // (no in-edges: empty softmax, no term)
            vmax_[i] = 0.f;
        }
        const float* gt_ = gterm_h ? gterm_h : binv_h;        // (no instruction terms: a mapped address -- row 0 of the column scales --, the mask is 0)
        const int64_t gt_ld = gterm_h ? a.t_ld : 0;
        // (LGT: the next hop's node logits are accumulated as the values are finished -- their LDS reads and FMAs run under the
        //  latency of the batches' loads instead of as a pass of their own: even / odd columns in separate packed chains)
        [[maybe_unused]] ha_f32x2 an2[LGT ? RT : 1][8];
        if constexpr (LGT) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) an2[i][jj] = ha_f32x2{0.f, 0.f};
        }
        [[maybe_unused]] const float* vn_l = reinterpret_cast<const float*>(smem + HA_VN0);
        float4 skA[2], tgA[2], skB[2], tgB[2];
        auto load_batch = [&](int b, float4 (&sk)[2], float4 (&tg)[2]) {
            const int i = b / (2 * TN), j = (b >> 1) % TN, qp = b & 1;
            int r = (wr * RT + i) * 32 + m, hb = hh;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(r), "+v"(hb));             // (addresses formed here, per batch: carried from batch to batch -- row parts, column parts -- they spill)
#endif
            const int gt_row = reinterpret_cast<const int*>(row_l)[3 * HA_ROWS + r];      // (row of graph_term: out of LDS per batch, not carried in registers)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cc = min(cbase + (wc * TN + j) * 32 + 8 * (2 * qp + u) + 4 * hb, C - 4);       // (columns past C: clamped re-reads, never used)
                sk[u] = ha_row_load(X4in_h + (((int64_t)t * NQ + (cc >> 2)) * HA_ROWS + r) * 4);
                tg[u] = *reinterpret_cast<const float4*>(gt_ + (int64_t)gt_row * gt_ld + cc);
            }
        };
        auto consume_batch = [&](int b, const float4 (&sk_)[2], const float4 (&tg_)[2]) {
            const int i = b / (2 * TN), j = (b >> 1) % TN, qp = b & 1;
            int r = (wr * RT + i) * 32 + m;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(r));
#endif
            const int64_t node = out_h ? reinterpret_cast<const int*>(row_l)[4 * HA_ROWS + r] : 0;      // (row of `out`: ns + r, or through the packed plan's row map)
            const float rf = rf_[i], tm = tmask_[i];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = 2 * qp + u;
                const float4 (&sk)[2] = sk_;
                const float4 (&tg)[2] = tg_;
                const int c0 = cbase + (wc * TN + j) * 32 + 8 * q + 4 * hh;
                if (c0 >= C) continue;
                const float4 bv = *reinterpret_cast<const float4*>(cc_l + c0);
                const float4 bi = *reinterpret_cast<const float4*>(cc_l + 512 + c0);
                float4 v;
                v.x = acc[i][j][4 * q] * (rf * bv.x) + tg[u].x * tm + bi.x + sk[u].x;
                v.y = acc[i][j][4 * q + 1] * (rf * bv.y) + tg[u].y * tm + bi.y + sk[u].y;
                v.z = acc[i][j][4 * q + 2] * (rf * bv.z) + tg[u].z * tm + bi.z + sk[u].z;
                v.w = acc[i][j][4 * q + 3] * (rf * bv.w) + tg[u].w * tm + bi.w + sk[u].w;
                if (relu_h) {                         // torch's eval BatchNorm: y = x (w invstd) + (b - mean w invstd), then ReLU
                    const float4 sc = *reinterpret_cast<const float4*>(cc_l + 1024 + c0);
                    const float4 sh = *reinterpret_cast<const float4*>(cc_l + 1536 + c0);
                    v.x = fmaxf(v.x * sc.x + sh.x, 0.f); v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
                    v.z = fmaxf(v.z * sc.z + sh.z, 0.f); v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
                }
                if (!on_[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!defer) {
                    if (X4out_h) ha_row_store(X4out_h + (((int64_t)t * CQ + (c0 >> 2)) * HA_ROWS + r) * 4, v);
                    if (out_h && on_[i]) ha_row_store(out_h + node * a.out_ld + c0, v);
                }
                vmax_[i] = fmaxf(vmax_[i], fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                if constexpr (SEQ) {                  // the finished values stay in the accumulator registers: stores and node logits below
                    acc[i][j][4 * q] = v.x; acc[i][j][4 * q + 1] = v.y; acc[i][j][4 * q + 2] = v.z; acc[i][j][4 * q + 3] = v.w;
                }
                if constexpr (LGT) {
                    const ha_f32x2 a01{v.x, v.y}, a23{v.z, v.w};
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const float4 vn = *reinterpret_cast<const float4*>(vn_l + jj * C + c0);
                        an2[i][jj] = a01 * ha_f32x2{vn.x, vn.y} + an2[i][jj];
                        an2[i][jj] = a23 * ha_f32x2{vn.z, vn.w} + an2[i][jj];
                    }
                }
            }
        };
        load_batch(0, skA, tgA);
        if constexpr (LGT) {                          // Vn of the next hop is in place everywhere (the first batch's loads travel meanwhile)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int b = 0; b < NB; b += 2) {
            if (b + 1 < NB) load_batch(b + 1, skB, tgB);
            consume_batch(b, skA, tgA);
            if (b + 1 < NB) {
                if (b + 2 < NB) load_batch(b + 2, skA, tgA);
                consume_batch(b + 1, skB, tgB);
            }
        }
#pragma unroll
        for (int i = 0; i < RT; ++i)
            if ((SEQ ? more : a.gmax_out != nullptr) && on_[i]) atomicMax(&gm_l[g_[i] - gf], __float_as_uint(vmax_[i]));
        if constexpr (LGT) {
            // the two column halves of a row (lanes m and m + 32) meet; lane m leaves the wave's partial at part[wc][r][0..8) (A' ring, idle)
            float* part = reinterpret_cast<float*>(smem + HA_A0);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                float an[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    an[jj] = an2[i][jj][0] + an2[i][jj][1];
                    an[jj] += __shfl_xor(an[jj], 32, 64);
                }
                if (hh == 0) {
                    float* dst = part + ((wc * HA_ROWS) + (wr * RT + i) * 32 + m) * 8;
                    *reinterpret_cast<float4*>(dst) = make_float4(an[0], an[1], an[2], an[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(an[4], an[5], an[6], an[7]);
                }
            }
        }
    }
    if constexpr (!SEQ) {
        if constexpr (LGT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the partial logits are in LDS; the row stores need not be complete)
            __builtin_amdgcn_s_barrier();
            const float* part = reinterpret_cast<const float*>(smem + HA_A0);
            float2 sacc = *reinterpret_cast<const float2*>(part + tid * 2);
#pragma unroll
            for (int w = 1; w < WC; ++w) {
                const float2 pw = *reinterpret_cast<const float2*>(part + w * HA_ROWS * 8 + tid * 2);
                sacc.x += pw.x; sacc.y += pw.y;
            }
            if ((tid >> 2) < cnt) *reinterpret_cast<float2*>(a.a_node_out + (int64_t)cpart * a.an_part_stride + (int64_t)(ns + (tid >> 2)) * 8 + (tid & 3) * 2) = sacc;
        }
        if (a.gmax_out) {
            __syncthreads();
            const int ngl = a.node_graph[ns + cnt - 1] - gf + 1;
            if (tid < ngl) a.gmax_out[(int64_t)cpart * a.gm_part_stride + gf + tid] = __uint_as_float(gm_l[tid]);
        }
    } else if (more) {
        // ---- coefficient phase of hop + 1, inside the workgroup ---------------------------------------------------------------
        GVQA_HA_STAMP(hop, 4);                                // epilogue arithmetic done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's Vn / edge-half / weight DMAs have landed (no store is in flight yet)
        __syncthreads();                                      // ... everybody's; the per-graph maxima are complete; the x ring is idle
        // the rows leave chunk-major -- this workgroup's OWN input of the next hop (its DMAs bring them back from step 0 on, its
        // epilogue reads the skip rows): workgroup scope, the waves of a workgroup share the CU's vector L1 and its XCD's L2; the
        // stores only have to be complete before the barrier that opens the next hop's loop.  (Agent-scope fences write back and
        // invalidate the XCD's whole L2, every hop of every tile: 2.37 ms per launch against 2.19 for the per-hop launches.)  Chunks
        // 0 .. 2 also go straight into the x ring: the next hop needs no priming DMA for them.
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(m), "+v"(hh));                // (opaque again: the row / chunk offsets are recomputed here, not carried -- spilled -- from the pass above)
#endif
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int r = (wr * RT + i) * 32 + m;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = cbase + (wc * TN + j) * 32 + 8 * q + 4 * hh;
                    if (c0 >= C) continue;
                    const float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    ha_row_store(X4out_h + (((int64_t)t * CQ + (c0 >> 2)) * HA_ROWS + r) * 4, v);
                    if (j == 0 && q < 2 && c0 < 12) *reinterpret_cast<float4*>(smem + HA_X0 + (c0 >> 2) * 2048 + r * 16) = v;
                }
        }
        node_logits_to_part();
        __syncthreads();
        {   // a_node = sum over the WC wave columns, in place in wave column 0's slots (thread: 2 of the 128 x 8 values)
            float* part = reinterpret_cast<float*>(smem + HA_A0);
            float2 sacc = *reinterpret_cast<const float2*>(part + tid * 2);
#pragma unroll
            for (int w = 1; w < WC; ++w) {
                const float2 pw = *reinterpret_cast<const float2*>(part + w * HA_ROWS * 8 + tid * 2);
                sacc.x += pw.x; sacc.y += pw.y;
            }
            *reinterpret_cast<float2*>(part + tid * 2) = sacc;
        }
        __syncthreads();
        GVQA_HA_STAMP(hop, 6);                                // node logits reduced
        {   // leaky-relu + segment softmax of this lane's (node pi, head ph) over its in-edges (gat_skip.py:183-190; the denominator's
            // + 1e-16 as in torch_geometric.utils.softmax); the coefficients go to the slice in LDS and, the first 8, to registers
            const float* an_s = reinterpret_cast<const float*>(smem + HA_A0);
            float* st = reinterpret_cast<float*>(smem + HA_ST0) + ph * HA_ECAP + plo;
            coeffs_from_lds(an_s, st, tlog, hs.slope);
            // the next hop's input rows are the rows just produced: their graph's largest magnitude anchors the two-piece scale
            set_row_scale(__uint_as_float(gm_l[pg - gf]), false);
            unsigned* gm_n = reinterpret_cast<unsigned*>(smem + ((hop & 1) ? HA_GM0 : HA_GM1));
            if (tid < HA_ROWS) gm_n[tid] = 0u;
        }
        // (no barrier here: the next hop's first LDS writes and its first x DMA come behind the wait + barrier at its top, which
        //  also completes the row stores above)
        GVQA_HA_STAMP(hop, 7);                                // coefficients of the next hop in place
    }
    }   // hop
#undef GVQA_HA_STEP
#undef GVQA_HA_MFR
#undef GVQA_HA_STAMP
}

bool hopagg_supported(int H, int C, int Dn, int max_row_group_edges) {
    return H == 4 && C == Dn && C % 4 == 0 && C >= 32 && C <= 512 && max_row_group_edges <= HA_ECAP;
}

int launch_hopagg(int H, const HopAggArgs& a, int num_groups, hipStream_t stream, int col_parts) {
    GVQA_REQUIRE(H == 4 && a.group_ptr && a.rowptr && a.csr_src && a.a_node_in && a.a_edge && a.csr_eid && a.node_graph && a.X4in && a.Wk && a.binv && a.epc &&
                 a.gmax_in && (a.X4out || a.out), GVQA_E_INVALID, "hopagg: null operand");
    GVQA_REQUIRE(a.C % 4 == 0 && a.NCT <= 16 && a.NQ >= 1, GVQA_E_UNSUPPORTED, "hopagg: needs C %% 4 == 0 and C <= 512");
    if (num_groups == 0) return GVQA_OK;
    HopAggArgs b = a;
    HopAggSeq none;
    memset(&none, 0, sizeof(none));
#ifdef GVQA_PROBES
    static const int dbg = []() { const char* v = getenv("GVQA_HOPAGG_DEBUG"); return v ? atoi(v) : 0; }();
    b.dbg = dbg;
#endif
    GVQA_REQUIRE(!a.a_node_out || a.Vn_next, GVQA_E_INVALID, "hopagg: node logits out need the next hop's folded attention vectors");
    const dim3 grid((unsigned)num_groups), block(512);
    const bool lg = a.a_node_out != nullptr, narrow = a.NCT <= 10;
    if (col_parts > 1) GVQA_REQUIRE(col_parts == 4 && a.NCT > 12 && a.NCT <= 16 && a.an_part_stride > 0 && a.gm_part_stride > 0, GVQA_E_INVALID, "hopagg: column parts are 4 x 128 columns of a 512-wide hop");
#define GVQA_HA_LAUNCH(LG_)                                                                                                  \
    do {                                                                                                                     \
        if (col_parts > 1) hipLaunchKernelGGL((k_hopagg4<4, 2, 1, 2, false, LG_, true, 4>), dim3((unsigned)num_groups, 4), block, 0, stream, b, none); \
        else if (narrow) hipLaunchKernelGGL((k_hopagg4<4, 2, 1, 5, false, LG_, true>), grid, block, 0, stream, b, none);       \
        else hipLaunchKernelGGL((k_hopagg4<2, 4, 2, 4, false, LG_, true>), grid, block, 0, stream, b, none);                   \
    } while (0)
    if (lg) GVQA_HA_LAUNCH(true);
    else GVQA_HA_LAUNCH(false);
#undef GVQA_HA_LAUNCH
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// The K hops as one launch (see k_hopagg4<..., SEQ>): one workgroup per row group walks all hops of its rows.
int launch_hopagg_seq(int H, const HopAggArgs& a, const HopAggSeq& hs, int num_groups, hipStream_t stream) {
    GVQA_REQUIRE(H == 4 && a.group_ptr && a.rowptr && a.csr_src && a.a_node_in && a.node_graph && a.gmax_in && a.out && hs.csr_eid && hs.a_edge &&
                 hs.X4a && hs.X4b && hs.Wk && hs.epc && hs.Vn && hs.K >= 1 && hs.K <= HA_MAXHOPS, GVQA_E_INVALID, "hopagg_seq: null operand / hop count");
    GVQA_REQUIRE(a.C % 4 == 0 && a.NCT <= 16 && a.NQ >= 1 && a.NQ * 4 == a.C, GVQA_E_UNSUPPORTED, "hopagg_seq: needs C == Dn, C %% 4 == 0 and C <= 512");
    if (num_groups == 0) return GVQA_OK;
    HopAggArgs b = a;
#ifdef GVQA_PROBES
    static const int dbg = []() { const char* v = getenv("GVQA_HOPAGG_DEBUG"); return v ? atoi(v) : 0; }();
    b.dbg = dbg & ~32;
    // GVQA_HOPAGG_STAMPS=<path>: after every launch the phase stamps of all workgroups are copied back (synchronously) and written there
    static const char* stamp_path = getenv("GVQA_HOPAGG_STAMPS");
    HopAggSeq h2 = hs;
    unsigned long long* dstamps = nullptr;
    const size_t nst = (size_t)num_groups * hs.K * 8;
    if (stamp_path && hipMalloc(&dstamps, nst * 8) == hipSuccess) { (void)hipMemsetAsync(dstamps, 0, nst * 8, stream); h2.stamps = dstamps; }
    if (dstamps) {
        if (a.NCT <= 10) hipLaunchKernelGGL((k_hopagg4<4, 2, 1, 5, true>), dim3((unsigned)num_groups), dim3(512), 0, stream, b, h2);
        else hipLaunchKernelGGL((k_hopagg4<2, 4, 2, 4, true>), dim3((unsigned)num_groups), dim3(512), 0, stream, b, h2);
        unsigned long long* host = static_cast<unsigned long long*>(malloc(nst * 8));
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(host, dstamps, nst * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(stamp_path, "wb")) { int hdr[2] = {num_groups, hs.K}; fwrite(hdr, sizeof(int), 2, f); fwrite(host, 8, nst, f); fclose(f); }
        free(host);
        (void)hipFree(dstamps);
        GVQA_LAUNCH_CHECK();
        return GVQA_OK;
    }
#endif
    if (a.NCT <= 10) hipLaunchKernelGGL((k_hopagg4<4, 2, 1, 5, true>), dim3((unsigned)num_groups), dim3(512), 0, stream, b, hs);
    else hipLaunchKernelGGL((k_hopagg4<2, 4, 2, 4, true>), dim3((unsigned)num_groups), dim3(512), 0, stream, b, hs);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // namespace gvqa
