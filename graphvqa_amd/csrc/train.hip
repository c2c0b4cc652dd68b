// Tall-skinny products of the differentiable path (training): the attention logits a = X V with V = the attention
// vectors folded through the projection weights ([D, J], J = 2H node columns or K*H edge columns; gat_skip.py:134-135,151 under
// autograd), and the two products of their backward, dV = X^T G and dX = G V^T.  X is [R, D] with R = nodes or edges (64k / 256k
// rows of 2 KB at config 3): every kernel here streams X (or dX) through HBM exactly once and is bound by that stream; the
// arithmetic rides on the f32-input matrix cores (v_mfma_f32_32x32x2_f32, J padded to 32 columns) because a VALU form needs a
// cross-lane reduction (forward) or an LDS operand per FMA (backward) that costs more than the padding.
//
//   k_skinny_fwd      Y[R, J]  = X V         one wave per 32-row tile (8 per workgroup); V^T k-slices from LDS; transposed accumulators (a lane
//                                             owns 4 consecutive j of one row: one 16-byte store per 8 columns)
//   k_skinny_dv       P[b][D, J] = X_b^T G_b  one workgroup per chunk of rows, wave w owns every 4th 32-column tile of D; the
//                                             row dimension is the MFMA's k; partial results per workgroup (no atomics: the
//                                             sum order is fixed), reduced by k_skinny_dv_reduce
//   k_head_rows_add / _bwd                      the per-graph rows of the projection kept out of xp (gvqa.h)
//   k_skinny_dx       dX[R, D] = addend + G V^T   J FMAs per element out of registers (a thread keeps the V rows of its 4
//                                             columns for 4 rows); 16-byte stores
#include "common.h"
#include "gemm_tile.h"
#include <algorithm>
#include <mutex>

namespace gvqa {
namespace {

constexpr int SK_JP = 32;          // J padded to the MFMA's 32 columns
constexpr int SK_ROWS_DV_MIN = 128;    // rows per workgroup of k_skinny_dv: 128 below 128k rows, 256 above

// Y = X V.  LDS: Vs[D_pad][32] (k-major, J zero-padded to 32) -- at most 64 KiB for D <= 512.
__global__ __launch_bounds__(512) void k_skinny_fwd(int64_t R, int D, int J, const float* __restrict__ X, int64_t ldx,
                                                    const float* __restrict__ V, float* __restrict__ Y) {
    extern __shared__ float Vs[];
    const int Dp = (D + 31) & ~31;
    for (int i = threadIdx.x; i < Dp * SK_JP; i += 512) {
        const int k = i >> 5, j = i & 31;
        Vs[i] = (k < D && j < J) ? V[(int64_t)k * J + j] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int64_t ntiles = (R + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 8) {
        const int64_t row = tile * 32 + r;
        const bool row_ok = row < R;
        const float* xr = X + (row_ok ? row : 0) * ldx;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // k chunk of 32: this lane's 16 consecutive k of its row (64 contiguous bytes), half-waves take the two halves
        float4 a[4], an[4];
        auto load = [&](int c, float4 (&dst)[4]) {
            const int k0 = c * 32 + hf * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // always a load (a conditional one is compiled into a predicated load + wait, which serialises the stream): rows
                // past R read row 0 and are not stored; k past D re-reads k = 0 against a zero row of Vs
                const int k = k0 + 4 * q;
                dst[q] = *reinterpret_cast<const float4*>(xr + (k < D ? k : 0));
            }
        };
        const int nch = Dp >> 5;
        auto fma = [&](int c, const float4 (&av)[4]) {
            const float* vs = Vs + (c * 32 + hf * 16) * SK_JP + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // operands swapped: D'[j][row], so a lane ends up with 4 consecutive j of its row per register quad
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(4 * q + 0) * SK_JP], av[q].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(4 * q + 1) * SK_JP], av[q].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(4 * q + 2) * SK_JP], av[q].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(4 * q + 3) * SK_JP], av[q].w, acc, 0, 0, 0);
            }
        };
        // two chunks' loads in flight: the next chunk's are issued before this chunk's MFMAs (scheduling barriers keep the
        // compiler from sinking them behind a vmcnt(0))
        load(0, a);
        for (int c = 0; c < nch; c += 2) {
            load(c + 1, an);
            __builtin_amdgcn_sched_barrier(0);
            fma(c, a);
            __builtin_amdgcn_sched_barrier(0);
            load(c + 2, a);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < nch) fma(c + 1, an);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (row_ok) {
            float* yr = Y + row * J;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j0 = 8 * q + 4 * hf;
                if ((J & 3) == 0) {
                    if (j0 < J) *reinterpret_cast<float4*>(yr + j0) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (j0 + t < J) yr[j0 + t] = acc[4 * q + t];
                }
            }
        }
    }
}

// The same for J <= 16 on v_mfma_f32_16x16x4_f32 (half the padding, half the matrix-core time: the 32-wide form is bound by
// it at node-sized X): one wave per 16-row tile, a lane loads 16 bytes of its row per 16-k step (16 rows x 64 contiguous bytes
// per instruction), four steps in flight; D'[j][row]: lane (row = l % 16, g = l / 16) ends with j = 4 g .. 4 g + 3 of its row.
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_skinny_fwd16(int64_t R, int D, int J, const float* __restrict__ X, int64_t ldx,
                                                      const float* __restrict__ V, float* __restrict__ Y) {
    extern __shared__ float Vs[];                 // [D padded to 64][16], k-major, J zero-padded to 16
    const int Dp = (D + 63) & ~63;
    for (int i = threadIdx.x; i < Dp * 16; i += 512) {
        const int k = i >> 4, j = i & 15;
        Vs[i] = (k < D && j < J) ? V[(int64_t)k * J + j] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int64_t ntiles = (R + 15) >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 8) {
        const int64_t row = tile * 16 + r;
        const bool row_ok = row < R;
        const float* xr = X + (row_ok ? row : 0) * ldx;
        sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float4 a[4], an[4];
        // a group = 64 k = four 16-k steps; always loads (rows past R read row 0 and are not stored; k past D re-reads k = 0 against
        // a zero row of Vs)
        auto load = [&](int c, float4 (&dst)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = c * 64 + q * 16 + 4 * g;
                dst[q] = *reinterpret_cast<const float4*>(xr + (k < D ? k : 0));
            }
        };
        auto fma = [&](int c, const float4 (&av)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* vs = Vs + (c * 64 + q * 16 + 4 * g) * 16 + r;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vs[0], av[q].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vs[16], av[q].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vs[32], av[q].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vs[48], av[q].w, acc, 0, 0, 0);
            }
        };
        const int nch = Dp >> 6;
        load(0, a);
        for (int c = 0; c < nch; c += 2) {
            load(c + 1, an);
            __builtin_amdgcn_sched_barrier(0);
            fma(c, a);
            __builtin_amdgcn_sched_barrier(0);
            load(c + 2, a);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < nch) fma(c + 1, an);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (row_ok) {
            float* yr = Y + row * J;
            const int j0 = 4 * g;
            if ((J & 3) == 0) {
                if (j0 < J) *reinterpret_cast<float4*>(yr + j0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (j0 + t < J) yr[j0 + t] = acc[t];
            }
        }
    }
}

// Partial dV of one chunk of rows: P[blockIdx.x][d][j] = sum over the chunk's rows of X[row][d] G[row][j].
// MFMA roles: M = d (32 per tile), N = j (32, zero past J), k = rows (2 per instruction: half-waves).
template <int DT>   // 32-column tiles of D per wave (D <= 128 DT)
__global__ __launch_bounds__(256) void k_skinny_dv(int64_t R, int D, int J, const float* __restrict__ X, int64_t ldx,
                                                   const float* __restrict__ G, float* __restrict__ P, int chunk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * chunk;
    const int nrow = (int)((R - row0 < chunk) ? (R - row0) : chunk);
    f32x16 acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    int dcol[DT];
    bool dok[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        dcol[t] = (wave + 4 * t) * 32 + r;
        dok[t] = dcol[t] < D;
        if (!dok[t]) dcol[t] = 0;
    }
    const bool jok = r < J;
    const int jcol = jok ? r : 0;
    constexpr int U = 8;     // row pairs per step; two steps' loads are in flight (the next step's are issued before this step's MFMAs)
    // unconditional loads from clamped addresses (conditional loads are compiled into predicated loads with a wait each: the
    // stream would be serialised)
    // (only G is masked -- rows past the chunk and columns past J multiply X by zero; columns of X past D land in accumulators
    // that are not stored -- and the mask is applied when the value is consumed, not where it is loaded, so that a step's 40
    // loads are all issued before anything waits for them)
    auto load = [&](int s, float (&xa)[U][DT], float (&gb)[U], float (&gm)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int lr = s + 2 * u + hf;
            const bool ok = lr < nrow;
            const int64_t row = row0 + (ok ? lr : 0);
            gm[u] = (ok && jok) ? 1.f : 0.f;
            gb[u] = G[row * J + jcol];
#pragma unroll
            for (int t = 0; t < DT; ++t) xa[u][t] = X[row * ldx + dcol[t]];
        }
    };
    auto fma = [&](const float (&xa)[U][DT], const float (&gb)[U], const float (&gm)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float gv = gb[u] * gm[u];
#pragma unroll
            for (int t = 0; t < DT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u][t], gv, acc[t], 0, 0, 0);
        }
    };
    float xa0[U][DT], gb0[U], gm0[U], xa1[U][DT], gb1[U], gm1[U];
    load(0, xa0, gb0, gm0);
    for (int s = 0; s < nrow; s += 4 * U) {
        load(s + 2 * U, xa1, gb1, gm1);             // rows past the chunk are masked to zero
        __builtin_amdgcn_sched_barrier(0);
        fma(xa0, gb0, gm0);
        __builtin_amdgcn_sched_barrier(0);
        load(s + 4 * U, xa0, gb0, gm0);
        __builtin_amdgcn_sched_barrier(0);
        fma(xa1, gb1, gm1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // C/D map: column (j) = lane & 31, row (d within the tile) = (i & 3) + 8 (i >> 2) + 4 hf
    float* p = P + (int64_t)blockIdx.x * D * J;
    if (jok) {
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int d0 = (wave + 4 * t) * 32 + 4 * hf;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int d = d0 + (i & 3) + 8 * (i >> 2);
                if (d < D) p[(int64_t)d * J + r] = acc[t][i];
            }
        }
    }
}

// dV[i] = sum over the partial results, in a fixed order: 16 groups of a 1024-thread workgroup take every 16th partial (8 loads
// in flight per thread), then the 16 group sums are added in index order.
__global__ __launch_bounds__(1024) void k_skinny_dv_reduce(int nparts, int DJ, const float* __restrict__ P, float* __restrict__ dV) {
    __shared__ float part[16][64];
    const int col = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    if (i < DJ) {
        int b = pg;
        for (; b + 16 * 7 < nparts; b += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += P[(int64_t)(b + 16 * u) * DJ + i];
        }
        for (; b < nparts; b += 16) s[0] += P[(int64_t)b * DJ + i];
    }
    part[pg][col] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (pg == 0 && i < DJ) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][col];
        dV[i] = t;
    }
}

// dX[row][d] = (addend ? addend[row][d] : 0) + sum_j G[row][j] V[d][j].  A workgroup owns 32 consecutive rows: their G rows go
// to LDS once (read back as broadcasts), a thread owns 4 consecutive d of 16 of the rows and keeps the V rows of its columns in
// registers; 16-byte stores.
constexpr int SK_DX_ROWS = 32;
template <int JJ>
__global__ __launch_bounds__(256) void k_skinny_dx(int64_t R, int D, const float* __restrict__ G, const float* __restrict__ V,
                                                   const float* __restrict__ addend, int64_t ld_add, float* __restrict__ dX,
                                                   int64_t ldx) {
    __shared__ __attribute__((aligned(16))) float gs[SK_DX_ROWS * JJ];
    const int per_row = D >> 2;                   // D % 4 == 0
    for (int64_t rb = blockIdx.x; rb * SK_DX_ROWS < R; rb += gridDim.x) {
        const int64_t r0 = rb * SK_DX_ROWS;
        const int nr = (int)((R - r0 < SK_DX_ROWS) ? (R - r0) : SK_DX_ROWS);
        __syncthreads();                          // the previous tile's readers are done
        for (int i = threadIdx.x; i < nr * JJ; i += 256) gs[i] = G[r0 * JJ + i];
        __syncthreads();
        for (int idx = threadIdx.x; idx < per_row * 2; idx += 256) {
            const int half = idx / per_row, c = (idx - half * per_row) * 4;
            float vt[4][JJ];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < JJ; ++j) vt[t][j] = V[(int64_t)(c + t) * JJ + j];
            const int q1 = min(nr, half * 16 + 16);
#pragma unroll 2
            for (int q = half * 16; q < q1; ++q) {
                const int64_t row = r0 + q;
                float4 o = addend ? *reinterpret_cast<const float4*>(addend + row * ld_add + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                float g[JJ];
                if constexpr (JJ % 4 == 0) {      // four coefficients per LDS read
#pragma unroll
                    for (int j = 0; j < JJ; j += 4) {
                        const float4 g4 = *reinterpret_cast<const float4*>(&gs[q * JJ + j]);
                        g[j] = g4.x; g[j + 1] = g4.y; g[j + 2] = g4.z; g[j + 3] = g4.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < JJ; ++j) g[j] = gs[q * JJ + j];
                }
#pragma unroll
                for (int j = 0; j < JJ; ++j) {
                    o.x = fmaf(g[j], vt[0][j], o.x);
                    o.y = fmaf(g[j], vt[1][j], o.y);
                    o.z = fmaf(g[j], vt[2][j], o.z);
                    o.w = fmaf(g[j], vt[3][j], o.w);
                }
                *reinterpret_cast<float4*>(dX + row * ldx + c) = o;
            }
        }
    }
}

// any J <= 32: the same, J a run-time value (V and G re-read per element quad; rarely used widths)
__global__ __launch_bounds__(256) void k_skinny_dx_any(int64_t R, int D, int J, const float* __restrict__ G, const float* __restrict__ V,
                                                       const float* __restrict__ addend, int64_t ld_add, float* __restrict__ dX,
                                                       int64_t ldx) {
    const int per_row = D >> 2;
    const int64_t total = R * per_row;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int c = (int)(it % per_row) * 4;
        const int64_t row = it / per_row;
        float4 o = addend ? *reinterpret_cast<const float4*>(addend + row * ld_add + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < J; ++j) {
            const float g = G[row * J + j];
            o.x = fmaf(g, V[(int64_t)(c + 0) * J + j], o.x);
            o.y = fmaf(g, V[(int64_t)(c + 1) * J + j], o.y);
            o.z = fmaf(g, V[(int64_t)(c + 2) * J + j], o.z);
            o.w = fmaf(g, V[(int64_t)(c + 3) * J + j], o.w);
        }
        *reinterpret_cast<float4*>(dX + row * ldx + c) = o;
    }
}

// ---- per-graph rows of the projection kept out of xp (gvqa.h: gvqa_graph_head_rows_*) ----------------------------------------
constexpr int HR_MAXH = 8;

// y[i, c..c+3] += (1/H) sum_h s[i,h] R[g(i), h, c..c+3] + bias[c..] + skip[i, c..]; a thread per (row, 4 channels)
template <int H>
__global__ __launch_bounds__(256) void k_head_rows_add(int64_t N, int C, const int32_t* __restrict__ node_graph,
                                                       const int32_t* __restrict__ rowptr, const float* __restrict__ R,
                                                       const float* __restrict__ s, const float* __restrict__ bias,
                                                       const float* __restrict__ skip, int64_t ld_skip, float* __restrict__ y, int64_t ldy) {
    const int per_row = C >> 2;
    const int64_t total = N * per_row;
    const float inv_h = 1.0f / H;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int64_t i = it / per_row;
        const int c = (int)(it - i * per_row) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (R) {
            const float* r = R + (int64_t)node_graph[i] * H * C + c;
            const float ones = rowptr[i + 1] > rowptr[i] ? 1.f : 0.f;      // s without a mask: 1, or 0 for a node without in-edges
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float4 v = *reinterpret_cast<const float4*>(r + (int64_t)h * C);
                const float w = s ? s[i * H + h] : ones;
                acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
            }
        }
        float4* o = reinterpret_cast<float4*>(y + i * ldy + c);
        float4 cur = *o;
        cur.x = fmaf(acc.x, inv_h, cur.x); cur.y = fmaf(acc.y, inv_h, cur.y); cur.z = fmaf(acc.z, inv_h, cur.z); cur.w = fmaf(acc.w, inv_h, cur.w);
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + c);
            cur.x += b.x; cur.y += b.y; cur.z += b.z; cur.w += b.w;
        }
        if (skip) {
            const float4 k = *reinterpret_cast<const float4*>(skip + i * ld_skip + c);
            cur.x += k.x; cur.y += k.y; cur.z += k.z; cur.w += k.w;
        }
        *o = cur;
    }
}

// one workgroup per graph: dR[g,h,c] = (1/H) sum_{i in g} s[i,h] dy[i,c] (threads over channels, nodes in order) and, when asked
// for, ds[i,h] = (1/H) dy[i,:] . R[g,h,:] (a wave per node, lanes over channels)
template <int H>
__global__ __launch_bounds__(256) void k_head_rows_bwd(int C, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ rowptr,
                                                       const float* __restrict__ dy, int64_t ld_dy,
                                                       const float* __restrict__ R, const float* __restrict__ s, float* __restrict__ dR,
                                                       float* __restrict__ ds, float* __restrict__ dcol) {
    const int g = blockIdx.x;
    const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
    const float inv_h = 1.0f / H;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        float4 acc[H];
        float4 col = make_float4(0.f, 0.f, 0.f, 0.f);                  // sum over the graph's rows of dy (bias gradient, per graph)
#pragma unroll
        for (int h = 0; h < H; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = n0; i < n1; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(dy + (int64_t)i * ld_dy + c);
            col.x += v.x; col.y += v.y; col.z += v.z; col.w += v.w;
            const float ones = rowptr[i + 1] > rowptr[i] ? 1.f : 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float w = s ? s[(int64_t)i * H + h] : ones;
                acc[h].x = fmaf(w, v.x, acc[h].x); acc[h].y = fmaf(w, v.y, acc[h].y);
                acc[h].z = fmaf(w, v.z, acc[h].z); acc[h].w = fmaf(w, v.w, acc[h].w);
            }
        }
        if (dR) {
#pragma unroll
            for (int h = 0; h < H; ++h)
                *reinterpret_cast<float4*>(dR + ((int64_t)g * H + h) * C + c) =
                    make_float4(acc[h].x * inv_h, acc[h].y * inv_h, acc[h].z * inv_h, acc[h].w * inv_h);
        }
        if (dcol) *reinterpret_cast<float4*>(dcol + (int64_t)g * C + c) = col;
    }
    if (!ds) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = n0 + wave; i < n1; i += 4) {
        float acc[H];
#pragma unroll
        for (int h = 0; h < H; ++h) acc[h] = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(dy + (int64_t)i * ld_dy + c);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float4 r = *reinterpret_cast<const float4*>(R + ((int64_t)g * H + h) * C + c);
                acc[h] += v.x * r.x + v.y * r.y + v.z * r.z + v.w * r.w;
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float v = acc[h];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) ds[(int64_t)i * H + h] = v * inv_h;
        }
    }
}

// ---- C = X^T Y on the fp16 matrix cores (the weight gradient dW = dy^T x of the hop projection: a reduction over all N rows) ------
// The split GEMM of csrc/split3.hip contracts over the contiguous dimension of fragment-major packed operands, so both operands
// are packed TRANSPOSED here (k_split2h_pack_t: X [R, M] -> two-piece fp16 fragments of X^T, rows of X = the k dimension), in
// split-K chunks of KC rows: chunk z is a complete packed operand [M x KC], the GEMM runs once with gridDim.z = S chunks into S
// partial results, and k_splitk_reduce adds them in a fixed order.  One power-of-two scale per operand (from its largest
// magnitude, k_absmax or the producer's): a piece pair carries 22 bits below each ELEMENT's own exponent as long as the element
// is within 2^18 of the largest one (fp16's exponent range), and what is smaller still contributes less than fp32 rounding of
// the sum.
typedef _Float16 tn_f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_absmax(int64_t rows, int cols, const float* __restrict__ X, int64_t ld, unsigned* __restrict__ out) {
    // a workgroup walks whole rows (blockIdx.x, then + gridDim.x); four 16-byte loads of a thread in flight
    const int c4 = cols >> 2;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    auto upd = [](float m, const float4& v) { return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); };
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float4* xr = reinterpret_cast<const float4*>(X + r * ld);
        int c = threadIdx.x;
        for (; c + 768 < c4; c += 1024) {
            const float4 a = xr[c], b = xr[c + 256], d = xr[c + 512], e = xr[c + 768];
            m0 = upd(m0, a); m1 = upd(m1, b); m2 = upd(m2, d); m3 = upd(m3, e);
        }
        for (; c < c4; c += 256) m0 = upd(m0, xr[c]);
    }
    float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // one atomic per workgroup: atomics on one address serialise at ~20 ns each
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(out, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));   // non-negative floats order like their bit patterns
}

constexpr int TN_SLAB_ROWS = 64, TN_SLAB_COLS = 256, TN_LDS_LD = 260;
// grid (S * KC / 64, ceil(M / 256)); a workgroup turns a [64 rows x 256 columns] slab of X into 4 k blocks x 8 column tiles.
// NT: the same slab also leaves the fragments of X itself (rows = rows of X, k = its columns: 2 row tiles x 16 k blocks) in the
// layout gvqa_split2h_pack writes, with the operand's one scale for every row -- the backward of a projection needs dy both
// ways (dx = dy W contracts over dy's columns, dW = dy^T x over its rows), and reads it once.
struct PackNt { uint16_t* P; float* inv; int RT, KB; };
template <bool NT>
__global__ __launch_bounds__(256) void k_split2h_pack_t(int64_t R, int M, const float* __restrict__ X, int64_t ldx,
                                                        const float* __restrict__ absmax, int nmax, int KC, int T,
                                                        uint16_t* __restrict__ P, float* __restrict__ inv, PackNt nt) {
    __shared__ float Xs[TN_SLAB_ROWS * TN_LDS_LD];
    __shared__ float mx_s[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * TN_SLAB_ROWS;         // (row slabs on grid.x: no 65535 limit)
    const int c0 = blockIdx.y * TN_SLAB_COLS;
    const int z = (int)(r0 / KC), kb0 = (int)((r0 - (int64_t)z * KC) >> 4), KBc = KC >> 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = tid + i * 256, row = idx >> 6, c = (idx & 63) * 4;
        const bool ok = r0 + row < R && c0 + c < M;
        const float* src = X + (ok ? (r0 + row) * ldx + c0 + c : 0);
        float4 v = *reinterpret_cast<const float4*>(src);
        const float m = ok ? 1.f : 0.f;
        v.x *= m; v.y *= m; v.z *= m; v.w *= m;
        *reinterpret_cast<float4*>(&Xs[row * TN_LDS_LD + c]) = v;
    }
    {   // the operand's largest magnitude = the maximum over the producer's slices
        float m = tid < nmax ? absmax[tid] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) mx_s[wave] = m;
    }
    __syncthreads();
    const int e = split2h_exponent(fmaxf(fmaxf(mx_s[0], mx_s[1]), fmaxf(mx_s[2], mx_s[3])));
    const float scale = pow2i(e);
    if (P && kb0 == 0 && c0 + tid < T * 32) inv[(int64_t)z * T * 32 + c0 + tid] = pow2i(-e);
    const int mloc = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int u = wave + 4 * j, kbl = u >> 3, tl = u & 7;
        const int tile = blockIdx.y * 8 + tl;
        if (tile >= T || !P) continue;
        tn_f16x8 p0, p1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = Xs[(kbl * 16 + kh * 8 + i) * TN_LDS_LD + tl * 32 + mloc] * scale;
            const _Float16 a = (_Float16)x;
            p0[i] = a;
            p1[i] = (_Float16)(x - (float)a);
        }
        uint16_t* o = P + ((int64_t)z * T + tile) * KBc * 1024 + (int64_t)(kb0 + kbl) * 1024 + lane * 8;
        *reinterpret_cast<uint4*>(o) = __builtin_bit_cast(uint4, p0);
        *reinterpret_cast<uint4*>(o + 512) = __builtin_bit_cast(uint4, p1);
    }
    if constexpr (NT) {
        const int64_t rt0 = r0 >> 5;
        if (blockIdx.y == 0 && tid < TN_SLAB_ROWS && rt0 + (tid >> 5) < nt.RT) nt.inv[r0 + tid] = pow2i(-e);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = wave + 4 * j, rtl = u >> 4, kbl = u & 15;
            const int64_t rt = rt0 + rtl;
            const int kb = (c0 >> 4) + kbl;
            if (rt >= nt.RT || kb >= nt.KB) continue;
            const float* xr = &Xs[(rtl * 32 + mloc) * TN_LDS_LD + kbl * 16 + kh * 8];
            const float4 a = *reinterpret_cast<const float4*>(xr), b = *reinterpret_cast<const float4*>(xr + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            tn_f16x8 p0, p1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float x = v[i] * scale;
                const _Float16 hi = (_Float16)x;
                p0[i] = hi;
                p1[i] = (_Float16)(x - (float)hi);
            }
            uint16_t* o = nt.P + (rt * nt.KB + kb) * 1024 + lane * 8;
            *reinterpret_cast<uint4*>(o) = __builtin_bit_cast(uint4, p0);
            *reinterpret_cast<uint4*>(o + 512) = __builtin_bit_cast(uint4, p1);
        }
    }
}

// C[m, n] = sum_z P[z][m][n], z in order
__global__ __launch_bounds__(256) void k_splitk_reduce(int S, int M, int N, const float* __restrict__ P, float* __restrict__ C, int64_t ldc) {
    const int n4 = N >> 2;
    const int64_t total = (int64_t)M * n4, MN = (int64_t)M * N;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int64_t m = it / n4;
        const int n = (int)(it - m * n4) * 4;
        float4 acc = *reinterpret_cast<const float4*>(P + m * N + n);
        for (int z = 1; z < S; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(P + (int64_t)z * MN + m * N + n);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(C + m * ldc + n) = acc;
    }
}

// ---- attention vectors folded through a projection weight, and the adjoint (gvqa.h: gvqa_fold_attention_*) ----------------------
// V[k, h] = sum_c W[h C + c, k] att_a[h C + c]  (and V[k, H + h] with att_b): grid (ceil(Kin / 64), H), 1024 threads = 64 columns
// x 16 slices of c; the slices' sums are added in index order
__global__ __launch_bounds__(1024) void k_fold_att_fwd(int H, int C, int Kin, const float* __restrict__ W, int64_t ldw,
                                                       const float* __restrict__ att_a, const float* __restrict__ att_b,
                                                       float* __restrict__ V, int J) {
    __shared__ float pa[16][64], pb[16][64];
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + col, h = blockIdx.y;
    float sa = 0.f, sb = 0.f;
    if (k < Kin) {
        for (int c = part; c < C; c += 16) {
            const int r = h * C + c;
            const float w = W[(int64_t)r * ldw + k];
            sa = fmaf(w, att_a[r], sa);
            if (att_b) sb = fmaf(w, att_b[r], sb);
        }
    }
    pa[part][col] = sa; pb[part][col] = sb;
    __syncthreads();
    if (part == 0 && k < Kin) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { ta += pa[q][col]; tb += pb[q][col]; }
        V[(int64_t)k * J + h] = ta;
        if (att_b) V[(int64_t)k * J + H + h] = tb;
    }
}
// dW[r, k] = att_a[r] dV[k, h] + att_b[r] dV[k, H + h]  (r = h C + c): a thread per (row, column)
__global__ __launch_bounds__(256) void k_fold_att_bwd_w(int H, int C, int Kin, const float* __restrict__ att_a, const float* __restrict__ att_b,
                                                        const float* __restrict__ dV, int J, float* __restrict__ dW, int64_t ld_dw) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= Kin) return;
    for (int r = blockIdx.y; r < H * C; r += gridDim.y) {
        const int h = r / C;
        float v = att_a[r] * dV[(int64_t)k * J + h];
        if (att_b) v = fmaf(att_b[r], dV[(int64_t)k * J + H + h], v);
        dW[(int64_t)r * ld_dw + k] = v;
    }
}
// datt_a[r] = sum_k W[r, k] dV[k, h]  (and datt_b with dV[k, H + h]): a wave per row
__global__ __launch_bounds__(256) void k_fold_att_bwd_att(int H, int C, int Kin, const float* __restrict__ W, int64_t ldw,
                                                          const float* __restrict__ dV, int J, float* __restrict__ datt_a,
                                                          float* __restrict__ datt_b) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= H * C) return;
    const int h = r / C;
    float sa = 0.f, sb = 0.f;
    for (int k = lane; k < Kin; k += 64) {
        const float w = W[(int64_t)r * ldw + k];
        sa = fmaf(w, dV[(int64_t)k * J + h], sa);
        if (datt_b) sb = fmaf(w, dV[(int64_t)k * J + H + h], sb);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
    if (lane == 0) {
        datt_a[r] = sa;
        if (datt_b) datt_b[r] = sb;
    }
}

struct TnPlan {
    int S, KC, KBc, TA, TB;
    size_t off_pa, off_pb, off_ia, off_ib, off_part, off_max, total;
};
static TnPlan tn_plan(int64_t R, int64_t M, int64_t N) {
    TnPlan p;
    const int64_t tiles = cdiv(M, 256) * cdiv(N, 256);
    // (chunks of >= 256 rows: a reduction over a few thousand rows -- the per-graph products of the hop, R = graphs -- still spreads over the chip)
    int64_t S = std::max<int64_t>(1, std::min<int64_t>((256 + tiles - 1) / tiles, cdiv(R, R >= 16384 ? 1024 : 256)));
    S = std::min<int64_t>(S, 1024);
    const int64_t KC = cdiv(cdiv(std::max<int64_t>(R, 1), S), 64) * 64;
    p.S = (int)cdiv(std::max<int64_t>(R, 1), KC);
    p.KC = (int)KC; p.KBc = (int)(KC / 16);
    p.TA = (int)cdiv(M, 32); p.TB = (int)cdiv(N, 32);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    p.off_pa = take((size_t)p.S * p.TA * p.KBc * 2048);
    p.off_pb = take((size_t)p.S * p.TB * p.KBc * 2048);
    p.off_ia = take((size_t)p.S * p.TA * 32 * 4);
    p.off_ib = take((size_t)p.S * p.TB * 32 * 4);
    p.off_part = take((size_t)p.S * M * N * 4);
    p.off_max = take(256);
    p.total = off;
    return p;
}

// backward of y = x W^T: dx = dy W (contracts over dy's columns) and dW = dy^T x (over its rows)
struct LbPlan {
    TnPlan tn;
    int KCw, KBw, RT, TBw;
    size_t off_nt, off_inv_nt, off_wt, off_inv_wt, off_max, total;
};
static LbPlan lb_plan(int64_t R, int64_t M, int64_t K) {
    LbPlan p;
    p.tn = tn_plan(R, M, K);
    p.KCw = (int)(cdiv(M, 64) * 64); p.KBw = p.KCw / 16;
    p.RT = (int)cdiv(std::max<int64_t>(R, 1), 32); p.TBw = (int)cdiv(K, 32);
    size_t off = p.tn.total;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    p.off_nt = take((size_t)p.RT * p.KBw * 2048);
    p.off_inv_nt = take(((size_t)p.RT * 32 + 64) * 4);
    p.off_wt = take((size_t)p.TBw * p.KBw * 2048);
    p.off_inv_wt = take((size_t)p.TBw * 32 * 4);
    p.off_max = take(256);
    p.total = off;
    return p;
}

}  // namespace

// split GEMM of csrc/split3.hip
int launch_linear_split(int np, int64_t M, int64_t N, int64_t K, const void* Apk, const void* Bpk, LinearEpilogue ep, float* C,
                        int64_t ldc, hipStream_t stream, int batch, const float* a_inv_batched, const float* b_inv_batched);
}  // namespace gvqa

using namespace gvqa;

extern "C" {

int gvqa_skinny_forward(int64_t R, int64_t D, int64_t J, const float* X, int64_t ldx, const float* V, float* Y, void* stream) {
    GVQA_REQUIRE(R >= 0 && D > 0 && J > 0 && J <= SK_JP, GVQA_E_INVALID, "gvqa_skinny_forward: J must be in [1, %d]", SK_JP);
    GVQA_REQUIRE(D % 4 == 0 && ldx % 4 == 0 && ldx >= D && D <= 1024, GVQA_E_INVALID, "gvqa_skinny_forward: D %% 4 == 0, D <= 1024, ldx %% 4 == 0");
    if (R == 0) return GVQA_OK;
    GVQA_REQUIRE(X && V && Y, GVQA_E_INVALID, "gvqa_skinny_forward: null pointer");
    const int Dp = ((int)D + 31) & ~31;
    const size_t lds = (size_t)Dp * SK_JP * sizeof(float);
    {   // the dynamic-LDS opt-in is a per-device function attribute: set once per device
        static std::mutex mu;
        static bool attr_set[64] = {};
        int dev = 0;
        GVQA_HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            GVQA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skinny_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * SK_JP * 4));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    if (J <= 16) {       // (at most 64 KiB of LDS: below the default limit)
        const int Dp16 = ((int)D + 63) & ~63;
        const int grid16 = (int)std::min<int64_t>(cdiv(cdiv(R, 16), 8), 512);
        hipLaunchKernelGGL(k_skinny_fwd16, dim3(grid16), dim3(512), (size_t)Dp16 * 16 * sizeof(float), (hipStream_t)stream, R, (int)D, (int)J, X,
                           ldx, V, Y);
        GVQA_LAUNCH_CHECK();
        return GVQA_OK;
    }
    const int64_t ntiles = cdiv(R, 32);
    const int grid = (int)std::min<int64_t>(cdiv(ntiles, 8), 512);
    hipLaunchKernelGGL(k_skinny_fwd, dim3(grid), dim3(512), lds, (hipStream_t)stream, R, (int)D, (int)J, X, ldx, V, Y);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

size_t gvqa_skinny_backward_weight_workspace_bytes(int64_t R, int64_t D, int64_t J) {
    return (size_t)std::max<int64_t>(cdiv(R, SK_ROWS_DV_MIN), 1) * (size_t)D * (size_t)J * sizeof(float) + 256;
}

int gvqa_skinny_backward_weight(int64_t R, int64_t D, int64_t J, const float* X, int64_t ldx, const float* G, float* dV,
                                void* ws, size_t ws_bytes, void* stream) {
    GVQA_REQUIRE(R >= 0 && D > 0 && J > 0 && J <= SK_JP && D <= 1024 && ldx >= D, GVQA_E_INVALID,
                 "gvqa_skinny_backward_weight: J in [1, %d], D <= 1024", SK_JP);
    GVQA_REQUIRE(dV, GVQA_E_INVALID, "gvqa_skinny_backward_weight: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) {
        GVQA_HIP_CHECK(hipMemsetAsync(dV, 0, (size_t)D * J * sizeof(float), st));
        return GVQA_OK;
    }
    GVQA_REQUIRE(X && G && ws, GVQA_E_INVALID, "gvqa_skinny_backward_weight: null pointer");
    GVQA_REQUIRE(ws_bytes >= gvqa_skinny_backward_weight_workspace_bytes(R, D, J), GVQA_E_WORKSPACE,
                 "gvqa_skinny_backward_weight: workspace too small");
    const int chunk = R >= (1 << 17) ? 2 * SK_ROWS_DV_MIN : SK_ROWS_DV_MIN;
    const int nparts = (int)cdiv(R, chunk);
    float* P = static_cast<float*>(ws);
    const int dtiles = (int)cdiv(D, 32), DT = (int)cdiv(dtiles, 4);
#define GVQA_DV(T) hipLaunchKernelGGL(k_skinny_dv<T>, dim3(nparts), dim3(256), 0, st, R, (int)D, (int)J, X, ldx, G, P, chunk)
    switch (DT) {
        case 1: GVQA_DV(1); break;
        case 2: GVQA_DV(2); break;
        case 3: GVQA_DV(3); break;
        case 4: GVQA_DV(4); break;
        case 5: GVQA_DV(5); break;
        case 6: GVQA_DV(6); break;
        case 7: GVQA_DV(7); break;
        default: GVQA_DV(8); break;
    }
#undef GVQA_DV
    GVQA_LAUNCH_CHECK();
    const int DJ = (int)(D * J);
    hipLaunchKernelGGL(k_skinny_dv_reduce, dim3((unsigned)cdiv(DJ, 64)), dim3(1024), 0, st, nparts, DJ, P, dV);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

int gvqa_skinny_backward_input(int64_t R, int64_t D, int64_t J, const float* G, const float* V, const float* addend,
                               int64_t ld_add, float* dX, int64_t ldx, void* stream) {
    GVQA_REQUIRE(R >= 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldx >= D && J > 0 && J <= SK_JP, GVQA_E_INVALID,
                 "gvqa_skinny_backward_input: D %% 4 == 0, ldx %% 4 == 0, J in [1, %d]", SK_JP);
    GVQA_REQUIRE(!addend || (ld_add % 4 == 0 && ld_add >= D), GVQA_E_INVALID, "gvqa_skinny_backward_input: ld_add %% 4 == 0");
    if (R == 0) return GVQA_OK;
    GVQA_REQUIRE(G && V && dX, GVQA_E_INVALID, "gvqa_skinny_backward_input: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = R * (D / 4);                                 // (run-time-J kernel: a thread per element quad)
    const int grid = (int)std::min<int64_t>(cdiv(R, SK_DX_ROWS), 16384);
#define GVQA_DX(JJ) hipLaunchKernelGGL(k_skinny_dx<JJ>, dim3(grid), dim3(256), 0, st, R, (int)D, G, V, addend, ld_add, dX, ldx)
    switch (J) {
        case 1: GVQA_DX(1); break;
        case 2: GVQA_DX(2); break;
        case 4: GVQA_DX(4); break;
        case 8: GVQA_DX(8); break;
        case 12: GVQA_DX(12); break;
        case 16: GVQA_DX(16); break;
        case 20: GVQA_DX(20); break;
        case 24: GVQA_DX(24); break;
        case 32: GVQA_DX(32); break;
        default:
            hipLaunchKernelGGL(k_skinny_dx_any, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 8192)), dim3(256), 0, st, R, (int)D, (int)J, G, V, addend,
                               ld_add, dX, ldx);
            break;
    }
#undef GVQA_DX
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

int gvqa_graph_head_rows_add(const gvqa_graph* g, int64_t C, int64_t H, const float* R, const float* s, const float* bias,
                             const float* skip, int64_t ld_skip, float* y, int64_t ld_y, void* stream) {
    GVQA_REQUIRE(g && g->valid, GVQA_E_INVALID, "graph_head_rows_add: graph not built");
    GVQA_REQUIRE(C > 0 && C % 4 == 0 && ld_y % 4 == 0 && ld_y >= C && H >= 1 && H <= HR_MAXH, GVQA_E_INVALID,
                 "graph_head_rows_add: C %% 4 == 0, 1 <= H <= %d", HR_MAXH);
    if (g->num_nodes == 0) return GVQA_OK;
    GVQA_REQUIRE(y && (!skip || (ld_skip % 4 == 0 && ld_skip >= C)), GVQA_E_INVALID, "graph_head_rows_add: null tensor / skip stride");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = g->num_nodes * (C / 4);
    const int grid = (int)std::min<int64_t>(cdiv(total, 256), 16384);
#define GVQA_HRA(HH) hipLaunchKernelGGL(k_head_rows_add<HH>, dim3(grid), dim3(256), 0, st, g->num_nodes, (int)C, g->node_graph, g->rowptr, R, s, bias, skip, ld_skip, y, ld_y)
    switch (H) {
        case 1: GVQA_HRA(1); break;
        case 2: GVQA_HRA(2); break;
        case 3: GVQA_HRA(3); break;
        case 4: GVQA_HRA(4); break;
        case 5: GVQA_HRA(5); break;
        case 6: GVQA_HRA(6); break;
        case 7: GVQA_HRA(7); break;
        default: GVQA_HRA(8); break;
    }
#undef GVQA_HRA
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

int gvqa_graph_head_rows_backward(const gvqa_graph* g, int64_t C, int64_t H, const float* dy, int64_t ld_dy, const float* R,
                                  const float* s, float* dR, float* ds, float* dcol, void* stream) {
    GVQA_REQUIRE(g && g->valid, GVQA_E_INVALID, "graph_head_rows_backward: graph not built");
    GVQA_REQUIRE(C > 0 && C % 4 == 0 && ld_dy % 4 == 0 && ld_dy >= C && H >= 1 && H <= HR_MAXH, GVQA_E_INVALID,
                 "graph_head_rows_backward: C %% 4 == 0, 1 <= H <= %d", HR_MAXH);
    if (g->num_graphs == 0) return GVQA_OK;
    GVQA_REQUIRE((dR || dcol) && (dy || g->num_nodes == 0) && (!ds || R), GVQA_E_INVALID, "graph_head_rows_backward: null tensor");
    hipStream_t st = (hipStream_t)stream;
#define GVQA_HRB(HH) hipLaunchKernelGGL(k_head_rows_bwd<HH>, dim3((unsigned)g->num_graphs), dim3(256), 0, st, (int)C, g->graph_ptr, g->rowptr, dy, ld_dy, R, s, dR, ds, dcol)
    switch (H) {
        case 1: GVQA_HRB(1); break;
        case 2: GVQA_HRB(2); break;
        case 3: GVQA_HRB(3); break;
        case 4: GVQA_HRB(4); break;
        case 5: GVQA_HRB(5); break;
        case 6: GVQA_HRB(6); break;
        case 7: GVQA_HRB(7); break;
        default: GVQA_HRB(8); break;
    }
#undef GVQA_HRB
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

size_t gvqa_linear_tn_workspace_bytes(int64_t R, int64_t M, int64_t N) {
    if (R < 0 || M <= 0 || N <= 0) return 0;
    return tn_plan(R, M, N).total;
}

int gvqa_linear_tn_split2h(int64_t R, int64_t M, int64_t N, const float* X, int64_t ldx, const float* Y, int64_t ldy,
                           const float* x_absmax, int x_absmax_n, const float* y_absmax, int y_absmax_n, float* C, int64_t ldc,
                           void* ws, size_t ws_bytes, void* stream) {
    GVQA_REQUIRE((!x_absmax || (x_absmax_n >= 1 && x_absmax_n <= GVQA_ABSMAX_SLOTS)) && (!y_absmax || (y_absmax_n >= 1 && y_absmax_n <= GVQA_ABSMAX_SLOTS)),
                 GVQA_E_INVALID, "linear_tn: 1 <= absmax count <= %d", GVQA_ABSMAX_SLOTS);
    GVQA_REQUIRE(R >= 0 && M > 0 && N > 0 && M < (1ll << 30) && N < (1ll << 30) && R < (1ll << 40), GVQA_E_INVALID, "linear_tn: bad size");
    GVQA_REQUIRE(M % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldc % 4 == 0 && ldx >= M && ldy >= N && ldc >= N,
                 GVQA_E_INVALID, "linear_tn: M, N and the leading dimensions must be multiples of 4");
    GVQA_REQUIRE(C, GVQA_E_INVALID, "linear_tn: null result");
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) {
        GVQA_HIP_CHECK(hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, st));
        return GVQA_OK;
    }
    GVQA_REQUIRE(X && Y && ws, GVQA_E_INVALID, "linear_tn: null operand");
    const TnPlan p = tn_plan(R, M, N);
    GVQA_REQUIRE(ws_bytes >= p.total, GVQA_E_WORKSPACE, "linear_tn: workspace too small (%zu < %zu)", ws_bytes, p.total);
    char* base = static_cast<char*>(ws);
    uint16_t* PA = reinterpret_cast<uint16_t*>(base + p.off_pa);
    uint16_t* PB = reinterpret_cast<uint16_t*>(base + p.off_pb);
    float* IA = reinterpret_cast<float*>(base + p.off_ia);
    float* IB = reinterpret_cast<float*>(base + p.off_ib);
    float* part = reinterpret_cast<float*>(base + p.off_part);
    unsigned* mx = reinterpret_cast<unsigned*>(base + p.off_max);
    if (!x_absmax || !y_absmax) {
        GVQA_HIP_CHECK(hipMemsetAsync(mx, 0, 8, st));
        if (!x_absmax) {
            hipLaunchKernelGGL(k_absmax, dim3((unsigned)std::min<int64_t>(R, 1024)), dim3(256), 0, st, R, (int)M, X, ldx, mx);
            x_absmax = reinterpret_cast<const float*>(mx); x_absmax_n = 1;
        }
        if (!y_absmax) {
            hipLaunchKernelGGL(k_absmax, dim3((unsigned)std::min<int64_t>(R, 1024)), dim3(256), 0, st, R, (int)N, Y, ldy, mx + 1);
            y_absmax = reinterpret_cast<const float*>(mx + 1); y_absmax_n = 1;
        }
        GVQA_LAUNCH_CHECK();
    }
    float* dst = p.S == 1 ? C : part;
    int rc;
    if (get_option(GVQA_OPT_TN_DIRECT) && linear_tn_direct_applies(p.KC, ldx, ldy) && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 7) == 0) {
        rc = launch_linear_tn_direct(R, M, N, X, ldx, Y, ldy, x_absmax, x_absmax_n, y_absmax, y_absmax_n, p.KC, p.S, dst, p.S == 1 ? ldc : N, M * N, st);
    } else {
        const unsigned slabs = (unsigned)((int64_t)p.S * p.KC / TN_SLAB_ROWS);
        hipLaunchKernelGGL(k_split2h_pack_t<false>, dim3(slabs, (unsigned)cdiv(M, TN_SLAB_COLS)), dim3(256), 0, st, R, (int)M, X, ldx, x_absmax,
                           x_absmax_n, p.KC, p.TA, PA, IA, PackNt{});
        hipLaunchKernelGGL(k_split2h_pack_t<false>, dim3(slabs, (unsigned)cdiv(N, TN_SLAB_COLS)), dim3(256), 0, st, R, (int)N, Y, ldy, y_absmax,
                           y_absmax_n, p.KC, p.TB, PB, IB, PackNt{});
        GVQA_LAUNCH_CHECK();
        LinearEpilogue ep{nullptr, nullptr, 0, nullptr, 0, 0};
        ep.zs_a = (int64_t)p.TA * p.KBc * 1024; ep.zs_b = (int64_t)p.TB * p.KBc * 1024; ep.zs_c = M * N;
        ep.zs_ia = (int64_t)p.TA * 32; ep.zs_ib = (int64_t)p.TB * 32;
        rc = launch_linear_split(2, M, N, p.KC, PA, PB, ep, dst, p.S == 1 ? ldc : N, st, p.S, IA, IB);
    }
    if (rc != GVQA_OK) return rc;
    if (p.S > 1) {
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)std::min<int64_t>(cdiv(M * (N / 4), 256), 4096)), dim3(256), 0, st, p.S, (int)M, (int)N,
                           part, C, ldc);
        GVQA_LAUNCH_CHECK();
    }
    return GVQA_OK;
}

int gvqa_fold_attention_forward(int64_t H, int64_t C, int64_t Kin, const float* W, int64_t ldw, const float* att_a, const float* att_b,
                                float* V, void* stream) {
    GVQA_REQUIRE(H >= 1 && H <= 65535 && C >= 1 && Kin >= 1 && ldw >= Kin && H * C < (1ll << 31) && Kin < (1ll << 31), GVQA_E_INVALID,
                 "fold_attention_forward: bad sizes");
    GVQA_REQUIRE(W && att_a && V, GVQA_E_INVALID, "fold_attention_forward: null tensor");
    const int J = (int)(att_b ? 2 * H : H);
    hipLaunchKernelGGL(k_fold_att_fwd, dim3((unsigned)cdiv(Kin, 64), (unsigned)H), dim3(1024), 0, (hipStream_t)stream, (int)H, (int)C, (int)Kin,
                       W, ldw, att_a, att_b, V, J);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

int gvqa_fold_attention_backward(int64_t H, int64_t C, int64_t Kin, const float* W, int64_t ldw, const float* att_a, const float* att_b,
                                 const float* dV, float* dW, int64_t ld_dw, float* datt_a, float* datt_b, void* stream) {
    GVQA_REQUIRE(H >= 1 && H <= 65535 && C >= 1 && Kin >= 1 && ldw >= Kin && H * C < (1ll << 31) && Kin < (1ll << 31), GVQA_E_INVALID,
                 "fold_attention_backward: bad sizes");
    GVQA_REQUIRE(W && att_a && dV && (!dW || ld_dw >= Kin) && (!att_b == !datt_b || !datt_a), GVQA_E_INVALID,
                 "fold_attention_backward: null tensor / leading dimension");
    hipStream_t st = (hipStream_t)stream;
    const int J = (int)(att_b ? 2 * H : H);
    if (dW) hipLaunchKernelGGL(k_fold_att_bwd_w, dim3((unsigned)cdiv(Kin, 256), (unsigned)std::min<int64_t>(H * C, 1024)), dim3(256), 0, st, (int)H,
                               (int)C, (int)Kin, att_a, att_b, dV, J, dW, ld_dw);
    if (datt_a) hipLaunchKernelGGL(k_fold_att_bwd_att, dim3((unsigned)cdiv(H * C, 4)), dim3(256), 0, st, (int)H, (int)C, (int)Kin, W, ldw, dV, J,
                                   datt_a, att_b ? datt_b : nullptr);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

size_t gvqa_linear_backward_workspace_bytes(int64_t R, int64_t M, int64_t K) {
    if (R < 0 || M <= 0 || K <= 0) return 0;
    return lb_plan(R, M, K).total;
}

int gvqa_linear_backward_split2h(int64_t R, int64_t M, int64_t K, const float* dy, int64_t ld_dy, const float* W, int64_t ldw,
                                 const float* x, int64_t ldx, const float* dy_absmax, int dy_absmax_n, float* dx, int64_t ld_dx,
                                 int dx_accumulate, float* dW, int64_t ld_dw, void* ws, size_t ws_bytes, void* stream) {
    return gvqa_linear_backward_split2h_hint(R, M, K, dy, ld_dy, W, ldw, x, ldx, dy_absmax, dy_absmax_n, nullptr, 0, dx, ld_dx, dx_accumulate, dW, ld_dw,
                                             ws, ws_bytes, stream);
}

int gvqa_linear_backward_split2h_hint(int64_t R, int64_t M, int64_t K, const float* dy, int64_t ld_dy, const float* W, int64_t ldw,
                                      const float* x, int64_t ldx, const float* dy_absmax, int dy_absmax_n, const float* x_absmax, int x_absmax_n,
                                      float* dx, int64_t ld_dx, int dx_accumulate, float* dW, int64_t ld_dw, void* ws, size_t ws_bytes, void* stream) {
    gvqa_linear_backward_extras ex;
    memset(&ex, 0, sizeof(ex));
    ex.x_absmax = x_absmax; ex.x_absmax_n = x_absmax_n;
    return gvqa_linear_backward_split2h_ex(R, M, K, dy, ld_dy, W, ldw, x, ldx, dy_absmax, dy_absmax_n, dx, ld_dx, dx_accumulate, dW, ld_dw, &ex, ws, ws_bytes,
                                           stream);
}

int gvqa_linear_backward_split2h_ex(int64_t R, int64_t M, int64_t K, const float* dy, int64_t ld_dy, const float* W, int64_t ldw, const float* x,
                                    int64_t ldx, const float* dy_absmax, int dy_absmax_n, float* dx, int64_t ld_dx, int dx_accumulate, float* dW,
                                    int64_t ld_dw, const gvqa_linear_backward_extras* ex, void* ws, size_t ws_bytes, void* stream) {
    const float* x_absmax = ex ? ex->x_absmax : nullptr;
    int x_absmax_n = ex ? ex->x_absmax_n : 0;
    const float* lr_g = ex ? ex->lowrank_g : nullptr;
    const float* lr_v = ex ? ex->lowrank_v : nullptr;
    const int lr_J = ex ? ex->J : 0;
    const float* dx_addend = ex ? ex->addend : nullptr;
    const int64_t ld_addend = ex ? (ex->ld_addend ? ex->ld_addend : K) : 0;
    GVQA_REQUIRE(!x_absmax || (x_absmax_n >= 1 && x_absmax_n <= GVQA_ABSMAX_SLOTS), GVQA_E_INVALID, "linear_backward: 1 <= absmax count <= %d", GVQA_ABSMAX_SLOTS);
    GVQA_REQUIRE((!lr_g && !dx_addend) || dx, GVQA_E_INVALID, "linear_backward: the rank-J term and the addend belong to dx");
    GVQA_REQUIRE(R >= 0 && M > 0 && K > 0 && M < (1ll << 30) && K < (1ll << 30) && R < (1ll << 31), GVQA_E_INVALID, "linear_backward: bad size");
    GVQA_REQUIRE(M % 4 == 0 && K % 4 == 0 && ld_dy % 4 == 0 && ld_dy >= M, GVQA_E_INVALID, "linear_backward: M, K, ld_dy multiples of 4");
    GVQA_REQUIRE((!dx || (W && ldw % 4 == 0 && ldw >= K && ld_dx % 4 == 0 && ld_dx >= K)) && (!dW || (x && ldx % 4 == 0 && ldx >= K && ld_dw % 4 == 0 && ld_dw >= K)),
                 GVQA_E_INVALID, "linear_backward: operands of the requested gradients (leading dimensions multiples of 4)");
    GVQA_REQUIRE(!dy_absmax || (dy_absmax_n >= 1 && dy_absmax_n <= GVQA_ABSMAX_SLOTS), GVQA_E_INVALID, "linear_backward: 1 <= absmax count <= %d", GVQA_ABSMAX_SLOTS);
    if (!dx && !dW) return GVQA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) {
        if (dW) GVQA_HIP_CHECK(hipMemset2DAsync(dW, (size_t)ld_dw * 4, 0, (size_t)K * 4, (size_t)M, st));
        return GVQA_OK;
    }
    GVQA_REQUIRE(dy && ws, GVQA_E_INVALID, "linear_backward: null operand");
    const LbPlan p = lb_plan(R, M, K);
    GVQA_REQUIRE(ws_bytes >= p.total, GVQA_E_WORKSPACE, "linear_backward: workspace too small (%zu < %zu)", ws_bytes, p.total);
    char* base = static_cast<char*>(ws);
    uint16_t* PA = reinterpret_cast<uint16_t*>(base + p.tn.off_pa);
    uint16_t* PB = reinterpret_cast<uint16_t*>(base + p.tn.off_pb);
    float* IA = reinterpret_cast<float*>(base + p.tn.off_ia);
    float* IB = reinterpret_cast<float*>(base + p.tn.off_ib);
    float* part = reinterpret_cast<float*>(base + p.tn.off_part);
    uint16_t* PN = reinterpret_cast<uint16_t*>(base + p.off_nt);
    float* IN = reinterpret_cast<float*>(base + p.off_inv_nt);
    uint16_t* PW = reinterpret_cast<uint16_t*>(base + p.off_wt);
    float* IW = reinterpret_cast<float*>(base + p.off_inv_wt);
    unsigned* mx = reinterpret_cast<unsigned*>(base + p.off_max);       // [0] dy, [1] x, [2] W
    GVQA_HIP_CHECK(hipMemsetAsync(mx, 0, 16, st));
    if (!dy_absmax) {
        hipLaunchKernelGGL(k_absmax, dim3((unsigned)std::min<int64_t>(R, 1024)), dim3(256), 0, st, R, (int)M, dy, ld_dy, mx);
        dy_absmax = reinterpret_cast<const float*>(mx); dy_absmax_n = 1;
    }
    if (dW && !x_absmax) {
        hipLaunchKernelGGL(k_absmax, dim3((unsigned)std::min<int64_t>(R, 1024)), dim3(256), 0, st, R, (int)K, x, ldx, mx + 1);
        x_absmax = reinterpret_cast<const float*>(mx + 1); x_absmax_n = 1;
    }
    if (dx) hipLaunchKernelGGL(k_absmax, dim3((unsigned)std::min<int64_t>(M, 1024)), dim3(256), 0, st, M, (int)K, W, ldw, mx + 2);
    const unsigned slabs = (unsigned)((int64_t)p.tn.S * p.tn.KC / TN_SLAB_ROWS);
    const dim3 gdy(slabs, (unsigned)cdiv(M, TN_SLAB_COLS));
    // GVQA_OPT_TN_DIRECT: both products read dy (and x) as they are -- dW transposes on the way into its fragment image, dx converts its rows
    // of dy in the kernel (tn_direct.hip) -- and dy is not packed at all
    const bool direct = get_option(GVQA_OPT_TN_DIRECT) != 0 && linear_tn_direct_applies(p.tn.KC, ld_dy, dW ? ldx : 4) &&
                        ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dW ? x : nullptr)) & 7) == 0;
    const bool direct_dx = dx && get_option(GVQA_OPT_TN_DIRECT) != 0 && M % 16 == 0 && linear_nn_direct_applies(R, K, M, ld_dy) &&
                           ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dx_addend)) & 15) == 0 &&
                           ld_dx % 4 == 0 && ld_addend % 4 == 0;
    GVQA_REQUIRE((!lr_g && !dx_addend) || (direct_dx && (!dW || direct)), GVQA_E_UNSUPPORTED,
                 "linear_backward: the rank-J term / addend of dx ride in the direct product's epilogue (GVQA_OPT_TN_DIRECT, M %% 16 == 0, 16-byte aligned dy)");
    if (dx && direct_dx && (!dW || direct)) { /* no pack of dy */ }
    else if (dx) hipLaunchKernelGGL(k_split2h_pack_t<true>, gdy, dim3(256), 0, st, R, (int)M, dy, ld_dy, dy_absmax, dy_absmax_n, p.tn.KC, p.tn.TA,
                               (dW && !direct) ? PA : nullptr, IA, PackNt{PN, IN, p.RT, p.KBw});
    else if (!direct) hipLaunchKernelGGL(k_split2h_pack_t<false>, gdy, dim3(256), 0, st, R, (int)M, dy, ld_dy, dy_absmax, dy_absmax_n, p.tn.KC, p.tn.TA, PA, IA,
                                         PackNt{});
    GVQA_LAUNCH_CHECK();
    if (dx) {
        // B operand: W^T [K x M] = the transposed pack of W as ONE chunk of KCw rows
        hipLaunchKernelGGL(k_split2h_pack_t<false>, dim3((unsigned)(p.KCw / TN_SLAB_ROWS), (unsigned)cdiv(K, TN_SLAB_COLS)), dim3(256), 0, st, M, (int)K,
                           W, ldw, reinterpret_cast<const float*>(mx + 2), 1, p.KCw, p.TBw, PW, IW, PackNt{});
        GVQA_LAUNCH_CHECK();
        LinearEpilogue ep{nullptr, nullptr, 0, nullptr, 0, 0};
        if (dx_accumulate) { ep.addend = dx; ep.ld_add = ld_dx; }      // dx += dy W (the GEMM's epilogue reads the element it writes)
        const int rc = (direct_dx && (!dW || direct))
                           ? launch_linear_nn_direct(R, K, M, dy, ld_dy, dy_absmax, dy_absmax_n, PW, p.KBw, p.TBw, IW, dx, ld_dx, dx_accumulate, st, lr_g, lr_v,
                                                     lr_J, dx_addend, ld_addend)
                           : launch_linear_split(2, R, K, p.KCw, PN, PW, ep, dx, ld_dx, st, 1, IN, IW);
        if (rc != GVQA_OK) return rc;
    }
    if (dW) {
        float* dst = p.tn.S == 1 ? dW : part;
        int rc;
        if (direct) {
            rc = launch_linear_tn_direct(R, M, K, dy, ld_dy, x, ldx, dy_absmax, dy_absmax_n, x_absmax, x_absmax_n, p.tn.KC, p.tn.S, dst,
                                         p.tn.S == 1 ? ld_dw : K, M * K, st);
        } else {
            hipLaunchKernelGGL(k_split2h_pack_t<false>, dim3(slabs, (unsigned)cdiv(K, TN_SLAB_COLS)), dim3(256), 0, st, R, (int)K, x, ldx,
                               x_absmax, x_absmax_n, p.tn.KC, p.tn.TB, PB, IB, PackNt{});
            GVQA_LAUNCH_CHECK();
            LinearEpilogue ep{nullptr, nullptr, 0, nullptr, 0, 0};
            ep.zs_a = (int64_t)p.tn.TA * p.tn.KBc * 1024; ep.zs_b = (int64_t)p.tn.TB * p.tn.KBc * 1024; ep.zs_c = M * K;
            ep.zs_ia = (int64_t)p.tn.TA * 32; ep.zs_ib = (int64_t)p.tn.TB * 32;
            rc = launch_linear_split(2, M, K, p.tn.KC, PA, PB, ep, dst, p.tn.S == 1 ? ld_dw : K, st, p.tn.S, IA, IB);
        }
        if (rc != GVQA_OK) return rc;
        if (p.tn.S > 1) {
            hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)std::min<int64_t>(cdiv(M * (K / 4), 256), 4096)), dim3(256), 0, st, p.tn.S, (int)M,
                               (int)K, part, dW, ld_dw);
            GVQA_LAUNCH_CHECK();
        }
    }
    return GVQA_OK;
}

}  // extern "C"
