// Dense projections on the CDNA4 f32 matrix cores:  C[M,N] = A[M,K] . B[N,K]^T (+bias)(relu).
//
// This is torch.nn.Linear's math for the reference's node / edge projections
// (gat_skip.py:133,150; lcgn.py:144-149,230; GINE/GCN MLPs).  Exact fp32: the f32-input MFMA
// `v_mfma_f32_32x32x2_f32` is a k-ordered fmaf chain (no TF32/xf32 exists on gfx950), so parity
// with the fp32 reference holds at rounding level while running at the matrix-core rate.
//
// Tiling (64-wide wavefronts): block = 4 waves, block tile BM x BN, K step 32, both operands
// K-contiguous ("NT"), staged global -> registers -> LDS with 16-byte accesses and double
// buffered so the next tile's HBM/L2 loads fly under the current tile's MFMAs (one barrier per
// K step).  LDS rows are padded to 36 floats: a wave's ds_read_b128 fragment reads (16 rows at
// the same 16-byte column) then hit 16 distinct 16-byte slots of the 256-byte bank row.
// The K index inside a 32x32x2 step is free to permute (both operands use the same map), so
// lane-half kk = lane>>5 takes the 4 consecutive k's [8t+4kk, 8t+4kk+4) from ONE b128 read and
// feeds 4 MFMA steps from it.
#include <dlfcn.h>

#include <algorithm>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

#include "common.h"
#include "gemm_tile.h"

namespace gvqa {


// A16: A is stored as bf16; C16: C, and the addend / mul operands of the epilogue, are stored as bf16
// (leading dimensions are in elements of the respective type).  B (weights) and bias are fp32.
template <int BM, int BN, int WR, int WC, bool VEC, int BK = 32, bool A16 = false, bool C16 = false>
__global__ __launch_bounds__(64 * WR * WC) void k_linear_f32(int M, int N, int K, const float* __restrict__ A_,
                                                    int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                    LinearEpilogue ep, float* C_, int64_t ldc,
                                                    int64_t strideA, int64_t strideB, int64_t strideC) {
    typedef typename std::conditional<A16, uint16_t, float>::type TA;
    typedef typename std::conditional<C16, uint16_t, float>::type TC;
    const TA* A = reinterpret_cast<const TA*>(A_);
    TC* C = reinterpret_cast<TC*>(C_);
    constexpr int LDS_LD = BK + 4;                // padded leading dimension (floats)
    constexpr int RQ = BK / 4;                    // float4 per tile row
    constexpr int NTH = 64 * WR * WC;             // threads per block
    constexpr int WM = BM / WR, WN = BN / WC;     // wave tile
    constexpr int MT = WM / 32, NT = WN / 32;     // 32x32 MFMA tiles per wave
    constexpr int A_V4 = BM * BK / 4 / NTH;       // float4 staged per thread
    constexpr int B_V4 = (BN * BK / 4 + NTH - 1) / NTH;
    static_assert(BM * BK / 4 % NTH == 0, "A tile must divide evenly");

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_LD];

    A += (int64_t)blockIdx.z * strideA;
    B += (int64_t)blockIdx.z * strideB;
    C += (int64_t)blockIdx.z * strideC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_V4], rb[B_V4];

    // Branch-free tile loads: out-of-range rows are CLAMPED to a valid row (their products are never
    // stored) and the zero-fill of the K tail is a select applied only when the registers are
    // written to LDS (store_tile), so that nothing consumes the loaded values -- and forces a vmcnt
    // wait -- before the MFMAs of the current tile have been issued.  (A branch around a load, or an
    // early select, makes the prefetch synchronous.)  The per-thread row pointers are formed ONCE;
    // inside the K loop a load is pointer + uniform offset (the address/clamp arithmetic per tile
    // used to cost ~2.6 VALU instructions per MFMA).
    const TA* pa[A_V4];
    const float* pb[B_V4];
    int ca[A_V4], cb[B_V4];           // k offset of this thread's float4 within a tile
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
        const int idx = tid + i * NTH;
        ca[i] = (idx % RQ) * 4;
        pa[i] = A + (int64_t)min(m0 + idx / RQ, M - 1) * lda + ca[i];
    }
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
        const int idx = min(tid + i * NTH, BN * BK / 4 - 1);
        cb[i] = (idx % RQ) * 4;
        pb[i] = B + (int64_t)min(n0 + idx / RQ, N - 1) * ldb + cb[i];
    }
    auto load_vec = [&](const float* __restrict__ p, int k0, int c, bool tail) -> float4 {
        if (VEC) {
            if (!tail) return *reinterpret_cast<const float4*>(p + k0);
            return *reinterpret_cast<const float4*>(p + (min(k0 + c, K - 4) - c));
        }
        const float* q = p - c;          // row start
        return make_float4(q[min(k0 + c + 0, K - 1)], q[min(k0 + c + 1, K - 1)], q[min(k0 + c + 2, K - 1)],
                           q[min(k0 + c + 3, K - 1)]);
    };
    auto load_vec_a = [&](const TA* __restrict__ p, int k0, int c, bool tail) -> float4 {
        if constexpr (!A16) {
            return load_vec(reinterpret_cast<const float*>(p), k0, c, tail);
        } else {
            if (VEC) {      // 4 bf16 = 8 bytes
                const uint2 raw = *reinterpret_cast<const uint2*>(p + (tail ? (min(k0 + c, K - 4) - c) : k0));
                return make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xFFFF0000u),
                                   __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xFFFF0000u));
            }
            const uint16_t* q = reinterpret_cast<const uint16_t*>(p) - c;
            return make_float4(bf16_to_f32(q[min(k0 + c + 0, K - 1)]), bf16_to_f32(q[min(k0 + c + 1, K - 1)]),
                               bf16_to_f32(q[min(k0 + c + 2, K - 1)]), bf16_to_f32(q[min(k0 + c + 3, K - 1)]));
        }
    };
    auto mask_k = [&](float4 v, int gk) -> float4 {
        if (gk + 0 >= K) v.x = 0.f;
        if (gk + 1 >= K) v.y = 0.f;
        if (gk + 2 >= K) v.z = 0.f;
        if (gk + 3 >= K) v.w = 0.f;
        return v;
    };
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const bool tail = k0 + BK > K;       // block-uniform
#pragma unroll
        for (int i = 0; i < A_V4; ++i) ra[i] = load_vec_a(pa[i], k0, ca[i], tail);
#pragma unroll
        for (int i = 0; i < B_V4; ++i) rb[i] = load_vec(pb[i], k0, cb[i], tail);
    };
    auto store_tile = [&](int buf, int kt) {
        const int k0 = kt * BK;
        const bool tail = k0 + BK > K;      // block-uniform: only the last K tile needs masking
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            const int idx = tid + i * NTH;
            const int r = idx / RQ, c4 = (idx % RQ) * 4;
            float4 v = ra[i];
            if (tail) v = mask_k(v, k0 + c4);
            *reinterpret_cast<float4*>(&As[buf][r * LDS_LD + c4]) = v;
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            const int idx = tid + i * NTH;
            if (idx < BN * BK / 4) {
                const int r = idx / RQ, c4 = (idx % RQ) * 4;
                float4 v = rb[i];
                if (tail) v = mask_k(v, k0 + c4);
                *reinterpret_cast<float4*>(&Bs[buf][r * LDS_LD + c4]) = v;
            }
        }
    };

    const int nkt = (K + BK - 1) / BK;
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);   // global loads in flight during the MFMAs below
        const float* as = &As[cur][(wr * WM + frow) * LDS_LD + fk];
        const float* bs = &Bs[cur][(wc * WN + frow) * LDS_LD + fk];
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            float4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_LD + kg * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bf[j] = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_LD + kg * 8);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1, kt + 1);
        __syncthreads();
    }

    // Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gc = n0 + wc * WN + j * 32 + ccol;
        if (gc >= N) continue;
        const float bv = ep.bias ? ep.bias[gc] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int gr0 = m0 + wr * WM + i * 32 + crow0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = gr0 + (r & 3) + 8 * (r >> 2);
                if (gr < M) {
                    float v = acc[i][j][r] + bv;
                    if (ep.addend) v += load_elem<C16>(ep.addend, (int64_t)gr * ep.ld_add + gc);
                    if (ep.mul) v *= load_elem<C16>(ep.mul, (int64_t)gr * ep.ld_mul + gc);
                    if (ep.relu == 1) v = fmaxf(v, 0.f);
                    else if (ep.relu == 2) v = v > 0.f ? v : expf(v) - 1.f;   // ELU(alpha = 1), torch's exp(x) - 1 form
                    if constexpr (C16) C[(int64_t)gr * ldc + gc] = f32_to_bf16(v);
                    else C[(int64_t)gr * ldc + gc] = v;
                }
            }
        }
    }
}

// Epilogue from TRANSPOSED accumulators (MFMA operands swapped): D'[n, m] with "column" = lane & 31 -> m,
// "row" = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> n, so a lane owns 4 consecutive columns of one C row per register
// quad.  16 float4 stores per lane, no LDS round trip, no barrier.
__device__ __forceinline__ void store_tile_transposed(f32x16 (&acc)[2][2], int M, int N, int m0, int n0, int wr, int wc, int lane,
                                                      const LinearEpilogue& ep, float* C, int64_t ldc) {
    const int mrow = lane & 31, ncol0 = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int gr = m0 + wr * 64 + i * 32 + mrow;
        if (gr >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gc = n0 + wc * 64 + j * 32 + 8 * q + ncol0;
                if (gc >= N) continue;                 // N % 4 == 0: a quad is entirely inside or outside
                float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (ep.bias) {
                    const float4 b4 = *reinterpret_cast<const float4*>(ep.bias + gc);
                    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                }
                if (ep.addend) {
                    const float4 a4 = *reinterpret_cast<const float4*>(ep.addend + (int64_t)gr * ep.ld_add + gc);
                    v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
                }
                if (ep.mul) {
                    const float4 m4 = *reinterpret_cast<const float4*>(ep.mul + (int64_t)gr * ep.ld_mul + gc);
                    v.x *= m4.x; v.y *= m4.y; v.z *= m4.z; v.w *= m4.w;
                }
                if (ep.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                else if (ep.relu == 2) {
                    v.x = v.x > 0.f ? v.x : expf(v.x) - 1.f; v.y = v.y > 0.f ? v.y : expf(v.y) - 1.f;
                    v.z = v.z > 0.f ? v.z : expf(v.z) - 1.f; v.w = v.w > 0.f ? v.w : expf(v.w) - 1.f;
                }
                *reinterpret_cast<float4*>(C + (int64_t)gr * ldc + gc) = v;
            }
    }
}

// ----------------------------------------------------------------------------------------------
// LDS-DMA variant of the 128 x 128 tile (K % 32 == 0, 16-byte aligned rows).  The operand tiles go
// HBM/L2 -> LDS with `global_load_lds_dwordx4`: no VGPR staging, no ds_write, nothing for the waves to
// do between MFMAs but the fragment reads.  A K step is 32 floats = 128-byte tile rows, a wave
// instruction fills 8 of them (64 lanes x 16 B, lane-linear), and the tile is XOR-swizzled -- slot
// (row r, position q) holds k-chunk q ^ ((r >> 1) & 7), chosen by the lane through its GLOBAL address --
// so the 32 rows of a ds_read_b128 fragment read spread over all 16-byte columns of the bank row.
// The DMA is issued from inline asm and ordered by counted s_waitcnt vmcnt + barriers; tile t+1 lands
// while tile t is multiplied (a K step is 64 MFMAs = 4096 issue cycles per wave, far longer than the
// DMA latency).
__global__ __launch_bounds__(256) void k_linear_f32_dma(int M, int N, int K, const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb, LinearEpilogue ep,
                                                        float* C, int64_t ldc) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int TILE_BYTES = 128 * BK * 4;                  // one operand tile: 16 KiB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE_BYTES];      // {A, B} x 2 buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* pa[4];
    const float* pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 4 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        pa[j] = A + (int64_t)min(m0 + row, M - 1) * lda + chunk * 4;
        pb[j] = B + (int64_t)min(n0 + row, N - 1) * ldb + chunk * 4;
    }
    const int nt = K / BK;
    auto issue = [&](int buf) {
        const unsigned dst = lds_base + buf * 2 * TILE_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_b(pa[j], __builtin_amdgcn_readfirstlane(dst + j * 4096));
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_b(pb[j], __builtin_amdgcn_readfirstlane(dst + TILE_BYTES + j * 4096));
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa[j] += BK; pb[j] += BK; }
    };

    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 1) & 7;
    unsigned xo[4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) xo[kg] = (unsigned)(((kg * 2 + fh) ^ swz) * 16);
    const unsigned a_row = (unsigned)((wr * 64 + frow) * 128), b_row = (unsigned)((wc * 64 + frow) * 128);

    auto issue2 = [&](int buf, int q) {          // quarter q of a tile's DMAs: A rows and B rows of one 32-row group
        const unsigned dst = lds_base + buf * 2 * TILE_BYTES + wave * 1024 + q * 4096;
        lds_dma16_b(pa[q], __builtin_amdgcn_readfirstlane(dst));
        lds_dma16_b(pb[q], __builtin_amdgcn_readfirstlane(dst + TILE_BYTES));
        pa[q] += BK; pb[q] += BK;
    };
    // multiply tile `buf`; pf >= 0: also DMA the next tile into buffer pf, a quarter per k group, issued
    // between MFMA groups so that the M0 set-up / address traffic hides under matrix-core time
    auto multiply = [&](int buf, int pf) {
        const unsigned char* at = smem + buf * 2 * TILE_BYTES;
        const unsigned char* bt = at + TILE_BYTES;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            float4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(at + a_row + i * 32 * 128 + xo[kg]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const float4*>(bt + b_row + j * 32 * 128 + xo[kg]);
            // k-major order: consecutive MFMAs hit different accumulators (a dependent MFMA is 4 issues away).
            // Operands swapped (B fragment first): the accumulators hold the TRANSPOSED 32 x 32 tile, i.e. a lane owns
            // 4 CONSECUTIVE COLUMNS of one C row per register quad -- the epilogue stores them as float4 straight from
            // the registers
#define GVQA_MFMA_K(c_)                                                                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)             \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j].c_, af[i].c_, acc[i][j], 0, 0, 0);
            GVQA_MFMA_K(x)
            if (pf >= 0) issue2(pf, kg);
            GVQA_MFMA_K(y) GVQA_MFMA_K(z) GVQA_MFMA_K(w)
#undef GVQA_MFMA_K
        }
    };
    {
        // ONE barrier per K step: wait t (issued a whole step ago) -> barrier -> multiply t while DMAing t+1
        // into the other buffer.  The barrier serves both orders: tile t has landed for every wave, and
        // every wave has finished reading the buffer of step t-1 before anyone refills it.
        issue(0);
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            multiply(cur, t + 1 < nt ? (cur ^ 1) : -1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
    store_tile_transposed(acc, M, N, m0, n0, wr, wc, lane, ep, C, ldc);
}

// ----------------------------------------------------------------------------------------------
// Skinny streaming product C[M, N <= 32] = A[M, K] B[N, K]^T (the all-hops edge logits: [E, De] x [De, K H], 537 MB of edge features
// read once at config 3).  HBM-bound: what matters is bytes in flight per CU.  The register-staged 128 x 32 kernel keeps one 16 KiB A
// tile per workgroup on its way (three workgroups per CU: 48 KiB) and reaches 4.1 TB/s; here the A tiles go HBM -> LDS by
// `global_load_lds_dwordx4` into a FOUR-stage ring (three tiles = 48 KiB in flight per workgroup, two workgroups per CU), one barrier
// per K step, the 16 f32 MFMAs of a wave per step a small fraction of its time.  Tile format and swizzle as in k_linear_f32_dma.
__global__ __launch_bounds__(256) void k_linear_f32_skinny_dma(int M, int N, int K, const float* __restrict__ A, int64_t lda,
                                                               const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc) {
    constexpr int BM = 128, BK = 32, NS = 4;
    constexpr int A_BYTES = BM * BK * 4, B_BYTES = 32 * BK * 4, STAGE = A_BYTES + B_BYTES;       // 16 KiB + 4 KiB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // DMA duty of this wave per K step: A row groups wave, wave + 4, wave + 8, wave + 12 (8 rows x 128 bytes each) and B row group wave
    const float* pa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 4 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        pa[j] = A + (int64_t)min(m0 + row, M - 1) * lda + chunk * 4;
    }
    const float* pb;
    {
        const int row = wave * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        pb = B + (int64_t)min(row, N - 1) * ldb + chunk * 4;
    }
    // (round 6: K % 4 == 0 instead of K % 32 == 0 -- the reference's real edge width is 300.  The last step's chunks past K are fetched from the
    //  row's first chunk instead -- a mapped address -- and the A fragment is zeroed at use)
    const int nt = (K + BK - 1) / BK;
    const int my_chunk4 = ((lane & 7) ^ ((((wave * 8 + (lane >> 3))) >> 1) & 7)) * 4;       // k offset of this lane's chunk inside a step (the row groups wave + 4 j share wave's swizzle)
    auto issue = [&](int t) {                    // K step t (clamped: steps past the last re-load it into a free slot, so that the counted waits stay uniform)
        const int tt = min(t, nt - 1);
        const int64_t off = (tt * BK + my_chunk4 < K) ? (int64_t)tt * BK : -(int64_t)my_chunk4;
        const unsigned dst = lds_base + (unsigned)(t % NS) * STAGE + (unsigned)wave * 1024u;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_b(pa[j] + off, __builtin_amdgcn_readfirstlane(dst + j * 4096));
        lds_dma16_b(pb + off, __builtin_amdgcn_readfirstlane(dst + A_BYTES));
    };
    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 1) & 7;
    unsigned xo[4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) xo[kg] = (unsigned)(((kg * 2 + fh) ^ swz) * 16);
    const unsigned a_row = (unsigned)((wave * 32 + frow) * 128), b_row = (unsigned)(frow * 128);
    issue(0); issue(1); issue(2);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");         // this wave's DMAs of step t have landed (steps t + 1, t + 2: 5 each may be in flight)
        __builtin_amdgcn_s_barrier();                               // ... everybody's; everybody is done with the slot of step t - 1
        issue(t + 3);
        const unsigned char* at = smem + (t % NS) * STAGE;
        const unsigned char* bt = at + A_BYTES;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            float4 af = *reinterpret_cast<const float4*>(at + a_row + xo[kg]);
            const float4 bf = *reinterpret_cast<const float4*>(bt + b_row + xo[kg]);
            if (t * BK + (kg * 2 + fh) * 4 >= K) af = make_float4(0.f, 0.f, 0.f, 0.f);      // (the last step's chunks past K)
            // (operands swapped, B first: a lane owns 4 consecutive columns of one C row per register quad)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.x, af.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.y, af.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.z, af.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.w, af.w, acc, 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the clamped re-loads of the last steps)
    const int row = m0 + wave * 32 + frow;
    if (row < M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = 8 * q + 4 * fh;
            if (c0 < N) *reinterpret_cast<float4*>(C + (int64_t)row * ldc + c0) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

// ---- opt-in vendor backend (comparison only) -----------------------------------------------------
// The hand-written kernels are the default everywhere.  GVQA_OPT_VENDOR_GEMM (GVQA_GEMM_BACKEND=rocblas) routes PLAIN fp32
// products (no epilogue, or a plain accumulate through BLAS beta = 1) above 2 GFLOP to rocBLAS so that
// bench.py can print the vendor's number next to ours (`projection_vendor`); the library is resolved with
// dlopen at first use -- no link-time or header dependency -- and one handle is kept per device.
namespace {
// rocblas-types.h values (stable ABI constants of the C API)
enum { ROCBLAS_OP_NONE = 111, ROCBLAS_OP_TRANSPOSE = 112, ROCBLAS_DATATYPE_F32_R = 151, ROCBLAS_GEMM_ALGO_STANDARD = 0 };
typedef void* rb_handle;
typedef int (*rb_create_t)(rb_handle*);
typedef int (*rb_set_stream_t)(rb_handle, hipStream_t);
typedef int (*rb_sgemm_t)(rb_handle, int, int, int, int, int, const float*, const float*, int, const float*, int,
                          const float*, float*, int);
// rocblas_gemm_ex: D = alpha op(A) op(B) + beta C with separate C and D
typedef int (*rb_gemm_ex_t)(rb_handle, int, int, int, int, int, const void*, const void*, int, int, const void*, int, int,
                            const void*, const void*, int, int, void*, int, int, int, int, int32_t, uint32_t);
struct Vendor {
    bool tried = false;
    bool rb_ok = false;
    rb_create_t rb_create = nullptr;
    rb_set_stream_t rb_set_stream = nullptr;
    rb_sgemm_t rb_sgemm = nullptr;
    rb_gemm_ex_t rb_gemm_ex = nullptr;
    std::map<int, rb_handle> handles;      // per device
};
Vendor g_vendor;
std::mutex g_vendor_mu;

void vendor_init_locked() {
    Vendor& v = g_vendor;
    if (v.tried) return;
    v.tried = true;
    void* lib = dlopen("librocblas.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return;
    v.rb_create = reinterpret_cast<rb_create_t>(dlsym(lib, "rocblas_create_handle"));
    v.rb_set_stream = reinterpret_cast<rb_set_stream_t>(dlsym(lib, "rocblas_set_stream"));
    v.rb_sgemm = reinterpret_cast<rb_sgemm_t>(dlsym(lib, "rocblas_sgemm"));
    v.rb_gemm_ex = reinterpret_cast<rb_gemm_ex_t>(dlsym(lib, "rocblas_gemm_ex"));
    if (v.rb_create && v.rb_set_stream && v.rb_sgemm) v.rb_ok = true;
}

rb_handle vendor_handle_locked() {
    Vendor& v = g_vendor;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto it = v.handles.find(dev);
    if (it != v.handles.end()) return it->second;
    rb_handle h = nullptr;
    if (v.rb_create(&h) != 0) h = nullptr;
    v.handles.emplace(dev, h);
    return h;
}

// C_rm[M,N] = A_rm[M,K] . B_rm[N,K]^T  ==  column-major  D[N,M] = op_T(B as [K,N]) . (A as [K,M]).
// addend (optional): C = A.B^T + addend; in place (addend == C, same ld) through beta = 1, otherwise
// through rocblas_gemm_ex's separate C / D operands.
bool vendor_sgemm(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                  int64_t ldc, const float* addend, int64_t ld_add, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_vendor_mu);
    vendor_init_locked();
    Vendor& v = g_vendor;
    if (!v.rb_ok) return false;
    rb_handle h = vendor_handle_locked();
    if (!h || v.rb_set_stream(h, stream) != 0) return false;
    const float one = 1.f, zero = 0.f;
    if (addend) {
        if (addend == C && ld_add == ldc)
            return v.rb_sgemm(h, ROCBLAS_OP_TRANSPOSE, ROCBLAS_OP_NONE, (int)N, (int)M, (int)K, &one, B, (int)ldb, A, (int)lda, &one,
                              C, (int)ldc) == 0;
        if (!v.rb_gemm_ex) return false;
        return v.rb_gemm_ex(h, ROCBLAS_OP_TRANSPOSE, ROCBLAS_OP_NONE, (int)N, (int)M, (int)K, &one, B, ROCBLAS_DATATYPE_F32_R,
                            (int)ldb, A, ROCBLAS_DATATYPE_F32_R, (int)lda, &one, addend, ROCBLAS_DATATYPE_F32_R, (int)ld_add, C,
                            ROCBLAS_DATATYPE_F32_R, (int)ldc, ROCBLAS_DATATYPE_F32_R, ROCBLAS_GEMM_ALGO_STANDARD, 0, 0) == 0;
    }
    return v.rb_sgemm(h, ROCBLAS_OP_TRANSPOSE, ROCBLAS_OP_NONE, (int)N, (int)M, (int)K, &one, B, (int)ldb, A, (int)lda, &zero, C,
                      (int)ldc) == 0;
}

bool vendor_has_rocblas() {                 // opted in AND loadable (the library is only dlopen'ed once asked for)
    if (!get_option(GVQA_OPT_VENDOR_GEMM)) return false;
    std::lock_guard<std::mutex> lk(g_vendor_mu);
    vendor_init_locked();
    return g_vendor.rb_ok;
}
}  // namespace

const char* gemm_backend_name() {
    if (vendor_has_rocblas())
        return "OPT-IN vendor: rocblas for plain fp32 products >= 2 GFLOP (GVQA_OPT_VENDOR_GEMM); gvqa kernels otherwise";
    if (get_option(GVQA_OPT_PROJECTION) == GVQA_PROJECTION_F32)
        return "gvqa::k_linear_f32_dma / gvqa::k_linear_f32 (hand-written f32-input MFMA; GVQA_OPT_PROJECTION = f32)";
    if (get_option(GVQA_OPT_PROJECTION) == GVQA_PROJECTION_SPLIT3)
        return "gvqa::k_linear_split3<..., NP=3> (hand-written: fp32 as three exact bf16 pieces, six bf16-MFMA products, fp32 accumulate) "
               "for the hop projections; gvqa::k_linear_f32* (f32-input MFMA) for every other product";
    return "gvqa::k_linear_split3<..., NP=2> (hand-written: fp32 rows as two scaled fp16 pieces, three fp16-MFMA products, fp32 "
           "accumulate) for the hop projections; gvqa::k_linear_f32* (f32-input MFMA) for every other product";
}

// C[m, :] = bias (the beta = 1 operand of a vendor GEMM with a bias-only epilogue)
__global__ __launch_bounds__(256) void k_fill_bias_rows(int64_t M, int N, const float* __restrict__ bias, float* __restrict__ C,
                                                        int64_t ldc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t r = i / N;
    const int c = (int)(i - r * N);
    C[r * ldc + c] = bias[c];
}

int launch_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                  int64_t ldb, const float* bias, int relu, float* C, int64_t ldc, int batch,
                  int64_t strideA, int64_t strideB, int64_t strideC, hipStream_t stream) {
    LinearEpilogue ep{bias, nullptr, 0, nullptr, 0, relu};
    return launch_linear_ex(M, N, K, A, lda, B, ldb, ep, C, ldc, batch, strideA, strideB, strideC, stream);
}

int launch_linear_ex(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                     int64_t ldb, LinearEpilogue ep, float* C, int64_t ldc, int batch, int64_t strideA,
                     int64_t strideB, int64_t strideC, hipStream_t stream) {
    return launch_linear_t(M, N, K, A, lda, B, ldb, ep, C, ldc, batch, strideA, strideB, strideC, 0, stream);
}

// dtype_flags: bit 0 = A stored as bf16, bit 1 = C / addend / mul stored as bf16 (pointers are then
// uint16_t* passed through the float* parameters; leading dimensions in elements).
int launch_linear_t(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                    int64_t ldb, LinearEpilogue ep, float* C, int64_t ldc, int batch, int64_t strideA,
                    int64_t strideB, int64_t strideC, int dtype_flags, hipStream_t stream) {
    GVQA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 1, GVQA_E_INVALID, "linear: negative size");
    GVQA_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), GVQA_E_INVALID, "linear: size overflow");
    if (M == 0 || N == 0) return GVQA_OK;
    GVQA_REQUIRE(K > 0, GVQA_E_INVALID, "linear: K must be positive");
    if (cdiv(M, 128) > 65535) {
        // more row tiles than grid.y holds: run row chunks (rows are independent; every operand that has M rows moves along)
        GVQA_REQUIRE(batch == 1, GVQA_E_INVALID, "linear: M too large for a batched launch");
        const int64_t chunk = (int64_t)65535 * 128;
        const int64_t ea = (dtype_flags & 1) ? 2 : 4, ec = (dtype_flags & 2) ? 2 : 4;      // element bytes of A / of C, addend, mul
        auto adv = [](const float* q, int64_t elems, int64_t esz) {
            return q ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(q) + elems * esz) : q;
        };
        for (int64_t m0 = 0; m0 < M; m0 += chunk) {
            LinearEpilogue e2 = ep;
            e2.addend = adv(ep.addend, m0 * ep.ld_add, ec);
            e2.mul = adv(ep.mul, m0 * ep.ld_mul, ec);
            int rc = launch_linear_t(std::min(chunk, M - m0), N, K, adv(A, m0 * lda, ea), lda, B, ldb, e2,
                                     const_cast<float*>(adv(C, m0 * ldc, ec)), ldc, 1, 0, 0, 0, dtype_flags, stream);
            if (rc) return rc;
        }
        return GVQA_OK;
    }
    GVQA_REQUIRE(A && B && C, GVQA_E_INVALID, "linear: null operand");
    GVQA_REQUIRE(lda >= K && ldb >= K && ldc >= N, GVQA_E_INVALID, "linear: leading dimension too small");
    GVQA_REQUIRE((!ep.addend || ep.ld_add >= N) && (!ep.mul || ep.ld_mul >= N), GVQA_E_INVALID,
                 "linear: epilogue leading dimension too small");
    GVQA_REQUIRE((!ep.addend && !ep.mul) || batch == 1, GVQA_E_INVALID, "linear: addend/mul epilogue is not batched");
    // 16-byte vector loads need 16-byte aligned rows
    if (dtype_flags == 0 && !ep.bias && !ep.mul && !ep.relu && batch == 1 && N > 64 && M < (1ll << 31) &&
        2.0 * M * N * K >= 2e9 && vendor_has_rocblas()) {
        if (vendor_sgemm(M, N, K, A, lda, B, ldb, C, ldc, ep.addend, ep.ld_add, stream)) return GVQA_OK;
    }
    // bias-only epilogue on a large product: write the bias rows (one pass over C, ~4 % of the GEMM
    // time at 30 GFLOP) and accumulate onto them with the vendor kernel
    if (dtype_flags == 0 && ep.bias && !ep.addend && !ep.mul && !ep.relu && batch == 1 && N > 64 &&
        2.0 * M * N * K >= 8e9 && vendor_has_rocblas()) {
        hipLaunchKernelGGL(k_fill_bias_rows, dim3((unsigned)cdiv(M * N, 256)), dim3(256), 0, stream, M, (int)N, ep.bias, C, ldc);
        GVQA_LAUNCH_CHECK();
        if (vendor_sgemm(M, N, K, A, lda, B, ldb, C, ldc, C, ldc, stream)) return GVQA_OK;
    }
    const bool a16 = dtype_flags & 1;
    const bool vec = (K % 4 == 0) && (K >= 4) && (lda % 4 == 0) && (ldb % 4 == 0) && (strideA % 4 == 0) &&
                     (strideB % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & (a16 ? 7 : 15)) == 0) &&
                     ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    if (dtype_flags) {      // bf16-storage variants: default tile shapes only
#define GVQA_LAUNCH_T(BM_, BN_, WR_, WC_, BK_, A16_, C16_)                                                              \
        do {                                                                                                        \
            dim3 grid((unsigned)cdiv(N, BN_), (unsigned)cdiv(M, BM_), (unsigned)batch);                             \
            if (vec) hipLaunchKernelGGL((k_linear_f32<BM_, BN_, WR_, WC_, true, BK_, A16_, C16_>), grid, dim3(256), 0, stream, \
                                        (int)M, (int)N, (int)K, A, lda, B, ldb, ep, C, ldc, strideA, strideB, strideC);  \
            else hipLaunchKernelGGL((k_linear_f32<BM_, BN_, WR_, WC_, false, BK_, A16_, C16_>), grid, dim3(256), 0, stream, \
                                    (int)M, (int)N, (int)K, A, lda, B, ldb, ep, C, ldc, strideA, strideB, strideC);     \
        } while (0)
#define GVQA_LAUNCH_T_SHAPE(A16_, C16_)                                                  \
        do {                                                                             \
            if (N <= 32) GVQA_LAUNCH_T(128, 32, 4, 1, 32, A16_, C16_);                   \
            else if (N <= 64) GVQA_LAUNCH_T(128, 64, 2, 2, 32, A16_, C16_);              \
            else GVQA_LAUNCH_T(128, 128, 2, 2, 16, A16_, C16_);                          \
        } while (0)
        if (dtype_flags == 1) GVQA_LAUNCH_T_SHAPE(true, false);
        else if (dtype_flags == 2) GVQA_LAUNCH_T_SHAPE(false, true);
        else GVQA_LAUNCH_T_SHAPE(true, true);
#undef GVQA_LAUNCH_T_SHAPE
#undef GVQA_LAUNCH_T
        GVQA_LAUNCH_CHECK();
        return GVQA_OK;
    }
#define GVQA_LAUNCH_LINEAR(BM_, BN_, WR_, WC_)                                                          \
    do {                                                                                               \
        dim3 grid((unsigned)cdiv(N, BN_), (unsigned)cdiv(M, BM_), (unsigned)batch);                    \
        if (vec)                                                                                       \
            hipLaunchKernelGGL((k_linear_f32<BM_, BN_, WR_, WC_, true>), grid, dim3(64 * WR_ * WC_), 0, stream, \
                               (int)M, (int)N, (int)K, A, lda, B, ldb, ep, C, ldc, strideA,            \
                               strideB, strideC);                                                      \
        else                                                                                           \
            hipLaunchKernelGGL((k_linear_f32<BM_, BN_, WR_, WC_, false>), grid, dim3(64 * WR_ * WC_), 0, stream, \
                               (int)M, (int)N, (int)K, A, lda, B, ldb, ep, C, ldc, strideA,            \
                               strideB, strideC);                                                      \
    } while (0)
    static const int tile_sel = []() { const char* v = getenv("GVQA_GEMM_TILE"); return v ? atoi(v) : 0; }();
    // the LDS-DMA kernel stores C (and reads bias / addend / mul) as float4: whole 4-column quads, 16-byte aligned rows
    auto al16 = [&](const void* q, int64_t ld) { return !q || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld % 4 == 0); };
    const bool ep4_ok = N % 4 == 0 && al16(C, ldc) && al16(ep.addend, ep.ld_add) && al16(ep.mul, ep.ld_mul) && al16(ep.bias, 4);
    // tall skinny streams (the all-hops edge logits): the four-stage LDS-DMA ring keeps three A tiles per workgroup in flight
    if (N <= 32 && N % 4 == 0 && K % 4 == 0 && K >= 96 && vec && batch == 1 && M >= 16384 && !ep.bias && !ep.addend && !ep.mul && !ep.relu &&
        ep4_ok && tile_sel == 0) {
        hipLaunchKernelGGL(k_linear_f32_skinny_dma, dim3((unsigned)cdiv(M, 128)), dim3(256), 0, stream, (int)M, (int)N, (int)K, A, lda, B, ldb, C, ldc);
    }
    else if (N <= 32) GVQA_LAUNCH_LINEAR(128, 32, 4, 1);
    else if (N <= 64) GVQA_LAUNCH_LINEAR(128, 64, 2, 2);
    // per-graph products (M = graphs): a 128 x 128 tile is >= 27 us of MFMA issue for its 4 waves however few
    // tiles there are -- when they cannot fill the chip, quarter tiles put 4x the CUs to work
    // (measured on the per-graph instruction terms, 5 x [2048 x 516 x 512]: 400 full tiles over 768 resident slots 91 us)
    else if (tile_sel == 0 && cdiv(M, 128) * cdiv(N, 128) * batch < 512) GVQA_LAUNCH_LINEAR(64, 64, 2, 2);
    else if (tile_sel == 1 && M >= 256) GVQA_LAUNCH_LINEAR(256, 128, 4, 2);
    else if (tile_sel == 2 && M >= 256) GVQA_LAUNCH_LINEAR(256, 128, 2, 2);
    else if (tile_sel == 3) GVQA_LAUNCH_LINEAR(128, 128, 4, 2);
    // LDS-DMA staging: the default for chip-filling products with whole K steps (+10 % over the register-staged
    // kernel at the config-3 projection); GVQA_GEMM_TILE=5/6 force the register-staged kernels
    else if ((tile_sel == 7 || (tile_sel == 0 && cdiv(M, 128) * cdiv(N, 128) >= 256)) && batch == 1 &&
             K % 32 == 0 && vec && ep4_ok) {
        dim3 grid((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 128));
        hipLaunchKernelGGL(k_linear_f32_dma, grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, A, lda, B, ldb, ep, C, ldc);
    }
    // short K: K step 16, 3 blocks/CU (+4.5 % at K = 512) -- unless the grid cannot fill the chip anyway: a lone
    // block per CU is bound by the latency of its serial K steps, and K step 32 halves their number
    else if (tile_sel == 5 || (tile_sel == 0 && K <= 1024 && cdiv(M, 128) * cdiv(N, 128) * batch >= 256)) {
        dim3 grid((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 128), (unsigned)batch);
        if (vec) hipLaunchKernelGGL((k_linear_f32<128, 128, 2, 2, true, 16>), grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, A, lda, B, ldb, ep, C, ldc, strideA, strideB, strideC);
        else hipLaunchKernelGGL((k_linear_f32<128, 128, 2, 2, false, 16>), grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, A, lda, B, ldb, ep, C, ldc, strideA, strideB, strideC);
    }
    else GVQA_LAUNCH_LINEAR(128, 128, 2, 2);    // also tile_sel == 6: force K step 32
#undef GVQA_LAUNCH_LINEAR
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // namespace gvqa

extern "C" const char* gvqa_gemm_backend(void) { return gvqa::gemm_backend_name(); }

extern "C" int gvqa_linear_f32_ex(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                                  int64_t ldb, const float* bias, const float* addend, int64_t ld_add,
                                  const float* mul, int64_t ld_mul, int relu, float* C, int64_t ldc, void* stream) {
    gvqa::LinearEpilogue ep{bias, addend, ld_add, mul, ld_mul, relu};
    return gvqa::launch_linear_ex(M, N, K, A, lda, B, ldb, ep, C, ldc, 1, 0, 0, 0, static_cast<hipStream_t>(stream));
}

extern "C" int gvqa_linear_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                               int64_t ldb, const float* bias, int relu, float* C, int64_t ldc, void* stream) {
    return gvqa::launch_linear(M, N, K, A, lda, B, ldb, bias, relu, C, ldc, 1, 0, 0, 0,
                               static_cast<hipStream_t>(stream));
}
