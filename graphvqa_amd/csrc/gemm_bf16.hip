// Dense projections of bf16 node tensors on the CDNA4 bf16 matrix cores (LCGN bf16-node-feature mode,
// BASELINE config 5):   C[M,N] = sum_{p < P} A[M,K] . W_p[N,K]^T  (+bias) (+addend) (*mul) (act)
//
// A is a node tensor stored as bf16.  The fp32 weights are packed once per forward into P bf16
// pieces side by side, Wpk[N, P*K]:  P = 2 keeps W = W_hi + W_lo to 16 significant bits (the product
// with a bf16 operand is then as accurate as the bf16 storage of the node tensors allows -- the
// products a*w_hi and a*w_lo are exact in fp32 and the accumulation is fp32), P = 1 is the plain
// "bf16 weights" arithmetic.  `v_mfma_f32_32x32x16_bf16` runs at 16x the f32-input MFMA rate, so even
// the two-piece product is 8x cheaper in matrix-core time than the fp32 MFMA it replaces.
//
// Tiling (64-wide wavefronts): block = 4 waves (2 x 2), block tile 128 x 128, K step 64 (128 bytes
// per row), both operands K-contiguous, staged global -> registers -> LDS with 16-byte accesses,
// register double-buffered (next tile's loads fly under this tile's MFMAs, one barrier per K step).
// LDS rows are padded to 144 bytes = 36 dwords: the same bank geometry as the fp32 kernel, a wave's
// ds_read_b128 fragment read (32 rows x 2 k-halves) is conflict-free.  Fragment: lane (r = lane & 31,
// h = lane >> 5) feeds the 8 consecutive k's [16 kg + 8 h, +8) of row r to one MFMA from ONE b128 read.
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include "common.h"
#include "gemm_tile.h"

namespace gvqa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN, int WR, int WC, bool C16>
__global__ __launch_bounds__(64 * WR * WC) void k_linear_bf16(int M, int N, int K, int P, const uint16_t* __restrict__ A,
                                                              int64_t lda, const uint16_t* __restrict__ B, int64_t ldb,
                                                              LinearEpilogue ep, void* C_, int64_t ldc, int vec_ep) {
    constexpr int BK = 64;                        // bf16 per K step
    constexpr int LDS_LD = BK + 8;                // 144-byte rows
    constexpr int RQ = BK / 8;                    // 16-byte chunks per tile row
    constexpr int NTH = 64 * WR * WC;
    constexpr int WM = BM / WR, WN = BN / WC;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_V = BM * RQ / NTH, B_V = BN * RQ / NTH;
    static_assert(BM * RQ % NTH == 0 && BN * RQ % NTH == 0, "tiles must divide evenly");
    typedef typename std::conditional<C16, uint16_t, float>::type TC;
    TC* C = static_cast<TC*>(C_);

    constexpr int ST_LD = BN + 4;                 // fp32 row stride of the epilogue staging tile
    constexpr int OPER_BYTES = 2 * (BM + BN) * LDS_LD * 2, STAGE_BYTES = BM * ST_LD * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[OPER_BYTES > STAGE_BYTES ? OPER_BYTES : STAGE_BYTES];
    uint16_t (*As)[BM * LDS_LD] = reinterpret_cast<uint16_t (*)[BM * LDS_LD]>(smem);
    uint16_t (*Bs)[BN * LDS_LD] = reinterpret_cast<uint16_t (*)[BN * LDS_LD]>(smem + 2 * BM * LDS_LD * 2);

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

    // branch-free clamped loads (rows past the edge re-read a valid row and are never stored; the K
    // tail is zeroed when the registers are written to LDS), row pointers formed once
    const uint16_t* pa[A_V];
    const uint16_t* pb[B_V];
    const int c8 = (tid % RQ) * 8;                // k offset of this thread's chunk (same for all its rows: NTH % RQ == 0)
#pragma unroll
    for (int i = 0; i < A_V; ++i) pa[i] = A + (int64_t)min(m0 + (tid + i * NTH) / RQ, M - 1) * lda + c8;
#pragma unroll
    for (int i = 0; i < B_V; ++i) pb[i] = B + (int64_t)min(n0 + (tid + i * NTH) / RQ, N - 1) * ldb + c8;

    uint4 ra[A_V], rb[B_V];
    const int nk = (K + BK - 1) / BK, nt = nk * P;
    auto load_tile = [&](int kk, int p) {
        int k = kk * BK;
        if (k + c8 + 8 > K) k = K - 8 - c8;       // K % 8 == 0: the last whole chunk (masked at the LDS store)
#pragma unroll
        for (int i = 0; i < A_V; ++i) ra[i] = *reinterpret_cast<const uint4*>(pa[i] + k);
#pragma unroll
        for (int i = 0; i < B_V; ++i) rb[i] = *reinterpret_cast<const uint4*>(pb[i] + (int64_t)p * K + k);
    };
    auto store_tile = [&](int buf, int kk) {
        const unsigned keep = kk * BK + c8 >= K ? 0u : ~0u;   // chunk entirely past K (only in the last K tile): zeros
#pragma unroll
        for (int i = 0; i < A_V; ++i)
            *reinterpret_cast<uint4*>(&As[buf][((tid + i * NTH) / RQ) * LDS_LD + c8]) =
                make_uint4(ra[i].x & keep, ra[i].y & keep, ra[i].z & keep, ra[i].w & keep);
#pragma unroll
        for (int i = 0; i < B_V; ++i)
            *reinterpret_cast<uint4*>(&Bs[buf][((tid + i * NTH) / RQ) * LDS_LD + c8]) =
                make_uint4(rb[i].x & keep, rb[i].y & keep, rb[i].z & keep, rb[i].w & keep);
    };

    load_tile(0, 0);
    store_tile(0, 0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    int kk = 0, p = 0;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        int kk_n = kk + 1, p_n = p;
        if (kk_n == nk) { kk_n = 0; ++p_n; }
        if (t + 1 < nt) load_tile(kk_n, p_n);
        const uint16_t* as = &As[cur][(wr * WM + frow) * LDS_LD + fk];
        const uint16_t* bs = &Bs[cur][(wc * WN + frow) * LDS_LD + fk];
#pragma unroll
        for (int kg = 0; kg < BK / 16; ++kg) {
            bf16x8 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(as + i * 32 * LDS_LD + kg * 16));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bf[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bs + j * 32 * LDS_LD + kg * 16));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) store_tile(cur ^ 1, kk_n);
        __syncthreads();
        kk = kk_n; p = p_n;
    }

    tile_epilogue<BM, BN, WR, WC, C16>(acc, smem, M, N, m0, n0, ep, C, ldc, vec_ep);
}

// ---- LDS-DMA variant (K % 64 == 0) ------------------------------------------------------------
// Same tile and MFMA schedule, but the operand tiles go HBM/L2 -> LDS with `global_load_lds_dwordx4`
// (no VGPR staging, no ds_write): a wave instruction deposits its 64 lanes' 16-byte chunks at 64
// consecutive LDS slots, i.e. 8 unpadded 128-byte tile rows.  Unpadded rows would put the 32 rows of a
// fragment read on two 16-byte columns of the 256-byte bank row, so the tile is XOR-swizzled: slot
// (row r, position q) holds k-chunk q ^ ((r >> 1) & 7) -- the lane picks its GLOBAL chunk accordingly
// (still the same 128-byte line per 8 lanes) and the fragment read applies the same XOR.  The DMA is
// issued from inline asm (invisible to the compiler's LDS dependence tracking) and ordered by counted
// `s_waitcnt vmcnt` + barriers: tile t+1 is in flight while tile t is multiplied.
// Epilogue from TRANSPOSED accumulators (MFMA operands swapped: a lane owns 4 consecutive columns of one C row per
// register quad): 16 stores per lane of 8 bytes (bf16 C) / 16 bytes (fp32 C), addend / mul read the same way; no LDS
// round trip, no barrier.  Needs N % 4 == 0 and 8- / 16-byte aligned rows.
template <bool C16, typename TC, int MT = 2, int NT = 2>
__device__ __forceinline__ void store_tile_transposed_b(f32x16 (&acc)[MT][NT], int M, int N, int m0, int n0, int wr, int wc, int lane,
                                                        const LinearEpilogue& ep, TC* C, int64_t ldc) {
    const int mrow = lane & 31, ncol0 = 4 * (lane >> 5);
    auto load4 = [&](const float* base, int64_t elem, float (&o)[4]) {
        if constexpr (C16) {
            const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem);
            o[0] = __uint_as_float(raw.x << 16); o[1] = __uint_as_float(raw.x & 0xFFFF0000u);
            o[2] = __uint_as_float(raw.y << 16); o[3] = __uint_as_float(raw.y & 0xFFFF0000u);
        } else {
            const float4 t = *reinterpret_cast<const float4*>(base + elem);
            o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
        }
    };
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int gr = m0 + wr * (MT * 32) + i * 32 + mrow;
        if (gr >= M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gc = n0 + wc * (NT * 32) + j * 32 + 8 * q + ncol0;
                if (gc >= N) continue;                 // N % 4 == 0: a quad is entirely inside or outside
                float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                if (ep.bias) {
                    const float4 b4 = *reinterpret_cast<const float4*>(ep.bias + gc);
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                }
                if (ep.addend) {
                    float a4[4];
                    load4(ep.addend, (int64_t)gr * ep.ld_add + gc, a4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += a4[e];
                }
                if (ep.mul) {
                    float m4[4];
                    load4(ep.mul, (int64_t)gr * ep.ld_mul + gc, m4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= m4[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ep.relu == 1) v[e] = fmaxf(v[e], 0.f);
                    else if (ep.relu == 2) v[e] = v[e] > 0.f ? v[e] : expf(v[e]) - 1.f;
                }
                if constexpr (C16) {
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2*>(C + (int64_t)gr * ldc + gc) = o;
                } else {
                    *reinterpret_cast<float4*>(C + (int64_t)gr * ldc + gc) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
    }
}

template <bool C16>
__global__ __launch_bounds__(256) void k_linear_bf16_dma(int M, int N, int K, int P, const uint16_t* __restrict__ A, int64_t lda,
                                                         const uint16_t* __restrict__ B, int64_t ldb, LinearEpilogue ep,
                                                         void* C_, int64_t ldc) {
    constexpr int BM = 128, BN = 128, WC = 2, BK = 64;
    constexpr int TILE_BYTES = 128 * BK * 2;                  // one operand tile: 16 KiB
    typedef typename std::conditional<C16, uint16_t, float>::type TC;
    TC* C = static_cast<TC*>(C_);
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE_BYTES];      // {A, B} x 2 buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA j of wave w fills tile rows [(4 j + w) 8, +8): lane -> (row, slot position q), global chunk q ^ swizzle(row)
    const uint16_t* pa[4];
    const uint16_t* pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 4 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        pa[j] = A + (int64_t)min(m0 + row, M - 1) * lda + chunk * 8;
        pb[j] = B + (int64_t)min(n0 + row, N - 1) * ldb + chunk * 8;
    }
    const int nk = K / BK, nt = nk * P;
    int a_wrap = nk;                                           // tiles until the A k-offset wraps back to 0 (next weight piece)
    auto issue = [&](int buf) {
        const unsigned dst = lds_base + buf * 2 * TILE_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_b(pa[j], __builtin_amdgcn_readfirstlane(dst + j * 4096));
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_b(pb[j], __builtin_amdgcn_readfirstlane(dst + TILE_BYTES + j * 4096));
        const int adv = (--a_wrap == 0) ? BK - K : BK;         // B runs on through the pieces, A starts over
        if (a_wrap == 0) a_wrap = nk;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa[j] += adv; pb[j] += BK; }
    };

    // fragment read offsets: row R = w-tile base + i*32 + (lane & 31); swizzle((R >> 1) & 7) only depends on the lane
    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 1) & 7;
    unsigned xo[4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) xo[kg] = (unsigned)(((kg * 2 + fh) ^ swz) * 16);
    const unsigned a_row = (unsigned)((wr * 64 + frow) * 128), b_row = (unsigned)((wc * 64 + frow) * 128);

    int a_adv = BK;
    auto issue2 = [&](int buf, int q) {          // quarter q of a tile's DMAs (A rows and B rows of one 32-row group)
        const unsigned dst = lds_base + buf * 2 * TILE_BYTES + wave * 1024 + q * 4096;
        lds_dma16_b(pa[q], __builtin_amdgcn_readfirstlane(dst));
        lds_dma16_b(pb[q], __builtin_amdgcn_readfirstlane(dst + TILE_BYTES));
        pb[q] += BK;
        pa[q] += a_adv;
    };
    // ONE barrier per K step: wait t (issued a whole step ago) -> barrier -> multiply t while DMAing t+1 into
    // the other buffer, a quarter per k group between the MFMAs.  The barrier orders both hazards: tile t has
    // landed for every wave, and every wave is done reading the buffer of step t-1 before anyone refills it.
    issue(0);
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nt;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (more) {                                            // A k-offset of the tile after next: wraps at a piece boundary
            a_adv = (--a_wrap == 0) ? BK - K : BK;
            if (a_wrap == 0) a_wrap = nk;
        }
        const unsigned char* at = smem + cur * 2 * TILE_BYTES;
        const unsigned char* bt = at + TILE_BYTES;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(at + a_row + i * 32 * 128 + xo[kg]));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bt + b_row + j * 32 * 128 + xo[kg]));
            // operands swapped (B fragment first): transposed accumulators, see store_tile_transposed_b
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0], af[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[1], af[0], acc[0][1], 0, 0, 0);
            if (more) issue2(cur ^ 1, kg);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0], af[1], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[1], af[1], acc[1][1], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    store_tile_transposed_b<C16>(acc, M, N, m0, n0, wr, wc, lane, ep, C, ldc);
}

// ---- wide LDS-DMA variant: 256 x 128 tile, 8 waves, ALL weight pieces of a K step resident together -----------------
// The 128 x 128 kernel above walks the weight pieces as extra K range and re-loads the A tile for each: 15.6 bytes of
// L2 -> LDS traffic per kFLOP, i.e. the whole 64 B/clk/CU of the L1 path at full MFMA rate -- it is staging-bound at 780 TF.
// Here a K step brings the A tile ONCE (256 rows x 128 B) and the P piece tiles of W (P x 128 rows x 128 B); a wave (4 x 2
// layout, 64 x 64 wave tile) reads 2 A + 2 P B fragments per 16-deep sub-step for 4 P MFMAs: 7.6 B/kFLOP at P = 2.  Same
// XOR swizzle through the lanes' global addresses, same one-barrier-per-K-step ring of two LDS buffers, DMAs of the next step
// issued between the MFMA groups, transposed accumulators -> register epilogue.
template <bool C16, int P>
__global__ __launch_bounds__(512) void k_linear_bf16_wide(int M, int N, int K, const uint16_t* __restrict__ A, int64_t lda,
                                                          const uint16_t* __restrict__ B, int64_t ldb, LinearEpilogue ep,
                                                          void* C_, int64_t ldc) {
    constexpr int BM = 256, BN = 128, BK = 64;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;              // 32 KiB, 16 KiB per piece
    constexpr int STAGE = A_BYTES + P * B_BYTES;
    constexpr int NDMA = STAGE / 1024 / 8;                                   // DMA instructions per wave per K step
    typedef typename std::conditional<C16, uint16_t, float>::type TC;
    TC* C = static_cast<TC*>(C_);
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // DMA d of wave w covers stage rows-of-8 unit u = d * 8 + w: units [0, 32) are A rows, then 16 units per weight piece
    const uint16_t* src[NDMA];
    unsigned dst[NDMA];
#pragma unroll
    for (int d = 0; d < NDMA; ++d) {
        const int u = d * 8 + wave;
        const bool isA = u < BM / 8;
        const int ub = isA ? u : (u - BM / 8) % (BN / 8), piece = isA ? 0 : (u - BM / 8) / (BN / 8);
        const int row = ub * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        src[d] = isA ? A + (int64_t)min(m0 + row, M - 1) * lda + chunk * 8
                     : B + (int64_t)min(n0 + row, N - 1) * ldb + (int64_t)piece * K + chunk * 8;
        dst[d] = lds_base + (unsigned)u * 1024u;
    }
    auto issue_one = [&](int buf, int d) {
        lds_dma16_b(src[d], __builtin_amdgcn_readfirstlane(dst[d] + buf * STAGE));
        src[d] += BK;
    };

    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 1) & 7;
    unsigned xo[4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) xo[kg] = (unsigned)(((kg * 2 + fh) ^ swz) * 16);
    const unsigned a_row = (unsigned)((wr * 64 + frow) * 128), b_row = (unsigned)(A_BYTES + (wc * 64 + frow) * 128);

    const int nt = K / BK;
#pragma unroll
    for (int d = 0; d < NDMA; ++d) issue_one(0, d);
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nt;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* st = smem + cur * STAGE;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            bf16x8 af[2], bf[P][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + a_row + i * 32 * 128 + xo[kg]));
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bf[p][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + b_row + p * B_BYTES + j * 32 * 128 + xo[kg]));
            // operands swapped (B fragment first): transposed accumulators, see store_tile_transposed_b
#pragma unroll
            for (int p = P - 1; p >= 0; --p) {                    // low-order piece first
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[p][j], af[i], acc[i][j], 0, 0, 0);
                if (more) {                                       // the next step's DMAs, spread evenly over this step's 4 P MFMA groups
#pragma unroll
                    for (int d = 0; d < NDMA; ++d)
#ifndef GVQA_BF16_DMA_SPREAD
#define GVQA_BF16_DMA_SPREAD 4            /* MFMA groups of the step (of 4 P) that the next step's DMAs are issued between.  Round 5 spread them over all 4 P: the last ones
                                             then have a fraction of a step to land before the vmcnt(0) at the next barrier.  Within the first half: LCGN bf16 forward
                                             2.355 -> 2.25 ms (same box, alternating; 2: 2.30) */
#endif
                        if (d * (GVQA_BF16_DMA_SPREAD) / NDMA == kg * P + (P - 1 - p)) issue_one(cur ^ 1, d);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    store_tile_transposed_b<C16>(acc, M, N, m0, n0, wr, wc, lane, ep, C, ldc);
}

// ---- 256 x 256 tile, two-piece weights, three-stage ring (round 6) ----------------------------------------------------
// Why another tile: in the 256 x 128 kernel above a wave (64 x 64 of C, two weight pieces) reads 2 A + 4 B fragments from LDS for 8
// MFMAs -- per K step the CU's LDS moves 48 KiB of fragment reads + 16 KiB of DMA writes per 512 matrix-core clocks and SIMD: at
// 128 B/clk that IS 512 clocks, the LDS pipe saturates exactly where the matrix pipe does, and every bank conflict or issue bubble
// is lost MFMA time (0.66-0.71 PF/s issued on the LCGN shapes).  Here a wave owns 128 x 64 of C (4 x 2 tiles, 128 accumulator
// registers): 4 A + 4 B fragments for 16 MFMAs, LDS traffic 0.69 of the matrix-core time; and a [29785, 512] product is 234
// workgroups -- ONE round of the 256 CUs instead of two rounds of 468 half-size ones.  K step 32 (64-byte rows: a DMA instruction
// deposits 16 rows; slot (row r, position q) holds k-chunk q ^ ((r >> 2) & 3), so that the 16 rows of a fragment-read phase fall
// on 64 distinct banks), stage = 16 KiB of A + 2 x 16 KiB of weights, three stages (144 KiB), DMAs two steps ahead behind a
// counted wait, one barrier per step.
#ifndef GVQA_BIG_LOADERS
#define GVQA_BIG_LOADERS 0  /* 4: the DMAs are issued by four extra waves (one per SIMD) that do nothing else; the eight MFMA waves issue none */
#endif
#ifndef GVQA_BIG_ABL
#define GVQA_BIG_ABL 0      /* measurement builds (scripts/r06_big_ablation.sh): 1 no DMAs in the loop, 2 fragments read once, 4 no barrier, 8 no counted wait */
#endif
// (Measured and dropped: per-lane 64-bit source pointers advanced every step instead of a scalar base + constant lane offset -- 0.966 vs 1.006 PF/s
//  issued on [29785, 1536] x K 1024; the two waves of a SIMD issuing their DMAs in different halves of the step -- no change.  What the step spends
//  outside its MFMAs, same product, `profiles/r06_bf16_big_ablation.jsonl`: this kernel with MFMAs only 1.244 PF/s (tile rounds, prologue and epilogue included; the bare
//  MFMA stream of gvqa_mfma_stream below: 1.84 on random bf16 operands, 2.47 on zeros), without the DMAs 1.107, as shipped 1.006.  GVQA_BIG_LOADERS = 4 -- the DMAs issued by four extra waves, one per SIMD, that do
//  nothing else (162 registers, twelve waves per CU) -- 1.066-1.070 against 1.055 on the same box, nothing on the smaller products, the LCGN
//  forward 2.131 -> 2.118 ms (`r06_bf16_big_loaders_ab.jsonl`): what the DMAs cost is not issue slots of the MFMA waves but the data movement
//  itself.  Left as a build switch, off.)
template <bool C16>
__global__ __launch_bounds__(512 + 64 * GVQA_BIG_LOADERS) void k_linear_bf16_big(int M, int N, int K, const uint16_t* __restrict__ A, int64_t lda,
                                                         const uint16_t* __restrict__ B, int64_t ldb, LinearEpilogue ep,
                                                         void* C_, int64_t ldc) {
    constexpr int BM = 256, BN = 256, BK = 32, P = 2, NST = 3;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;              // 16 KiB each
    constexpr int STAGE = A_BYTES + P * B_BYTES;                             // 48 KiB
    constexpr int LDW = GVQA_BIG_LOADERS;                                    // loader waves (0: every MFMA wave issues its share)
    constexpr int NIW = LDW > 0 ? LDW : 8;                                   // waves that issue DMAs
    constexpr int NDMA = STAGE / 1024 / NIW;                                 // DMA instructions per issuing wave per K step (6, or 12 per loader)
    typedef typename std::conditional<C16, uint16_t, float>::type TC;
    TC* C = static_cast<TC*>(C_);
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = (wave >> 2) & 1, wc = wave & 3;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;
    const bool loader = LDW > 0 && wave >= 8;
    const int iw = LDW > 0 ? (wave & (NIW - 1)) : wave;                      // index among the issuing waves

    // DMA d of issuing wave w covers stage unit u = d * NIW + w (16 rows of 64 bytes): units [0, 16) are A rows, then 16 units per weight
    // piece.  Scalar base (the workgroup's first row at the step's K offset) + a 32-bit lane offset that never changes: no per-lane
    // pointer arithmetic in the loop.
    constexpr int DA = (BM / 16) / NIW;                                      // this wave's first DA units are A rows
    unsigned voff[NDMA];
#pragma unroll
    for (int d = 0; d < NDMA; ++d) {
        const int u = d * NIW + iw;
        const bool isA = d < DA;
        const int ub = isA ? u : (u - BM / 16) % (BN / 16), piece = isA ? 0 : (u - BM / 16) / (BN / 16);
        const int row = ub * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        voff[d] = isA ? (unsigned)(min(m0 + row, M - 1) - m0) * (unsigned)lda * 2u + (unsigned)chunk * 16u
                      : (unsigned)(min(n0 + row, N - 1) - n0) * (unsigned)ldb * 2u + (unsigned)piece * (unsigned)K * 2u + (unsigned)chunk * 16u;
    }
    const char* const a_u = reinterpret_cast<const char*>(A) + (int64_t)m0 * lda * 2;
    const char* const b_u = reinterpret_cast<const char*>(B) + (int64_t)n0 * ldb * 2;
    const unsigned sdst0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)iw * 1024u);
    auto issue_one = [&](unsigned stage_off, int d, int step) {
        const char* sb = (d < DA ? a_u : b_u) + (int64_t)step * (BK * 2);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(voff[d]), "s"(sb), "s"(sdst0 + (unsigned)d * (unsigned)(NIW * 1024) + stage_off)
                     : "memory", "m0");
    };
    const int nt = K / BK;
    if (LDW == 0 || loader) {
#pragma unroll
        for (int d = 0; d < NDMA; ++d) issue_one(0u, d, 0);
        if (nt > 1) {
#pragma unroll
            for (int d = 0; d < NDMA; ++d) issue_one((unsigned)STAGE, d, 1);
        }
    }
    unsigned cur_off = 0u, ld_off = 2u * STAGE;                // stage of the current step / of the step two ahead
    if constexpr (LDW > 0) {
        if (loader) {
            // loader wave: wait for step t (its DMAs are older than step t + 1's), meet the MFMA waves at the barrier -- which also says they are
            // done with stage (t - 1) % 3 --, refill that stage with step t + 2
            for (int t = 0; t < nt; ++t) {
                if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (t + 2 < nt) {
#pragma unroll
                    for (int d = 0; d < NDMA; ++d) issue_one(ld_off, d, t + 2);
                }
                ld_off = ld_off == (unsigned)(NST - 1) * STAGE ? 0u : ld_off + STAGE;
            }
            return;
        }
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 2) & 3;
    unsigned xo[2];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) xo[kg] = (unsigned)(((kg * 2 + fh) ^ swz) * 16);
    const unsigned a_row = (unsigned)((wr * 128 + frow) * 64), b_row = (unsigned)(A_BYTES + (wc * 64 + frow) * 64);

    [[maybe_unused]] bf16x8 abl_af[2][4], abl_bf[2][P][2];     // (measurement builds)
    for (int t = 0; t < nt; ++t) {
        const bool more = t + 2 < nt;
        // step t's DMAs are older than step t + 1's (the only ones that may still fly): counted wait, then the barrier that also
        // says every wave is done reading stage (t - 1) % 3 -- the one refilled below
        if constexpr (!(GVQA_BIG_ABL & 8) && LDW == 0) {
            if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (!(GVQA_BIG_ABL & 4)) __builtin_amdgcn_s_barrier();
        const unsigned char* st = smem + cur_off;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            bf16x8 af[4], bf[P][2];
            if ((GVQA_BIG_ABL & 2) && t > 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = abl_af[kg][i];
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int j = 0; j < 2; ++j) bf[p][j] = abl_bf[kg][p][j];
            } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + a_row + i * 32 * 64 + xo[kg]));
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bf[p][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + b_row + p * B_BYTES + j * 32 * 64 + xo[kg]));
            if constexpr ((GVQA_BIG_ABL & 2) != 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) abl_af[kg][i] = af[i];
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int j = 0; j < 2; ++j) abl_bf[kg][p][j] = bf[p][j];
            }
            }
            // operands swapped (B fragment first): transposed accumulators, see store_tile_transposed_b
#pragma unroll
            for (int p = P - 1; p >= 0; --p) {                    // low-order piece first
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[p][j], af[i], acc[i][j], 0, 0, 0);
                if (LDW == 0 && more && !(GVQA_BIG_ABL & 1)) {    // the DMAs of the step two ahead, spread over this step's four MFMA groups
#pragma unroll
                    for (int d = 0; d < NDMA; ++d)
                        if (d * 4 / NDMA == kg * P + (P - 1 - p)) issue_one(ld_off, d, t + 2);
                }
            }
        }
        cur_off = cur_off == (unsigned)(NST - 1) * STAGE ? 0u : cur_off + STAGE;
        ld_off = ld_off == (unsigned)(NST - 1) * STAGE ? 0u : ld_off + STAGE;
    }
    store_tile_transposed_b<C16, TC, 4, 2>(acc, M, N, m0, n0, wr, wc, lane, ep, C, ldc);
}

// Wpk[n, p*Kp + k] = p-th bf16 piece of W[n, k] (piece 0 = round-to-nearest bf16 of w, piece 1 = bf16 of the
// remainder), zero for K <= k < Kp (K padded up to the GEMM's multiple of 8)
__global__ __launch_bounds__(256) void k_pack_weight_bf16(int64_t rows, int K, int Kp, int P, const float* __restrict__ W,
                                                          int64_t ldw, uint16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Kp) return;
    const int64_t r = i / Kp;
    const int k = (int)(i - r * Kp);
    const float w = k < K ? W[r * ldw + k] : 0.f;
    const uint16_t hi = f32_to_bf16(w);
    uint16_t* o = out + r * (int64_t)P * Kp + k;
    o[0] = hi;
    if (P > 1) o[Kp] = f32_to_bf16(w - bf16_to_f32(hi));
}

int launch_pack_weight_bf16(int64_t rows, int K, int Kp, int P, const float* W, int64_t ldw, void* out, hipStream_t stream) {
    GVQA_REQUIRE(P == 1 || P == 2, GVQA_E_INVALID, "pack_weight_bf16: pieces must be 1 or 2");
    GVQA_REQUIRE(Kp >= K, GVQA_E_INVALID, "pack_weight_bf16: padded K smaller than K");
    if (rows == 0 || Kp == 0) return GVQA_OK;
    hipLaunchKernelGGL(k_pack_weight_bf16, dim3((unsigned)cdiv(rows * Kp, 256)), dim3(256), 0, stream, rows, K, Kp, P, W, ldw,
                       static_cast<uint16_t*>(out));
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

bool linear_bf16_supported(int64_t K, int64_t lda, const void* A, const void* Wpk) {
    return K >= 8 && K % 8 == 0 && lda % 8 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(Wpk) & 15) == 0;
}

// A: bf16 [M, lda]; Wpk: packed bf16 [N, P*K]; C and the addend / mul operands are bf16 when c16, fp32 otherwise.
int launch_linear_bf16(int64_t M, int64_t N, int64_t K, int P, const void* A, int64_t lda, const void* Wpk, LinearEpilogue ep,
                       void* C, int64_t ldc, bool c16, hipStream_t stream) {
    GVQA_REQUIRE(M >= 0 && N >= 0 && K > 0 && (P == 1 || P == 2), GVQA_E_INVALID, "linear_bf16: bad size");
    GVQA_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 30), GVQA_E_INVALID, "linear_bf16: size overflow");
    if (M == 0 || N == 0) return GVQA_OK;
    GVQA_REQUIRE(A && Wpk && C, GVQA_E_INVALID, "linear_bf16: null operand");
    GVQA_REQUIRE(linear_bf16_supported(K, lda, A, Wpk), GVQA_E_INVALID, "linear_bf16: K / lda must be multiples of 8, operands 16-byte aligned");
    GVQA_REQUIRE(lda >= K && ldc >= N, GVQA_E_INVALID, "linear_bf16: bad leading dimension");
    if (cdiv(M, 128) > 65535) {          // more row tiles than grid.y holds: row chunks
        const int64_t chunk = (int64_t)65535 * 128, ec = c16 ? 2 : 4;
        for (int64_t m0 = 0; m0 < M; m0 += chunk) {
            LinearEpilogue e2 = ep;
            if (ep.addend) e2.addend = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ep.addend) + m0 * ep.ld_add * ec);
            if (ep.mul) e2.mul = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ep.mul) + m0 * ep.ld_mul * ec);
            int rc = launch_linear_bf16(std::min(chunk, M - m0), N, K, P, static_cast<const uint16_t*>(A) + m0 * lda, lda, Wpk, e2,
                                        static_cast<char*>(C) + m0 * ldc * ec, ldc, c16, stream);
            if (rc) return rc;
        }
        return GVQA_OK;
    }
    dim3 grid((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 128));
    // vectorised epilogue: whole 8-column chunks, 16-byte aligned rows of C / addend / mul
    const int esz = c16 ? 2 : 4;
    auto al16 = [&](const void* q, int64_t ld) { return !q || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld * esz) % 16 == 0); };
    const int vec_ep = N % 8 == 0 && al16(C, ldc) && al16(ep.addend, ep.ld_add) && al16(ep.mul, ep.ld_mul) &&
                       (!ep.bias || (reinterpret_cast<uintptr_t>(ep.bias) & 3) == 0);
    const uintptr_t qa = c16 ? 7 : 15;        // quad = 8 bytes of bf16 / 16 bytes of fp32
    auto alq = [&](const void* q, int64_t ld) { return !q || ((reinterpret_cast<uintptr_t>(q) & qa) == 0 && ld % 4 == 0); };
    const bool vec_ep_quads = alq(C, ldc) && alq(ep.addend, ep.ld_add) && alq(ep.mul, ep.ld_mul) &&
                              (!ep.bias || (reinterpret_cast<uintptr_t>(ep.bias) & 15) == 0);
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* b = static_cast<const uint16_t*>(Wpk);
    static const bool no_dma = []() { const char* v = getenv("GVQA_BF16_GEMM"); return v && !strcmp(v, "regs"); }();
    // LDS-DMA staging: whole K steps, and the register epilogue's 4-column quads need aligned rows
    const bool quad_ok = N % 4 == 0 && vec_ep_quads;
    static const bool no_wide = []() { const char* v = getenv("GVQA_BF16_GEMM"); return v && !strcmp(v, "narrow"); }();
    // wide tile: whole K steps, a grid that fills the chip (>= 256 tiles of 256 x 128)
    // (measured, MFMA rate at the LCGN shapes: two-piece weights 808-852 TF wide vs 715 narrow; single piece 506-557 wide vs
    //  578-600 narrow -- without a second piece to share the A fragments the wide tile's longer block lifetime costs more)
    // 256 x 256 tile (two-piece weights, whole K steps of 32): from three quarters of a round of the CUs up to four rounds
    // (measured stand-alone, issued PF/s, 256 x 256 vs 256 x 128 -- `profiles/r06_bf16_gemm_ab.jsonl`: [29785, 512] x K 512 0.78 vs 0.72, x K 1024
    //  0.95 vs 0.90; [29785, 1536] x K 512 0.82 vs 0.78, x K 1024 0.89 vs 0.86; but [65536, 2048] x K 512 -- eight rounds -- 0.79 vs 0.85; the LCGN
    //  bf16 forward 2.207 -> 2.150 ms)
    static const bool no_big = []() { const char* v = getenv("GVQA_BF16_GEMM"); return v && (!strcmp(v, "wide") || !strcmp(v, "narrow")); }();
    const int64_t big_tiles = cdiv(M, 256) * cdiv(N, 256);
    if (P == 2 && K % 32 == 0 && !no_dma && !no_big && quad_ok && big_tiles >= 192 && big_tiles <= 4 * (int64_t)device_cu_count() && cdiv(M, 256) <= 65535) {
        dim3 gridb((unsigned)cdiv(N, 256), (unsigned)cdiv(M, 256));
        if (c16) hipLaunchKernelGGL(k_linear_bf16_big<true>, gridb, dim3(512 + 64 * GVQA_BIG_LOADERS), 0, stream, (int)M, (int)N, (int)K, a, lda, b, (int64_t)P * K, ep, C, ldc);
        else hipLaunchKernelGGL(k_linear_bf16_big<false>, gridb, dim3(512 + 64 * GVQA_BIG_LOADERS), 0, stream, (int)M, (int)N, (int)K, a, lda, b, (int64_t)P * K, ep, C, ldc);
        GVQA_LAUNCH_CHECK();
        return GVQA_OK;
    }
    if (P == 2 && K % 64 == 0 && !no_dma && !no_wide && quad_ok && cdiv(M, 256) * cdiv(N, 128) >= 256) {
        dim3 gridw((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 256));
#define GVQA_WIDE(C16_, P_) hipLaunchKernelGGL((k_linear_bf16_wide<C16_, P_>), gridw, dim3(512), 0, stream, (int)M, (int)N, (int)K, a, lda, b, \
                                               (int64_t)P * K, ep, C, ldc)
        if (c16) GVQA_WIDE(true, 2);
        else GVQA_WIDE(false, 2);
#undef GVQA_WIDE
        GVQA_LAUNCH_CHECK();
        return GVQA_OK;
    }
    if (K % 64 == 0 && !no_dma && quad_ok) {
        if (c16) hipLaunchKernelGGL(k_linear_bf16_dma<true>, grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, P, a, lda, b,
                                    (int64_t)P * K, ep, C, ldc);
        else hipLaunchKernelGGL(k_linear_bf16_dma<false>, grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, P, a, lda, b,
                                (int64_t)P * K, ep, C, ldc);
        GVQA_LAUNCH_CHECK();
        return GVQA_OK;
    }
    if (c16) hipLaunchKernelGGL((k_linear_bf16<128, 128, 2, 2, true>), grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, P, a, lda,
                                b, (int64_t)P * K, ep, C, ldc, vec_ep);
    else hipLaunchKernelGGL((k_linear_bf16<128, 128, 2, 2, false>), grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, P, a, lda,
                            b, (int64_t)P * K, ep, C, ldc, vec_ep);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// ---- the matrix pipes' own rate on given operands (measurement entry point, bench.py's `matrix_rate_measured`) ---------------------
// What `v_mfma_f32_32x32x16_{f16,bf16}` sustains on this chip at this moment with NOTHING else in the loop: eight waves per CU (two per SIMD, the
// occupancy of the hop kernels), eight accumulator tiles per wave, fragments loaded once from `operands` and rotated through the products.  The
// chip clocks to its power budget, and the budget depends on the operand bits (`profiles/r06_matrix_rate_probe.jsonl`): normally distributed fp16
// operands sustain 1.68 PF/s, bf16 1.84, small integers 2.1, zeros or ones 2.47-2.48 -- the data sheet's rate.
// The dense peak of the data sheet (2.5 PF/s) is the denominator of `roofline.frac`; this is the rate an MFMA-bound kernel can be compared with.
typedef _Float16 f16x8_probe __attribute__((ext_vector_type(8)));
template <bool BF, int ORDER>
__global__ __launch_bounds__(512) void k_mfma_stream(const uint4* __restrict__ operands, int nvec, float* __restrict__ sink, int iters) {
    const int g = blockIdx.x * 512 + threadIdx.x;
    uint4 a[4], b[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        a[f] = operands[(unsigned)(g * 8 + f) % (unsigned)nvec];
        b[f] = operands[(unsigned)(g * 8 + 4 + f) % (unsigned)nvec];
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)            // four K steps' worth of products: 16 MFMAs each, operands rotated
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        // ORDER 0: i-major (consecutive products share the A fragment, the B fragment alternates: the GEMM kernels' order);
                        // 1: snake (every consecutive pair shares one operand); 2: the same two fragments for 8 products in a row
                        const int j = ORDER == 1 ? ((i & 1) ? 1 - jj : jj) : jj;
                        const uint4 bb = ORDER == 2 ? b[(2 * p + u) & 3] : b[(2 * p + j + u) & 3], aa = ORDER == 2 ? a[(p + u) & 3] : a[(i + u) & 3];
                        if constexpr (BF) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bb), __builtin_bit_cast(bf16x8, aa), acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_probe, bb), __builtin_bit_cast(f16x8_probe, aa), acc[i][j], 0, 0, 0);
                    }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    sink[g] = t;
}

}  // namespace gvqa

extern "C" int gvqa_pack_weight_bf16(int64_t rows, int64_t K, int pieces, const float* W, int64_t ldw, void* Wpk, void* stream) {
    using namespace gvqa;
    GVQA_REQUIRE(rows >= 0 && K >= 0 && K < (1ll << 30) && ldw >= K, GVQA_E_INVALID, "pack_weight_bf16: bad size");
    GVQA_REQUIRE((W && Wpk) || rows * K == 0, GVQA_E_INVALID, "pack_weight_bf16: null operand");
    return launch_pack_weight_bf16(rows, (int)K, (int)K, pieces, W, ldw, Wpk, static_cast<hipStream_t>(stream));
}

extern "C" int gvqa_linear_bf16(int64_t M, int64_t N, int64_t K, int pieces, const void* A, int64_t lda, const void* Wpk,
                                const float* bias, const void* addend, int64_t ld_add, const void* mul, int64_t ld_mul, int relu,
                                void* C, int64_t ldc, int c_bf16, void* stream) {
    using namespace gvqa;
    GVQA_REQUIRE((!addend || ld_add >= N) && (!mul || ld_mul >= N), GVQA_E_INVALID, "linear_bf16: epilogue leading dimension too small");
    LinearEpilogue ep{bias, static_cast<const float*>(addend), ld_add, static_cast<const float*>(mul), ld_mul, relu};
    return launch_linear_bf16(M, N, K, pieces, A, lda, Wpk, ep, C, ldc, c_bf16 != 0, static_cast<hipStream_t>(stream));
}

extern "C" int gvqa_mfma_stream(const void* operands, size_t operand_bytes, float* sink, size_t sink_elems, int iters, int bf16, int64_t* flops_out,
                                void* stream) {
    using namespace gvqa;
    GVQA_REQUIRE(operands && sink && iters >= 1 && operand_bytes >= 1024 && (reinterpret_cast<uintptr_t>(operands) & 15) == 0, GVQA_E_INVALID,
                 "mfma_stream: needs a 16-byte aligned operand buffer of >= 1 KiB, a sink and iters >= 1");
    const int cus = device_cu_count();
    GVQA_REQUIRE(sink_elems >= (size_t)cus * 512, GVQA_E_WORKSPACE, "mfma_stream: sink holds %zu floats, %zu needed", sink_elems, (size_t)cus * 512);
    const int nvec = (int)std::min<size_t>(operand_bytes / 16, (size_t)1 << 24);
    // (bf16: bit 0 = bf16 products; bits 1-2 = the order of a step's products, see the kernel: 0 as the GEMM kernels issue them)
    const int order = (bf16 >> 1) & 3;
#define GVQA_MS(BF_, O_) hipLaunchKernelGGL((k_mfma_stream<BF_, O_>), dim3((unsigned)cus), dim3(512), 0, static_cast<hipStream_t>(stream), static_cast<const uint4*>(operands), nvec, sink, iters)
    if (bf16 & 1) { if (order == 1) GVQA_MS(true, 1); else if (order == 2) GVQA_MS(true, 2); else GVQA_MS(true, 0); }
    else { if (order == 1) GVQA_MS(false, 1); else if (order == 2) GVQA_MS(false, 2); else GVQA_MS(false, 0); }
#undef GVQA_MS
    GVQA_LAUNCH_CHECK();
    if (flops_out) *flops_out = (int64_t)cus * 8 * iters * 64 * 32768;      // workgroups x waves x iterations x 64 MFMAs x 2 x 32 x 32 x 16
    return GVQA_OK;
}
