// fp32-accurate dense projections on the CDNA4 bf16 matrix cores ("split3"):
//     C[M,N] = A[M,K] . B[N,K]^T  (+bias) (+addend) (*mul) (act),   A, B, C fp32.
//
// This is torch.nn.Linear's math for the node projection xp = lin_l(x_cat) of every GAT hop
// (/root/reference gat_skip.py:133) -- 76 % of the round-1 step on the f32-input MFMA (157 TF peak).
// gfx950 has no TF32/xf32, but an fp32 value is EXACTLY the sum of three bf16 pieces (8 significant
// bits each, round-to-nearest: v = p1 + p2 + p3 with |p2| <= 2^-9 |p1|, |p3| <= 2^-18 |p1|), and a
// product of two bf16 values is exact in fp32.  Keeping the six largest of the nine cross terms
//     a.b ~= a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1        (dropped: <= 3 * 2^-27 |a b|)
// every term is an exact product accumulated in fp32 by `v_mfma_f32_32x32x16_bf16` -- the same
// accuracy class as the k-ordered fmaf chain of the f32 MFMA (measured against fp64 in
// tests/test_gpu_split3.py), at 16x the matrix-core rate for 6x the work.
//
// What makes it MFMA-bound instead of staging-bound (a plain bf16 GEMM on a 128^2 tile needs the whole
// 64 B/clk/CU of the L1 path at full MFMA rate): the six products of a K step share THREE A fragments and
// THREE B fragments.  Per 16-deep K step a 256 x 256 block tile moves 48 KiB L2 -> LDS for 6 * 2*256*256*16
// flops = 3.9 B/kFLOP (16 B/clk/CU at full rate), and a wave reads 18 fragments for 48 MFMAs.
//
// Operand layout ("fragment-major", produced by k_split3_pack): the unit is the 1 KiB operand image of one
// 32x32x16 MFMA -- lane l = (row & 31) + 32 * ((k >> 3) & 1) holds the 8 consecutive k's of its row --
//     P[row / 32][k / 16][piece 0..2][lane 0..63][8 bf16]
// so one `global_load_lds_dwordx4` wave instruction copies one fragment HBM/L2 -> LDS (fully coalesced,
// 8 whole 128-byte lines), the LDS image is lane-linear and the fragment `ds_read_b128` (lane l reads
// bytes [16 l, 16 l + 16)) is conflict-free without padding or swizzle.  Rows are padded to 32 and K to 16
// with zeros by the pack kernel.
//
// Kernel: block = WM x WN waves, wave tile (32 TM) x (32 TN), block tile BM x BN = (32 TM WM) x (32 TN WN),
// K step 16, NBUF-deep LDS ring filled by LDS-DMA two steps ahead (counted `s_waitcnt vmcnt`, ONE raw
// barrier per K step: it orders "step s has landed for every wave" and "every wave is done reading the
// buffer that is refilled next").  MFMA operands are swapped (B fragment first) so the accumulators hold
// the transposed 32 x 32 tiles and a lane owns 4 consecutive columns of one C row per register quad: the
// epilogue is float4 stores straight from registers.
#include <algorithm>

#include "common.h"
#include "gemm_tile.h"

namespace gvqa {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// round-to-nearest-even bf16 of a finite fp32; values that would round up to infinity are truncated
// instead, inf / nan keep their top 16 bits (the remaining pieces are then garbage-in / garbage-out)
__device__ __forceinline__ uint16_t split3_rn(float f) {
    const unsigned u = __float_as_uint(f);
    unsigned t = u + 0x7FFFu + ((u >> 16) & 1u);
    if ((t & 0x7F800000u) == 0x7F800000u) t = u;
    return (uint16_t)(t >> 16);
}

// grid (ceil(KB / 4), RT); block 256 = 4 waves, wave w packs k block 4 blockIdx.x + w of row tile blockIdx.y
__global__ __launch_bounds__(256) void k_split3_pack(int64_t rows, int K, int KB, const float* __restrict__ X, int64_t ld,
                                                     uint16_t* __restrict__ out, int vec) {
    const int lane = threadIdx.x & 63, kb = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (kb >= KB) return;
    const int64_t rt = blockIdx.y;
    const int64_t row = rt * 32 + (lane & 31);
    const int k0 = kb * 16 + (lane >> 5) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (row < rows) {
        const float* src = X + row * ld + k0;
        if (vec && k0 + 8 <= K) {
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (k0 + e < K) v[e] = src[e];
        }
    }
    uint16_t p[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        p[0][e] = split3_rn(v[e]);
        const float r1 = v[e] - bf16_to_f32(p[0][e]);
        p[1][e] = split3_rn(r1);
        const float r2 = r1 - bf16_to_f32(p[1][e]);
        p[2][e] = split3_rn(r2);            // exact: at most 8 significant bits are left
    }
    uint16_t* o = out + ((rt * KB + kb) * 3) * 512 + lane * 8;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        uint4 w;
        w.x = (unsigned)p[q][0] | ((unsigned)p[q][1] << 16);
        w.y = (unsigned)p[q][2] | ((unsigned)p[q][3] << 16);
        w.z = (unsigned)p[q][4] | ((unsigned)p[q][5] << 16);
        w.w = (unsigned)p[q][6] | ((unsigned)p[q][7] << 16);
        *reinterpret_cast<uint4*>(o + q * 512) = w;
    }
}

// ILV: the DMAs of the step two ahead are issued one at a time between the six MFMA groups (their M0 set-up and
// address traffic then hides under matrix-core time) instead of in a burst behind the barrier; PRIO: raised wave
// priority over the MFMA groups; NOSTORE: measurement aid (main loop only, accumulators kept live).
__device__ __forceinline__ void keep_live(const f32x16& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#endif
}
// STAG > 0: blocks that share a CU (dispatch rounds of 256 blocks) start `stagger` x 8128 cycles apart, so that one
// block's C stores fall under its neighbours' main loops instead of every CU storing at the same time.
template <int WM, int WN, int TM, int TN, int NBUF, bool ILV, bool PRIO, bool NOSTORE, int STAG = 0, int EPI = 0>
__global__ __launch_bounds__(64 * WM * WN, (160 * 1024 / (NBUF * (WM * TM + WN * TN) * 3072)) * WM * WN / 4)
void k_linear_split3(int M, int N, int KB, const uint16_t* __restrict__ Apk, int rtA,
                                                               const uint16_t* __restrict__ Bpk, int rtB, LinearEpilogue ep,
                                                               float* __restrict__ C, int64_t ldc, int stagger) {
    constexpr int NW = WM * WN;
    constexpr int FA = WM * TM, FB = WN * TN;          // 32-row operand tiles per block: A rows, B rows (= C columns)
    constexpr int STAGE = (FA + FB) * 3072;            // bytes per K step: 3 pieces x 1 KiB per operand tile
    constexpr int TPW = (FA + FB) / NW;                // (tile, 3 pieces) triples each wave DMAs per K step
    static_assert((FA + FB) % NW == 0, "operand tiles must divide over the waves");
    static_assert(NBUF * STAGE <= 160 * 1024, "LDS ring exceeds 160 KiB");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NBUF * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int bm = blockIdx.y, bn = blockIdx.x;
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;
    if (STAG > 0) {
        const int slot = ((blockIdx.y * gridDim.x + blockIdx.x) >> 8) % STAG;
        for (int i = 0; i < slot * stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA duty of this wave: triples t = wave + q NW; t < FA: A tile t, else B tile t - FA.  Tiles past the
    // packed operand (edge blocks) re-read the last valid tile: their products are never stored.
    const uint16_t* src[TPW];
    unsigned dst[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * NW;
        const bool isA = t < FA;
        const int tile = isA ? min(bm * FA + t, rtA - 1) : min(bn * FB + (t - FA), rtB - 1);
        src[q] = (isA ? Apk : Bpk) + (int64_t)tile * KB * 1536 + lane * 8;
        dst[q] = lds_base + t * 3072;
    }
    auto issue = [&](int buf) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
                lds_dma16_b(src[q] + p * 512, __builtin_amdgcn_readfirstlane(dst[q] + buf * STAGE + p * 1024));
            src[q] += 1536;
        }
    };
    auto issue_one = [&](int buf, int n) {             // n-th of the 3 TPW DMAs of a step (n is a compile-time constant after unrolling)
        const int q = n / 3, p = n % 3;
        lds_dma16_b(src[q] + p * 512, __builtin_amdgcn_readfirstlane(dst[q] + buf * STAGE + p * 1024));
        if (p == 2) src[q] += 1536;
    };

    const unsigned a_off = (unsigned)(wr * TM * 3072 + lane * 16);
    const unsigned b_off = (unsigned)((FA + wc * TN) * 3072 + lane * 16);

    // prologue: NBUF - 1 steps in flight
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < KB) issue(s);
    int buf = 0, pf = NBUF - 1;                        // ring slot of step s / of the step issued in iteration s
    for (int s = 0; s < KB; ++s) {
        // steps s+1 .. s+NBUF-2 may stay in flight (3 TPW DMAs each); near the end fewer were issued
        const int ahead = min(NBUF - 2, KB - 1 - s);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 3 * TPW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * TPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = s + NBUF - 1 < KB;
        if (!ILV && more) issue(pf);                   // refills the slot read in iteration s-1
        const unsigned char* sb = smem + buf * STAGE;
        bf16x8_t af[TM][3], bfr[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                af[i][p] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + a_off + (i * 3 + p) * 1024));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bfr[j][p] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + b_off + (j * 3 + p) * 1024));
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        // smallest cross terms first, a1 b1 last; consecutive MFMAs hit different accumulators
#define GVQA_S3_PAIR(pa_, pb_, g_)                                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)              \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j][pb_], af[i][pa_], acc[i][j], 0, 0, 0);      \
        if (ILV && more) {                                                                                         \
            _Pragma("unroll") for (int n = (g_) * 3 * TPW / 6; n < ((g_) + 1) * 3 * TPW / 6; ++n) issue_one(pf, n); \
        }
        GVQA_S3_PAIR(2, 0, 0) GVQA_S3_PAIR(1, 1, 1) GVQA_S3_PAIR(0, 2, 2) GVQA_S3_PAIR(1, 0, 3) GVQA_S3_PAIR(0, 1, 4) GVQA_S3_PAIR(0, 0, 5)
#undef GVQA_S3_PAIR
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        pf = pf + 1 == NBUF ? 0 : pf + 1;
    }

    if (NOSTORE) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) keep_live(acc[i][j]);
        return;
    }
    auto finish = [&](float4 v, int gr, int gc) {      // bias / addend / mul / activation on 4 consecutive columns, then the store
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
    };
    // transposed accumulators: lane (m = lane & 31, h = lane >> 5) owns columns 8 q + 4 h + 0..3 of row m of tile (i, j)
    if (EPI == 0) {
        // straight from registers: a wave store covers 32 rows x 32 bytes
        const int mrow = lane & 31, ncol0 = 4 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gr = (bm * FA + wr * TM + i) * 32 + mrow;
            if (gr >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gc = (bn * FB + wc * TN + j) * 32 + 8 * q + ncol0;
                    if (gc >= N) continue;             // N % 4 == 0: a quad is entirely inside or outside
                    finish(make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]), gr, gc);
                }
        }
        return;
    }
    // EPI == 1: every 32 x 32 tile is turned through a wave-private 4 KiB LDS image (the operand ring is free after the
    // barrier) so that a wave store covers 8 rows x 128 bytes -- whole cache lines.  16-byte chunk c of row m sits at
    // slot c ^ (m & 7): conflict-free for the row-per-lane ds_write_b128 and the 8-lanes-per-row ds_read_b128 alike.
    __builtin_amdgcn_s_barrier();
    {
        const int m = lane & 31, h = lane >> 5;
        const int rr = lane >> 3, cc = lane & 7;       // read side: row within an 8-row pass, 16-byte chunk
        unsigned char* img = smem + wave * 8192;
        int flip = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                unsigned char* t = img + flip * 4096;
                flip ^= 1;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(t + m * 128 + (((2 * q + h) ^ (m & 7)) << 4)) =
                        make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                const int gc = (bn * FB + wc * TN + j) * 32 + cc * 4;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int r = ps * 8 + rr;
                    const float4 v = *reinterpret_cast<const float4*>(t + r * 128 + ((cc ^ (r & 7)) << 4));
                    const int gr = (bm * FA + wr * TM + i) * 32 + r;
                    if (gr < M && gc < N) finish(v, gr, gc);
                }
            }
    }
}

size_t split3_packed_bytes(int64_t rows, int64_t K) { return (size_t)cdiv(rows, 32) * (size_t)cdiv(K, 16) * 3072; }

int launch_split3_pack(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, hipStream_t stream) {
    GVQA_REQUIRE(rows >= 0 && K >= 0 && K < (1ll << 30) && ld >= K, GVQA_E_INVALID, "split3_pack: bad size");
    if (rows == 0 || K == 0) return GVQA_OK;
    GVQA_REQUIRE(X && packed, GVQA_E_INVALID, "split3_pack: null operand");
    GVQA_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, GVQA_E_INVALID, "split3_pack: packed buffer must be 16-byte aligned");
    const int KB = (int)cdiv(K, 16);
    const int64_t RT = cdiv(rows, 32);
    const int vec = (reinterpret_cast<uintptr_t>(X) & 15) == 0 && ld % 4 == 0;
    for (int64_t r0 = 0; r0 < RT; r0 += 65535) {       // grid.y holds 65535 row tiles
        const int64_t n = std::min<int64_t>(65535, RT - r0);
        hipLaunchKernelGGL(k_split3_pack, dim3((unsigned)cdiv(KB, 4), (unsigned)n), dim3(256), 0, stream, rows - r0 * 32, (int)K, KB,
                           X + r0 * 32 * ld, ld, static_cast<uint16_t*>(packed) + r0 * KB * 1536, vec);
    }
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

bool linear_split3_supported(int64_t N, const LinearEpilogue& ep, const float* C, int64_t ldc) {
    auto al16 = [&](const void* q, int64_t ld) { return !q || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld % 4 == 0); };
    return N % 4 == 0 && al16(C, ldc) && al16(ep.addend, ep.ld_add) && al16(ep.mul, ep.ld_mul) && al16(ep.bias, 4);
}

// Kernel variant by shape (GVQA_OPT_SPLIT3_VARIANT forces an exact instantiation; see the switch below).  Measured on
// MI355X (scripts/bench_split3.py): the 256 x 256 tile (8 waves, 3-deep ring, one block per CU) wins when its tiles
// fill the CUs evenly and K is long enough to amortise a tile's prologue / store tail; the 128 x 256 tile (4 waves,
// 2 blocks per CU whose store tails overlap each other's main loops) otherwise.
static int split3_variant(int64_t M, int64_t N, int KB) {
    const int forced = get_option(GVQA_OPT_SPLIT3_VARIANT);
    if (forced >= 10) return forced;
    auto eff = [](int64_t tiles, int64_t slots) { return (double)tiles / (double)(cdiv(tiles, slots) * slots); };
    const double e_big = eff(cdiv(M, 256) * cdiv(N, 256), 256), e_half = eff(cdiv(M, 128) * cdiv(N, 256), 512);
    return (KB >= 24 && e_big >= e_half - 0.02) ? 14 : 34;
}

int launch_linear_split3(int64_t M, int64_t N, int64_t K, const void* Apk, const void* Bpk, LinearEpilogue ep, float* C,
                         int64_t ldc, hipStream_t stream) {
    GVQA_REQUIRE(M >= 0 && N >= 0 && K > 0, GVQA_E_INVALID, "linear_split3: bad size");
    GVQA_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 30), GVQA_E_INVALID, "linear_split3: size overflow");
    if (M == 0 || N == 0) return GVQA_OK;
    GVQA_REQUIRE(Apk && Bpk && C, GVQA_E_INVALID, "linear_split3: null operand");
    GVQA_REQUIRE(ldc >= N && (!ep.addend || ep.ld_add >= N) && (!ep.mul || ep.ld_mul >= N), GVQA_E_INVALID,
                 "linear_split3: leading dimension too small");
    GVQA_REQUIRE(linear_split3_supported(N, ep, C, ldc), GVQA_E_UNSUPPORTED,
                 "linear_split3: N %% 4 == 0 and 16-byte aligned C / addend / mul / bias rows required");
    const int KB = (int)cdiv(K, 16), rtB = (int)cdiv(N, 32);
    const uint16_t* a = static_cast<const uint16_t*>(Apk);
    const uint16_t* b = static_cast<const uint16_t*>(Bpk);
    const int variant = split3_variant(M, N, KB);
    const int64_t bm = variant < 20 ? 256 : 128;
    const char* ssv = getenv("GVQA_SPLIT3_STAGGER");      // quarter units of the default start offset (tuning aid)
    const int stag_scale = ssv ? atoi(ssv) : 0;
    const int64_t rows_per_launch = 65535 * bm;        // grid.y limit: row chunks (rows are independent)
    for (int64_t m0 = 0; m0 < M; m0 += rows_per_launch) {
        const int64_t m = std::min(rows_per_launch, M - m0);
        LinearEpilogue e2 = ep;
        if (ep.addend) e2.addend = ep.addend + m0 * ep.ld_add;
        if (ep.mul) e2.mul = ep.mul + m0 * ep.ld_mul;
        const uint16_t* a2 = a + (m0 / 32) * (int64_t)KB * 1536;
        const int rt2 = (int)cdiv(m, 32);
#define GVQA_S3_LAUNCH(WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_)                                            \
        do {                                                                                                             \
            dim3 grid((unsigned)cdiv(N, 32 * WN_ * TN_), (unsigned)cdiv(m, 32 * WM_ * TM_));                              \
            /* a block's MFMA issue time x the STAG_ blocks sharing the SIMDs, split into STAG_ start offsets */         \
            const int stag = STAG_ > 0 ? (int)((int64_t)KB * 6 * TM_ * TN_ * 32 / 8128) : 0;                              \
            hipLaunchKernelGGL((k_linear_split3<WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_>), grid,       \
                               dim3(64 * WM_ * WN_), 0, stream, (int)m, (int)N, KB, a2, rt2, b, rtB, e2, C + m0 * ldc, ldc, \
                               stag_scale > 0 ? stag * stag_scale / 4 : stag);                                           \
        } while (0)
        switch (variant) {
            case 10: GVQA_S3_LAUNCH(2, 4, 4, 2, 3, false, false, false, 0, 0); break;
            case 11: GVQA_S3_LAUNCH(2, 4, 4, 2, 3, true, false, false, 0, 0); break;
            case 13: GVQA_S3_LAUNCH(2, 4, 4, 2, 3, true, false, true, 0, 0); break;
            case 20: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, false, false, false, 0, 0); break;
            case 21: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, true, false, false, 0, 0); break;
            case 22: GVQA_S3_LAUNCH(2, 2, 2, 2, 2, true, false, false, 0, 0); break;
            case 23: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, true, false, true, 0, 0); break;
            case 26: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, true, false, false, 2, 0); break;
            case 27: GVQA_S3_LAUNCH(2, 2, 2, 2, 2, true, false, false, 3, 0); break;
            case 14: GVQA_S3_LAUNCH(2, 4, 4, 2, 3, true, false, false, 0, 1); break;
            case 28: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, true, false, false, 0, 1); break;
            case 29: GVQA_S3_LAUNCH(2, 2, 2, 2, 2, true, false, false, 0, 1); break;
            case 30: GVQA_S3_LAUNCH(2, 2, 2, 4, 2, true, false, false, 0, 0); break;
            case 34: GVQA_S3_LAUNCH(2, 2, 2, 4, 2, true, false, false, 0, 1); break;
            case 31: GVQA_S3_LAUNCH(2, 2, 2, 4, 2, true, false, false, 2, 0); break;
            case 33: GVQA_S3_LAUNCH(2, 2, 2, 4, 2, true, false, true, 0, 0); break;
            default: return GVQA_E_INVALID;
        }
#undef GVQA_S3_LAUNCH
    }
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // namespace gvqa

extern "C" size_t gvqa_split3_packed_bytes(int64_t rows, int64_t K) {
    if (rows <= 0 || K <= 0) return 0;
    return gvqa::split3_packed_bytes(rows, K);
}

extern "C" int gvqa_split3_pack(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, void* stream) {
    return gvqa::launch_split3_pack(rows, K, X, ld, packed, static_cast<hipStream_t>(stream));
}

extern "C" int gvqa_linear_split3(int64_t M, int64_t N, int64_t K, const void* Apk, const void* Bpk, const float* bias,
                                  const float* addend, int64_t ld_add, const float* mul, int64_t ld_mul, int relu, float* C,
                                  int64_t ldc, void* stream) {
    gvqa::LinearEpilogue ep{bias, addend, ld_add, mul, ld_mul, relu};
    return gvqa::launch_linear_split3(M, N, K, Apk, Bpk, ep, C, ldc, static_cast<hipStream_t>(stream));
}
