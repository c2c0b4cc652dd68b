// fp32-accurate dense projections on the CDNA4 16-bit matrix cores:
//     C[M,N] = A[M,K] . B[N,K]^T  (+bias) (+addend) (*mul) (act),   A, B, C fp32.
// Two operand forms share the kernels below (template parameter NP = pieces per value): "split3", three exact bf16 pieces and
// six piece products -- described first -- and "split2h" (the default of the library), two scaled fp16 pieces and three
// products -- described at k_split2h_pack.
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
// buffer that is refilled next"; KS = 2: two K steps per stage and per barrier, two stages).  MFMA operands are swapped (B fragment first) so the accumulators hold
// the transposed 32 x 32 tiles and a lane owns 4 consecutive columns of one C row per register quad: the
// epilogue is float4 stores straight from registers.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "gemm_tile.h"

namespace gvqa {

// ablation switches of the fused epilogue / the main loop exist in the measurement build only (python -m graphvqa_amd.build --probes)
#ifdef GVQA_PROBES
#define GVQA_FH_DBG(bit_) (fh.debug & (bit_))
#else
#define GVQA_FH_DBG(bit_) false
#endif

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
template <int NP> struct SplitFrag { typedef bf16x8_t type; };
template <> struct SplitFrag<2> { typedef f16x8_t type; };
__device__ __forceinline__ f32x16 split_mfma(const bf16x8_t& a, const bf16x8_t& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 split_mfma(const f16x8_t& a, const f16x8_t& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// round-to-nearest-even bf16 of a finite fp32; values that would round up to infinity are truncated
// instead, inf / nan keep their top 16 bits (the remaining pieces are then garbage-in / garbage-out)
__device__ __forceinline__ uint16_t split3_rn(float f) {
    const unsigned u = __float_as_uint(f);
    unsigned t = u + 0x7FFFu + ((u >> 16) & 1u);
    if ((t & 0x7F800000u) == 0x7F800000u) t = u;
    return (uint16_t)(t >> 16);
}

// the three pieces of 8 consecutive k's of one row -> the lane's 16-byte slots of the three fragments at `o`
__device__ __forceinline__ void split3_store(const float (&v)[8], uint16_t* o) {
    uint16_t p[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        p[0][e] = split3_rn(v[e]);
        const float r1 = v[e] - bf16_to_f32(p[0][e]);
        p[1][e] = split3_rn(r1);
        const float r2 = r1 - bf16_to_f32(p[1][e]);
        p[2][e] = split3_rn(r2);            // exact: at most 8 significant bits are left
    }
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
    split3_store(v, out + ((rt * KB + kb) * 3) * 512 + lane * 8);
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
// NP: pieces per operand value -- 3: bf16 x 3 (six products), 2: scaled fp16 x 2 (three products; a_inv / b_inv = the rows'
// inverse power-of-two scales, applied to the accumulators before anything else in every epilogue)
// CHN = 1 (EPI = 2, NP = 2): chained hops -- skip rows out of the packed input, output as the next hop's packed operand + per-graph
// maxima (the epilogue of hop2.hip's CHAIN form on this kernel's 256 x 256 tile: no pack pass between hops)
template <int WM, int WN, int TM, int TN, int NBUF, bool ILV, bool PRIO, bool NOSTORE, int STAG = 0, int EPI = 0, int FH = 0, int NP = 3, int PIPE = 0, int KS = 1, int CHN = 0>
__global__ __launch_bounds__(64 * WM * WN, (160 * 1024 / (NBUF * (WM * TM + WN * TN) * NP * 1024)) * WM * WN / 4)
void k_linear_split3(int M, int N, int KB, const uint16_t* __restrict__ Apk, int rtA,
                                                               const uint16_t* __restrict__ Bpk, int rtB, LinearEpilogue ep,
                                                               float* __restrict__ C, int64_t ldc, int stagger, FusedHopArgs fh,
                                                               const float* __restrict__ a_inv, const float* __restrict__ b_inv) {
    static_assert(NP == 2 || NP == 3, "two fp16 pieces or three bf16 pieces");
    constexpr int NW = WM * WN;
    constexpr int FA = WM * TM, FB = WN * TN;          // 32-row operand tiles per block: A rows, B rows (= C columns)
    constexpr int FRAG = NP * 1024;                    // bytes of one (tile, K step): NP pieces x 1 KiB
    constexpr int NG = NP * (NP + 1) / 2;              // piece products kept per K step
    constexpr int TPW = (FA + FB + NW - 1) / NW;       // (tile, NP pieces) groups each wave DMAs per K step
    // (FA + FB not a multiple of the wave count -- the 128 x 512 tile: 20 tiles, 8 waves --: every wave still issues TPW groups, so
    //  that the counted waits are uniform; the surplus groups re-read the last B tile into unused slots of the padded stage)
    constexpr int STAGE = TPW * NW * FRAG;             // bytes per K step
    typedef typename SplitFrag<NP>::type frag_t;
    static_assert(EPI != 3 || (NP == 2 && KS == 1 && !NOSTORE), "packed-output epilogue: two-piece operands");
    static_assert(NBUF * STAGE * KS <= 160 * 1024, "LDS ring exceeds 160 KiB");
    static_assert(KS == 1 || (KS == 2 && NP == 2 && NBUF == 2 && !PIPE), "two K steps per stage: two-piece operands, two stages");
    static_assert(EPI != 2 || (WM == 2 && WN == 4 && TM == 4 && TN == 2), "the fused-hop epilogue is written for the 256 x 256 tile");
    static_assert(CHN == 0 || (EPI == 2 && NP == 2 && FH == 4), "chained epilogue: fused hop, two-piece operands, H = 4");
    constexpr bool CH = CHN != 0;
    // CHN = 2: ... and the hop's attention coefficients are computed IN this kernel (no coefficient kernel, no alpha_csr round trip):
    // partial node logits (left by the previous hop's column blocks / the pack pass) summed per row group, edge halves gathered through
    // csr_eid, leaky-relu + segment softmax per (node, head) in LDS between the main loop and the row image (gat_skip.py:180-190); the
    // epilogue leaves the NEXT hop's partial node logits of this column block (its finished rows . Vn_next over its own channels).
    constexpr bool IC = CHN == 2;
    constexpr unsigned IC_VN0 = 141 * 1024;            // (IC) [2 H][64] slice of the next hop's folded attention vectors: between the two 13 KiB regions
    constexpr unsigned IC_NG0 = 143 * 1024;            // (IC) [2][128] graph ids of the two groups' nodes
    constexpr unsigned CH_TAIL = 160 * 1024 - 3072;    // (CH) per group: inverse scales of the input slots | output scales by graph | output maxima by graph
    __shared__ __attribute__((aligned(1024))) unsigned char smem[EPI == 2 ? 160 * 1024 : NBUF * STAGE * KS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    int bm = blockIdx.y, bn = blockIdx.x;
    if (EPI != 2 && gridDim.z > 1) {                   // batched launch (split-K chunks): this batch's operands and result
        const int64_t z = blockIdx.z;
        Apk += z * ep.zs_a; Bpk += z * ep.zs_b; C += z * ep.zs_c;
        if (a_inv) a_inv += z * ep.zs_ia;
        if (b_inv) b_inv += z * ep.zs_ib;
    }
    if (EPI == 2 && fh.xcd_cols > 1) {
        // Workgroup w of the launch order runs on XCD w % 8 (each XCD has its own L2).  With the plain order the 8 column blocks
        // of a row block sit on 8 different XCDs and every L2 streams the whole A operand; here XCD x owns `xcd_cols` column
        // blocks (whose weights stay in its L2) of every (8 / column groups)-th row block, consecutive workgroups of an XCD
        // walking the column blocks of one row block.
        const int w = blockIdx.y * gridDim.x + blockIdx.x, x = w & 7, j = w >> 3;
        const int ncg = gridDim.x / fh.xcd_cols, cg = x % ncg, rg = x / ncg;
        bn = cg * fh.xcd_cols + j % fh.xcd_cols;
        bm = (j / fh.xcd_cols) * (8 / ncg) + rg;
    }
    // fused hop: the first fh.pair_blocks row blocks take TWO row groups (256 rows), the rest ONE (the last partial round of a launch run
    // as half tiles: the four waves of the empty half skip their fragment reads and MFMAs -- 585 workgroups on 256 CUs are three
    // rounds for 2.29 rounds of work at config 2; 510 pairs + 145 singles are two rounds and a short one)
    int rtf = bm * FA, grp0 = bm * 2;              // first A row tile / first row group of this block
    bool single = false;
    if constexpr (EPI == 2) {
        if (bm >= fh.pair_blocks) { grp0 = 2 * fh.pair_blocks + (bm - fh.pair_blocks); single = true; }
        rtf = grp0 * 4;
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;
    if (STAG > 0) {
        const int slot = ((blockIdx.y * gridDim.x + blockIdx.x) >> 8) % STAG;
        for (int i = 0; i < slot * stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // (measurement aid, NOSTORE variants: shader-clock cycles and 100 MHz ticks of this block -> C[2 block], C[2 block + 1])
    const uint64_t clk0 = NOSTORE ? __builtin_readcyclecounter() : 0, rt0 = NOSTORE ? __builtin_amdgcn_s_memrealtime() : 0;

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
    const unsigned lane16 = (unsigned)lane * 16u;
    // (two K segments of A, ep.a2: the first image holds a2_kb0 k blocks per row tile; at k block a2_kb0 the A tiles' sources move to the second image)
    const bool seg2 = EPI != 2 && NP == 2 && ep.a2 != nullptr;
    const int kb_sw = seg2 ? ep.a2_kb0 : 0x7fffffff;
    const uint16_t* src2[TPW];
    int kbi[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * NW;
        const bool isA = t < FA;
        const int tile = isA ? min(rtf + t, rtA - 1) : min(bn * FB + (t - FA), rtB - 1);
        // (wave-uniform: the ring's DMAs take them as scalar bases, the lane part is the constant lane16 below)
        src[q] = (isA ? Apk : Bpk) + (int64_t)tile * ((isA && seg2) ? ep.a2_kb0 : KB) * (NP * 512);
        src2[q] = (isA && seg2) ? ep.a2 + (int64_t)tile * ep.a2_KB * (NP * 512) : nullptr;
        kbi[q] = 0;
        dst[q] = lds_base + t * FRAG;
    }
    // accumulators from the first segment's row scale to the second's (lane (m, .) holds rows m of its tiles)
    auto rescale_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gr = min((bm * FA + wr * TM + i) * 32 + (lane & 31), M - 1);
            const float ratio = a_inv[gr] / ep.a2_inv[gr];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= ratio;
        }
    };
    // one (tile, k step) = NP 1 KiB fragments contiguous in global memory and in the LDS stage: ONE M0 set-up, NP DMAs
    auto issue_triple = [&](int buf, int q) {
        if constexpr (NP == 2 && EPI != 2) {
            if (kbi[q] == kb_sw && src2[q]) src[q] = src2[q];       // (wave-uniform)
            kbi[q] += 1;
        }
        lds_dma16_s<NP>(src[q], lane16, __builtin_amdgcn_readfirstlane(dst[q] + buf * STAGE));
        src[q] += NP * 512;
    };
    auto issue = [&](int buf) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) issue_triple(buf, q);
    };

    const unsigned a_off = (unsigned)(wr * TM * FRAG + lane * 16);
    const unsigned b_off = (unsigned)((FA + wc * TN) * FRAG + lane * 16);

    // ---- fused hop (EPI == 2): per-block constants, row-group metadata and the first group's CSR slice start their trips
    // from HBM before the main loop (their latency hides under it); see the epilogue below ------------------------------
    constexpr int Hh = FH > 0 ? FH : 1, cw = 256 / Hh, q4 = cw >> 2;
    constexpr int lq = Hh == 1 ? 6 : Hh == 2 ? 5 : Hh == 4 ? 4 : 3;
    static_assert(EPI != 2 || FH == 1 || FH == 2 || FH == 4 || FH == 8, "fused hop: H must be 1, 2, 4 or 8");
    constexpr int NTH = 64 * NW;
    // LDS of the epilogue: [0, 128 KiB) row image xs; [128, 160 KiB) the groups' regions [rowptr | csr_src | alpha] (RAW global
    // values, rebased where they are used; every sub-array padded to whole 64-lane DMA instructions).  When a region fits
    // 16 KiB, group 0's lives in [144, 160 KiB) -- outside the operand ring, so it is filled during the main loop -- and group
    // 1's in [128, 144 KiB), filled while group 0 is aggregated; otherwise one region at 128 KiB is used twice.
    const int src_off = 192, al_off = src_off + ((fh.e_cap + 63) & ~63);
    const int eid_off = al_off + ((fh.e_cap * Hh + 63) & ~63);          // (IC) the slots' COO edge ids
    const int reg_words = eid_off + (IC ? ((fh.e_cap + 63) & ~63) : 0);
    const bool two_regions = EPI == 2 && (CH ? reg_words * 4 <= 13 * 1024 : 2 * reg_words * 4 <= 32 * 1024);      // (CH: the last 3 KiB are taken)
    // a thread of the epilogue owns CV adjacent channel quads (8 channels when the head slice is >= 64 wide: the edge loop is
    // issue-bound, and its index / coefficient / address work is then shared by twice the FMAs) of `items` rows
    constexpr int CV = (EPI == 2 && q4 >= 16) ? 2 : 1;
    constexpr int lqv = lq - (CV == 2 ? 1 : 0);               // log2(lanes per row)
    const int cq = tid & ((q4 / CV) - 1);
    const int c = bn * cw + cq * CV * 4;                      // first of this thread's 4 CV output channels
    bool c_ok[CV];
    float4 bi[CV], sc[CV], sh[CV];
#pragma unroll
    for (int v = 0; v < CV; ++v) {
        c_ok[v] = EPI == 2 && c + 4 * v < fh.C;
        bi[v] = make_float4(0.f, 0.f, 0.f, 0.f); sh[v] = bi[v]; sc[v] = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    int g_ns[2] = {0, 0}, g_cnt[2] = {0, 0}, g_e0[2] = {0, 0}, g_ne[2] = {0, 0};
    // CSR slice + coefficients of a group -> LDS bytes [byte_base, ...) by LDS-DMA (4 bytes per lane: no alignment constraints);
    // every wave issues the same number of DMAs (lanes past the end re-load the last word into the padding)
    auto dma_region = [&](int gi, unsigned byte_base) {
        const unsigned base = lds_base + byte_base;
        const int n_rp = g_cnt[gi] + 1, n_src = g_ne[gi], n_al = g_ne[gi] * Hh;
        const int32_t* rp_g = fh.rowptr + g_ns[gi];
        const int32_t* src_g = fh.csr_src + g_e0[gi];
        [[maybe_unused]] const float* al_g = IC ? nullptr : fh.alpha_csr + (int64_t)g_e0[gi] * Hh;
        const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
        for (int u = wbase; u < n_rp; u += NTH)
            lds_dma4_b(rp_g + min(u + lane, n_rp - 1), __builtin_amdgcn_readfirstlane(base + (unsigned)u * 4u));
        for (int u = wbase; u < n_src; u += NTH)
            lds_dma4_b(src_g + min(u + lane, n_src - 1), __builtin_amdgcn_readfirstlane(base + (unsigned)(src_off + u) * 4u));
        if constexpr (IC) {                                   // (the coefficients are made here: the slots' edge ids instead)
            const int32_t* eid_g = fh.ic_csr_eid + g_e0[gi];
            for (int u = wbase; u < n_src; u += NTH)
                lds_dma4_b(eid_g + min(u + lane, n_src - 1), __builtin_amdgcn_readfirstlane(base + (unsigned)(eid_off + u) * 4u));
        } else {
            for (int u = wbase; u < n_al; u += NTH)
                lds_dma4_b(al_g + min(u + lane, n_al - 1), __builtin_amdgcn_readfirstlane(base + (unsigned)(al_off + u) * 4u));
        }
    };
    if constexpr (EPI == 2) {
#pragma unroll
        for (int v = 0; v < CV; ++v) {
            if (!c_ok[v]) continue;
            const int cv = c + 4 * v;
            if (fh.bias) bi[v] = *reinterpret_cast<const float4*>(fh.bias + cv);
            if (fh.bn_w) {          // torch's eval BatchNorm: y = x (w invstd) + (b - mean w invstd)
                const float4 w4 = *reinterpret_cast<const float4*>(fh.bn_w + cv), b4 = *reinterpret_cast<const float4*>(fh.bn_b + cv);
                const float4 m4 = *reinterpret_cast<const float4*>(fh.bn_m + cv), v4 = *reinterpret_cast<const float4*>(fh.bn_v + cv);
                sc[v].x = w4.x * (1.0f / sqrtf(v4.x + fh.bn_eps)); sc[v].y = w4.y * (1.0f / sqrtf(v4.y + fh.bn_eps));
                sc[v].z = w4.z * (1.0f / sqrtf(v4.z + fh.bn_eps)); sc[v].w = w4.w * (1.0f / sqrtf(v4.w + fh.bn_eps));
                sh[v].x = b4.x - m4.x * sc[v].x; sh[v].y = b4.y - m4.y * sc[v].y; sh[v].z = b4.z - m4.z * sc[v].z; sh[v].w = b4.w - m4.w * sc[v].w;
            }
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int grp = grp0 + gi;
            if (grp < fh.num_groups && !(single && gi == 1)) {                        // block-uniform
                g_ns[gi] = fh.group_ptr[grp];
                g_cnt[gi] = fh.group_ptr[grp + 1] - g_ns[gi];
                g_e0[gi] = fh.rowptr[g_ns[gi]];
                g_ne[gi] = min(fh.rowptr[g_ns[gi] + g_cnt[gi]] - g_e0[gi], fh.e_cap);     // (a wrong loader-side layout must not overrun the region)
            }
        }
        // older than every ring DMA: the counted waits of the main loop only get stricter by them, never wrong
        if (two_regions && g_cnt[0] > 0 && !GVQA_FH_DBG(8)) dma_region(0, 144 * 1024);
        if constexpr (IC) {
            // both groups' regions are needed together (the coefficient phase runs before the first row image): the second one and the
            // next hop's folded-vector slice of this column block also start now -- [128, 144 KiB) is above the operand ring
            if (g_cnt[1] > 0) dma_region(1, 128 * 1024);
            if (wave >= 4 && wave < 8) {      // the nodes' graph ids (instruction-term offsets, output scales and maxima are per graph) and input-slot scales
                const int gi = (wave - 4) >> 1, u = ((wave - 4) & 1) * 64;
                if (g_cnt[gi] > 0) {
                    lds_dma4_b(fh.node_graph + g_ns[gi] + min(u + lane, g_cnt[gi] - 1), __builtin_amdgcn_readfirstlane(lds_base + IC_NG0 + (unsigned)(gi * 128 + u) * 4u));
                    lds_dma4_b(fh.ch_a_inv + (grp0 + gi) * 128 + u + lane, __builtin_amdgcn_readfirstlane(lds_base + CH_TAIL + (unsigned)(gi * 1536 + u * 4)));
                }
            }
            if (tid < 256) reinterpret_cast<unsigned*>(smem + CH_TAIL + (tid >> 7) * 1536 + 1024)[tid & 127] = 0u;      // the groups' output maxima
            if (fh.ic_lp_out && wave < 2) {
                const int j = (wave * 64 + lane) >> 4, part = lane & 15;
                lds_dma16_b(fh.ic_vn_next + (int64_t)j * fh.C + min(bn * cw + part * 4, fh.C - 4),
                            __builtin_amdgcn_readfirstlane(lds_base + IC_VN0 + (unsigned)wave * 1024u));
            }
        }
    }
    // Rows of a group are aggregated in the order fh.row_order gives (most in-edges first: the edge loop's trip count is the
    // largest in-degree among the rows a wave covers, so rows of similar degree belong together).  Slot -> row for this
    // thread's slots of both groups, loaded now (a dependent load in front of the skip prefetch otherwise).
    constexpr int MAXIT = 4;
    constexpr int items = (128 << lqv) / NTH;                 // rows per thread: 128 (q4 / CV) / 512
    constexpr bool pre = items <= MAXIT;
    static_assert(!CH || pre, "chained epilogue: rows per thread within the prefetch window");
    int ord[2][MAXIT];
    if constexpr (EPI == 2 && pre) {
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int k = 0; k < MAXIT; ++k) {
                const int slot = (tid >> lqv) + k * (NTH >> lqv);
                ord[gi][k] = (fh.row_order && k < items && slot < g_cnt[gi]) ? fh.row_order[g_ns[gi] + slot] : slot;
            }
    }
    // two-piece operands: inverse scales of this wave's rows x the column block's single weight scale (k_split2h_pack, HEADS)
    float sab[TM];
    if constexpr (EPI == 2 && NP == 2) {
        const float sbu = b_inv[bn * 256];
#pragma unroll
        for (int i = 0; i < TM; ++i) sab[i] = a_inv[min(rtf + wr * TM + i, rtA - 1) * 32 + (lane & 31)] * sbu;
    }

    // (fused hop) the waves of a row half without a row group take no part in the products (they still issue their DMAs and meet the barriers)
    const bool mm_on = EPI != 2 || __builtin_amdgcn_readfirstlane((int)(wr == 0 || g_cnt[1] > 0)) != 0;
    if constexpr (KS == 2) {
        // ---- two K steps per stage and per barrier: a stage is 4 KiB per operand tile (two k blocks x two pieces, contiguous in
        // the packed operand), two stages in LDS; the next stage's DMAs (four per tile from one M0 set-up) go out between the
        // first MFMA groups of the current one.  Half the barriers and waits of the one-step loop; needs an even number of k blocks.
        constexpr int FRAG2 = 2 * FRAG, STAGE2 = 2 * STAGE;
        unsigned dst2[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q) dst2[q] = lds_base + (wave + q * NW) * FRAG2;
        auto issue_quad = [&](int b, int q) {
            if constexpr (EPI != 2) {
                if (kbi[q] == kb_sw && src2[q]) src[q] = src2[q];   // (wave-uniform; the launcher keeps a2_kb0 even for this loop)
                kbi[q] += 2;
            }
            lds_dma16_s<4>(src[q], lane16, __builtin_amdgcn_readfirstlane(dst2[q] + b * STAGE2));
            src[q] += 2 * NP * 512;
        };
        const unsigned a_off2 = (unsigned)(wr * TM * FRAG2 + lane * 16);
        const unsigned b_off2 = (unsigned)((FA + wc * TN) * FRAG2 + lane * 16);
#pragma unroll
        for (int q = 0; q < TPW; ++q) issue_quad(0, q);
        int b = 0;
        const int KB2 = KB >> 1;
        frag_t af[TM][NP], bfr[TN][NP];
        for (int s2 = 0; s2 < KB2; ++s2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool more = s2 + 1 < KB2;
            if constexpr (EPI != 2) { if (2 * s2 == kb_sw) rescale_acc(); }
            const unsigned char* sb = smem + b * STAGE2;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (mm_on) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        af[i][p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(sb + a_off2 + i * FRAG2 + sub * FRAG + p * 1024));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        bfr[j][p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(sb + b_off2 + j * FRAG2 + sub * FRAG + p * 1024));
                }
#define GVQA_S2_PAIR(pa_, pb_, g_)                                                                                  \
                if (mm_on) { _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)        \
                    acc[i][j] = split_mfma(bfr[j][pb_], af[i][pa_], acc[i][j]); }                                   \
                if (more && sub == 0 && (g_) < TPW) issue_quad(b ^ 1, (g_));
                GVQA_S2_PAIR(1, 0, 0) GVQA_S2_PAIR(0, 1, 1) GVQA_S2_PAIR(0, 0, 2)
#undef GVQA_S2_PAIR
            }
            b ^= 1;
        }
        static_assert(KS == 1 || TPW <= 3, "the next stage's DMAs go out behind the three MFMA groups of the first sub-step");
    } else if constexpr (PIPE) {
        // ---- software-pipelined main loop: the fragments of step s+1 are read from LDS (into a second register set) while the
        // matrix cores work on step s, so the LDS reads of all waves -- which leave the barrier together -- no longer sit
        // between the barrier and the first MFMA.  The whole ring is in flight: iteration s waits for step s+1, and step
        // s+NBUF goes into the slot of step s (whose fragments every wave finished reading before the barrier).
        static_assert(!PIPE || (NBUF == 4 && ILV), "pipelined loop: four-slot ring, interleaved DMA issue");
#pragma unroll
        for (int s = 0; s < NBUF; ++s)
            if (s < KB) issue(s);
        {
            const int fl = min(NBUF - 1, KB - 1);      // steps 1 .. may stay in flight
            if (fl >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NP * TPW) : "memory");
            else if (fl == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP * TPW) : "memory");
            else if (fl == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * TPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        frag_t fa[2][TM][NP], fb[2][TN][NP];
        auto read_frags = [&](int slot, frag_t (&a)[TM][NP], frag_t (&b)[TN][NP]) {
            const unsigned char* sb = smem + slot * STAGE;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    a[i][p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(sb + a_off + (i * NP + p) * 1024));
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    b[j][p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(sb + b_off + (j * NP + p) * 1024));
        };
        read_frags(0, fa[0], fb[0]);
        int slot = 0;                                  // ring slot of step s
        // (the wait / barrier / read-ahead also run in the last step, where they fetch a stale slot nobody uses: a run-time
        //  "is there a next step" branch around the reads makes the compiler wait for ALL LDS reads where the paths join,
        //  in front of the MFMAs -- and peeling the last step off made it spill)
        auto body = [&](int s, frag_t (&ca)[TM][NP], frag_t (&cb)[TN][NP], frag_t (&na)[TM][NP], frag_t (&nb)[TN][NP]) {
            const int nslot = slot + 1 == NBUF ? 0 : slot + 1;
            {
                const int ahead = min(NBUF - 2, KB - 2 - s);          // steps s+2 .. may stay in flight (<= 0 at the end)
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP * TPW) : "memory");
                else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * TPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // this wave's reads of step s (issued one iteration ago) are complete: its slot may be refilled after the barrier.
                // (the builtin, not inline asm: the compiler's own wait-count bookkeeping sees it and does not add an
                //  lgkmcnt(0) of its own AFTER the reads below, which would serialise them with the MFMAs again)
                __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0), vmcnt / expcnt untouched
                __builtin_amdgcn_s_barrier();
                read_frags(nslot, na, nb);
            }
            const bool more = s + NBUF < KB;
#define GVQA_S3_PAIR(pa_, pb_, g_)                                                                                 \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)          \
                acc[i][j] = split_mfma(cb[j][pb_], ca[i][pa_], acc[i][j]);                                         \
            if (more) {                                                                                            \
                _Pragma("unroll") for (int q = (g_) * TPW / NG + ((g_) * TPW % NG ? 1 : 0); q * NG < ((g_) + 1) * TPW; ++q) \
                    if (q * NG >= (g_) * TPW) issue_triple(slot, q);                                               \
            }
            if constexpr (NP == 3) {
                GVQA_S3_PAIR(2, 0, 0) GVQA_S3_PAIR(1, 1, 1) GVQA_S3_PAIR(0, 2, 2) GVQA_S3_PAIR(1, 0, 3) GVQA_S3_PAIR(0, 1, 4) GVQA_S3_PAIR(0, 0, 5)
            } else {
                GVQA_S3_PAIR(1, 0, 0) GVQA_S3_PAIR(0, 1, 1) GVQA_S3_PAIR(0, 0, 2)
            }
#undef GVQA_S3_PAIR
            slot = nslot;
        };
        for (int s = 0; s < KB; s += 2) {
            body(s, fa[0], fb[0], fa[1], fb[1]);
            if (s + 1 < KB) body(s + 1, fa[1], fb[1], fa[0], fb[0]);
        }
    } else {
    // prologue: NBUF - 1 steps in flight
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < KB) issue(s);
    int buf = 0, pf = NBUF - 1;                        // ring slot of step s / of the step issued in iteration s
    // (measurement aid, STAG == 0 only: `stagger` bits 16 / 32 / 64 = no fragment reads after step 0 / no DMA in the loop /
    //  no waits and barriers -- wrong results; scripts/bench_split3_loop.py prices the parts of a K step with them)
#ifdef GVQA_PROBES
    const int dbg = STAG == 0 ? stagger : 0;
#else
    constexpr int dbg = 0;
#endif
    static_assert(NBUF <= 4, "the counted waits below know at most two steps in flight behind the current one");
    frag_t af[TM][NP], bfr[TN][NP];
    for (int s = 0; s < KB; ++s) {
        if (!(dbg & 64)) {
            // steps s+1 .. s+NBUF-2 may stay in flight (NP TPW DMAs each); near the end fewer were issued
            const int ahead = min(NBUF - 2, KB - 1 - s);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP * TPW) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * TPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const bool more = s + NBUF - 1 < KB && !(dbg & 32);
        if constexpr (NP == 2 && EPI != 2) { if (s == kb_sw) rescale_acc(); }
        if (!ILV && more) issue(pf);                   // refills the slot read in iteration s-1
        const unsigned char* sb = smem + buf * STAGE;
        if (mm_on && (s == 0 || !(dbg & 16))) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    af[i][p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(sb + a_off + (i * NP + p) * 1024));
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bfr[j][p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(sb + b_off + (j * NP + p) * 1024));
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        // smallest cross terms first, a1 b1 last; consecutive MFMAs hit different accumulators
#define GVQA_S3_PAIR(pa_, pb_, g_)                                                                                 \
        if (mm_on) { _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)              \
            acc[i][j] = split_mfma(bfr[j][pb_], af[i][pa_], acc[i][j]); }                                          \
        if (ILV && more) {                                                                                         \
            _Pragma("unroll") for (int q = (g_) * TPW / NG + ((g_) * TPW % NG ? 1 : 0); q * NG < ((g_) + 1) * TPW; ++q) \
                if (q * NG >= (g_) * TPW) issue_triple(pf, q);                                                     \
        }
        if constexpr (NP == 3) {
            GVQA_S3_PAIR(2, 0, 0) GVQA_S3_PAIR(1, 1, 1) GVQA_S3_PAIR(0, 2, 2) GVQA_S3_PAIR(1, 0, 3) GVQA_S3_PAIR(0, 1, 4) GVQA_S3_PAIR(0, 0, 5)
        } else {
            GVQA_S3_PAIR(1, 0, 0) GVQA_S3_PAIR(0, 1, 1) GVQA_S3_PAIR(0, 0, 2)
        }
#undef GVQA_S3_PAIR
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        pf = pf + 1 == NBUF ? 0 : pf + 1;
    }
    }

    if (seg2) a_inv = ep.a2_inv;                       // (the accumulators are in the second segment's units now)
    if (NOSTORE) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) keep_live(acc[i][j]);
        if (tid == 0 && C) {
            const int blk = blockIdx.y * gridDim.x + blockIdx.x;
            C[2 * blk] = (float)(__builtin_readcyclecounter() - clk0);
            C[2 * blk + 1] = (float)(__builtin_amdgcn_s_memrealtime() - rt0);
        }
        return;
    }
    if constexpr (EPI == 2) {
        // ---- fused GAT aggregation (gat_skip.py:155-168,270-275 on the projected tile) --------------------------
        // The block's 256 rows are two row groups of <= 128 consecutive nodes that end on graph boundaries, its 256
        // columns are the channels [bn cw, bn cw + cw) of ALL H heads (weights packed head-interleaved), so every
        // neighbour row a node of these groups needs is in this tile.  Per group: the four waves that own its rows
        // put their accumulators into LDS (xs[row][256], 16-byte chunk c at slot c ^ (row & 7)), the group's CSR slice and
        // attention coefficients are brought in beside it, then all 512 threads do
        //     out[i, c] = (1/H) sum_h sum_{e: dst = i} alpha[e, h] xs[src_e][h cw + c]  (+ graph term) + bias + skip -> BN -> ReLU
        // and store `out`.  xp never reaches HBM.
        if (GVQA_FH_DBG(8)) return;
        float* xs = reinterpret_cast<float*>(smem);
        const float inv_h = 1.0f / Hh;
        const bool relu = fh.bn_w != nullptr;
        __syncthreads();                                      // main loop done: operand ring free
        if constexpr (IC) {
            // ---- attention coefficients of both row groups (gat_skip.py:180-190), in LDS.  Every wave's DMAs have landed (the main
            // loop's last wait is vmcnt(0)) and the barrier above made them visible: rowptr | src | eid of both groups and their nodes'
            // graph ids are in place.  ALL global reads of the phase -- edge halves through the edge ids, partial node logits, per-graph
            // logit offsets -- are issued before the first one is consumed: one memory latency for the phase, not one per read (the
            // first version of this phase ran gather, sums and offsets one after the other, group by group: +9 us per workgroup).
            float* an_s = reinterpret_cast<float*>(smem);     // [2][128][2 H] node logits, in the free operand ring
            if (!GVQA_FH_DBG(32)) {
            const int* ng_l = reinterpret_cast<const int*>(smem + IC_NG0);
            const int r = tid >> 2, hq = tid & 3;             // thread (row r, head hq) of the softmax
            const int sg = tid >> 8, sr = (tid & 255) >> 1, sh = tid & 1;     // thread (group sg, row sr, logit half sh) of the partial sums
            constexpr int SPT = 2;                            // slots per thread and group: the capacity is 522 edges <= 2 x 512
            float4 ae[2][SPT];
            int eidv[2][SPT];
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int* eid_w = reinterpret_cast<const int*>(smem + (gi == 0 ? 144 * 1024 : 128 * 1024)) + eid_off;
#pragma unroll
                for (int k = 0; k < SPT; ++k) eidv[gi][k] = eid_w[min(tid + k * NTH, max(g_ne[gi] - 1, 0))];
            }
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                for (int k = 0; k < SPT; ++k)
                    ae[gi][k] = (tid + k * NTH < g_ne[gi]) ? *reinterpret_cast<const float4*>(fh.ic_a_edge + (int64_t)eidv[gi][k] * fh.ic_a_edge_stride)
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
            float tl[2] = {0.f, 0.f};
            if (fh.graph_term) {
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
                    if (r < g_cnt[gi]) tl[gi] = fh.graph_term[(int64_t)ng_l[gi * 128 + r] * fh.t_ld + fh.C + hq];
            }
            // output scales of the rows this hop leaves (packed), one power of two per graph: thread (group tid >> 7, graph tid & 127 of the group)
            [[maybe_unused]] const int zg = tid >> 7, zt = tid & 127;
            [[maybe_unused]] int z_gf = 0, z_ngl = 0;
            float zM = 0.f, ztm = 0.f;
            float4 zbc = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool z_on = fh.ch_pnext != nullptr && tid < 256 && (zg == 0 ? g_cnt[0] : g_cnt[1]) > 0;
            if (z_on) {
                const int cnt_z = zg == 0 ? g_cnt[0] : g_cnt[1], ns_z = zg == 0 ? g_ns[0] : g_ns[1];
                z_gf = ng_l[zg * 128];
                z_ngl = ng_l[zg * 128 + cnt_z - 1] - z_gf + 1;
                if (zt < z_ngl) {
                    const int g = z_gf + zt;
                    if (fh.ic_pmin) {                          // the graph's largest INPUT magnitude as the previous hop's column blocks left it
                        for (int q = 0; q < fh.ic_parts_in; ++q) zM = fmaxf(zM, fh.ic_pmin[(int64_t)q * fh.ch_B + g]);
                    } else {                                   // first hop: from the input slots' scales (M = 2^14 / the smallest one)
                        const float* rinv_z = reinterpret_cast<const float*>(smem + CH_TAIL + zg * 1536);
                        const int r0 = max(fh.ch_graph_ptr[g] - ns_z, 0), r1 = min(fh.ch_graph_ptr[g + 1] - ns_z, cnt_z);
                        for (int rr_ = r0; rr_ < r1; ++rr_) zM = fmaxf(zM, rinv_z[rr_]);
                        zM *= 16384.f;
                    }
                    ztm = fh.ch_tmax ? fh.ch_tmax[g] : 0.f;
                    zbc = make_float4(fh.ch_bc[0], fh.ch_bc[1], fh.ch_bc[2], fh.ch_bc[3]);
                }
            }
            float4 sum4 = make_float4(0.f, 0.f, 0.f, 0.f);
            {   // node logits: the partial sets summed in a fixed order (deterministic)
                const int cnt_s = sg == 0 ? g_cnt[0] : g_cnt[1], ns_s = sg == 0 ? g_ns[0] : g_ns[1];
                if (sr < cnt_s) {
                    const float* lp = fh.ic_lp_in + (int64_t)(ns_s + sr) * 8 + sh * 4;
                    if (fh.ic_parts_in == 8) {                // (C = 512: all eight reads in flight together)
                        float4 v4[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v4[q] = *reinterpret_cast<const float4*>(lp + (int64_t)q * fh.ic_lp_stride);
#pragma unroll
                        for (int q = 0; q < 8; ++q) { sum4.x += v4[q].x; sum4.y += v4[q].y; sum4.z += v4[q].z; sum4.w += v4[q].w; }
                    } else {
                        for (int q = 0; q < fh.ic_parts_in; ++q) {
                            const float4 v4 = *reinterpret_cast<const float4*>(lp + (int64_t)q * fh.ic_lp_stride);
                            sum4.x += v4.x; sum4.y += v4.y; sum4.z += v4.z; sum4.w += v4.w;
                        }
                    }
                }
            }
            *reinterpret_cast<float4*>(an_s + (sg * 128 + sr) * 8 + sh * 4) = sum4;
            if (z_on && zt < z_ngl) {
                //   |out| <= max|BN scale| (M (max L1 norm of a weight row + 1) + max|instruction term| + max|bias|) + max|BN shift|   (hop2.hip)
                const float bound = (zbc.y * (zM * (zbc.x + 1.f) + ztm + zbc.w) + zbc.z) * 1.001f;
                reinterpret_cast<float*>(smem + CH_TAIL + zg * 1536)[128 + zt] = pow2i(split2h_exponent(bound));
            }
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                float* al_w = reinterpret_cast<float*>(smem + (gi == 0 ? 144 * 1024 : 128 * 1024)) + al_off;
#pragma unroll
                for (int k = 0; k < SPT; ++k)
                    if (tid + k * NTH < g_ne[gi]) *reinterpret_cast<float4*>(al_w + (tid + k * NTH) * 4) = ae[gi][k];
            }
            __syncthreads();
            // leaky-relu + segment softmax of (row r, head hq) of BOTH groups: the first 8 in-edges of a row branch-free in registers
            // (clamped slots, masked afterwards: every LDS read of a stage issued together, the two groups' chains interleaved), further
            // ones through the region in LDS
            {
                constexpr int DR = 8;
                float v[2][DR], ar2[2], mx2[2] = {-INFINITY, -INFINITY};
                int lo2[2], dg2[2];
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int* rp_w = reinterpret_cast<const int*>(smem + (gi == 0 ? 144 * 1024 : 128 * 1024));
                    const bool on_r = r < g_cnt[gi];
                    const int rr_ = on_r ? r : 0;
                    lo2[gi] = rp_w[rr_] - g_e0[gi];
                    dg2[gi] = on_r ? min(rp_w[rr_ + 1] - g_e0[gi], g_ne[gi]) - lo2[gi] : 0;
                    ar2[gi] = an_s[(gi * 128 + rr_) * 8 + 4 + hq] + tl[gi];
                }
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int* rp_w = reinterpret_cast<const int*>(smem + (gi == 0 ? 144 * 1024 : 128 * 1024));
                    const int* src_w = rp_w + src_off;
                    const float* al_w = reinterpret_cast<const float*>(rp_w + al_off);
                    const float* an_g = an_s + gi * 128 * 8;
                    const int last = max(lo2[gi] + dg2[gi] - 1, 0);
                    int sn[DR];
#pragma unroll
                    for (int e = 0; e < DR; ++e) sn[e] = min(max(src_w[min(lo2[gi] + e, last)] - g_ns[gi], 0), 127);
#pragma unroll
                    for (int e = 0; e < DR; ++e) {
                        const float raw = an_g[sn[e] * 8 + hq] + al_w[min(lo2[gi] + e, last) * 4 + hq];
                        float t_ = raw + ar2[gi];
                        t_ = t_ > 0.f ? t_ : t_ * fh.ic_slope;
                        v[gi][e] = e < dg2[gi] ? t_ : -INFINITY;
                        mx2[gi] = fmaxf(mx2[gi], v[gi][e]);
                    }
                }
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int* rp_w = reinterpret_cast<const int*>(smem + (gi == 0 ? 144 * 1024 : 128 * 1024));
                    const int* src_w = rp_w + src_off;
                    float* al_w = reinterpret_cast<float*>(const_cast<int*>(rp_w) + al_off);
                    const int* eid_w = rp_w + eid_off;
                    const float* an_g = an_s + gi * 128 * 8;
                    const int lo = lo2[gi], hi = lo + dg2[gi];
                    for (int sl = lo + DR; sl < hi; ++sl) {   // (rows with more than 8 in-edges)
                        const float raw = an_g[min(max(src_w[sl] - g_ns[gi], 0), 127) * 8 + hq] + al_w[sl * 4 + hq];
                        float t_ = raw + ar2[gi];
                        t_ = t_ > 0.f ? t_ : t_ * fh.ic_slope;
                        al_w[sl * 4 + hq] = t_;
                        mx2[gi] = fmaxf(mx2[gi], t_);
                    }
                    float den = 0.f;
#pragma unroll
                    for (int e = 0; e < DR; ++e) {
                        v[gi][e] = e < dg2[gi] ? __expf(v[gi][e] - mx2[gi]) : 0.f;
                        den += v[gi][e];
                    }
                    for (int sl = lo + DR; sl < hi; ++sl) {
                        const float ex = __expf(al_w[sl * 4 + hq] - mx2[gi]);
                        al_w[sl * 4 + hq] = ex;
                        den += ex;
                    }
                    den += 1e-16f;                           // torch_geometric.utils.softmax
                    const float rden = 1.0f / den;
#pragma unroll
                    for (int e = 0; e < DR; ++e)
                        if (e < dg2[gi]) {
                            const float al = v[gi][e] * rden;
                            al_w[(lo + e) * 4 + hq] = al;
                            if (fh.ic_alpha_out && bn == 0) fh.ic_alpha_out[(int64_t)eid_w[lo + e] * 4 + hq] = al;
                        }
                    for (int sl = lo + DR; sl < hi; ++sl) {
                        const float al = al_w[sl * 4 + hq] * rden;
                        al_w[sl * 4 + hq] = al;
                        if (fh.ic_alpha_out && bn == 0) fh.ic_alpha_out[(int64_t)eid_w[sl] * 4 + hq] = al;
                    }
                }
            }
            __syncthreads();                                  // coefficients in place; an_s (in the ring) is free for the row image
            if (z_on) {                                       // the next operand's inverse slot scales (every column block writes the same values)
                const int cnt_z = zg == 0 ? g_cnt[0] : g_cnt[1];
                const float* gscl_z = reinterpret_cast<const float*>(smem + CH_TAIL + zg * 1536) + 128;
                fh.ch_a_inv_next[(grp0 + zg) * 128 + zt] = zt < cnt_z ? 1.0f / gscl_z[ng_l[zg * 128 + zt] - z_gf] : 1.f;
            }
            }
        } else {
        if (two_regions) { if (g_cnt[1] > 0) dma_region(1, 128 * 1024); }
        else if (g_cnt[0] > 0) dma_region(0, 128 * 1024);
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int ns = g_ns[gi], cnt = g_cnt[gi], e0 = g_e0[gi];
            const bool live = cnt > 0;                        // block-uniform
            const unsigned region_base = (two_regions && gi == 0) ? 144 * 1024 : 128 * 1024;
            if (gi == 1) {
                __syncthreads();                              // group 0's image and (single) region are free
                if (!two_regions && live) dma_region(1, 128 * 1024);
            }
            // the graph ids of this thread's rows start their trip from HBM now, ahead of the LDS work.  (The skip rows used to be
            // fetched here as well: 16 registers held across the row-image write and the barrier put the H = 4 kernel into
            // scratch; loaded at the top of `process` instead -- still ahead of the edge loop -- 415 -> 406 us per launch.)
            int gid[MAXIT];
            if (pre && live) {
#pragma unroll
                for (int k = 0; k < MAXIT; ++k) {
                    const int slot = (tid >> lqv) + k * (NTH >> lqv);
                    const bool on = k < items && slot < cnt;
                    const int node = ns + (on ? ord[gi][k] : 0);
                    gid[k] = (CH || fh.graph_term) ? fh.node_graph[node] : 0;
                }
            }
            // (CH) this group's input-slot scales and cleared output maxima -> LDS tail; first / number of graph ids of the group
            [[maybe_unused]] float* rinv_l = reinterpret_cast<float*>(smem + CH_TAIL + gi * 1536);
            [[maybe_unused]] float* gscl_l = rinv_l + 128;
            [[maybe_unused]] unsigned* gmax_l = reinterpret_cast<unsigned*>(rinv_l + 256);
            [[maybe_unused]] int gf = 0, ngl = 0;
            [[maybe_unused]] const int grp = grp0 + gi;
            [[maybe_unused]] const bool chain_out = CH && fh.ch_pnext != nullptr;
            if constexpr (IC) {
                if (live) {                                   // (graph ids, input-slot scales and cleared maxima came in before the main loop)
                    const int* ng_g = reinterpret_cast<const int*>(smem + IC_NG0) + gi * 128;
                    gf = ng_g[0];
                    ngl = ng_g[cnt - 1] - gf + 1;
                }
            } else if constexpr (CH) {
                if (live) {
                    gf = fh.node_graph[ns];
                    ngl = fh.node_graph[ns + cnt - 1] - gf + 1;
                    if (tid < 128) { rinv_l[tid] = fh.ch_a_inv[grp * 128 + tid]; gmax_l[tid] = 0u; }
                }
            }
            if (live && wr == gi && !GVQA_FH_DBG(1)) {
                const int m = lane & 31, hh = lane >> 5;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r = i * 32 + m;
                            const int chunk = wc * 16 + j * 8 + 2 * q + hh;
                            float4 t = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                            if constexpr (NP == 2) { t.x *= sab[i]; t.y *= sab[i]; t.z *= sab[i]; t.w *= sab[i]; }
                            *reinterpret_cast<float4*>(xs + r * 256 + ((chunk ^ (r & 7)) << 2)) = t;
                        }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs (and prefetches) have landed
            __syncthreads();
            if constexpr (CH) {
                // first hop of a chain: the output rows' power-of-two scale per graph from an upper bound of their magnitudes (hop2.hip):
                //   |out| <= max|BN scale| (M (max L1 norm of a weight row + 1) + max|instruction term| + max|bias|) + max|BN shift|,
                // M = 2^14 / (smallest input scale of the graph's rows); later hops get their scales from the coefficient kernel
                if (!IC && live && chain_out && !fh.ch_gscale) {       // (block-uniform; IC: decided in the coefficient phase, for both groups)
                    if (tid < ngl) {
                        const int g = gf + tid;
                        const int r0 = max(fh.ch_graph_ptr[g] - ns, 0), r1 = min(fh.ch_graph_ptr[g + 1] - ns, cnt);
                        float M = 0.f;
                        if (IC && fh.ic_pmin) {                // the graph's largest INPUT magnitude as the previous hop's column blocks left it
                            for (int q = 0; q < fh.ic_parts_in; ++q) M = fmaxf(M, fh.ic_pmin[(int64_t)q * fh.ch_B + g]);
                        } else {
                            for (int r = r0; r < r1; ++r) M = fmaxf(M, rinv_l[r]);
                            M *= 16384.f;
                        }
                        const float tm = fh.ch_tmax ? fh.ch_tmax[g] : 0.f;
                        const float bound = (fh.ch_bc[1] * (M * (fh.ch_bc[0] + 1.f) + tm + fh.ch_bc[3]) + fh.ch_bc[2]) * 1.001f;
                        gscl_l[tid] = pow2i(split2h_exponent(bound));
                    }
                    __syncthreads();
                    if (tid < 128) fh.ch_a_inv_next[grp * 128 + tid] = tid < cnt ? 1.0f / gscl_l[fh.node_graph[ns + tid] - gf] : 1.f;
                }
            }
            if (live) {
                const int* rp_l = reinterpret_cast<const int*>(smem + region_base);
                const int* src_l = rp_l + src_off;
                const float* al_l = reinterpret_cast<const float*>(rp_l + al_off);
                const float4* xs4 = reinterpret_cast<const float4*>(xs);
                // one output row segment: node i of the group, channels [c, c + 4 CV)
                auto process = [&](int slot, int row, bool have, int gq_pre) {
                    const bool row_on = slot < cnt;
                    const int i = row_on ? row : 0;
                    const int lo = rp_l[i] - e0, hi = (row_on && !GVQA_FH_DBG(2)) ? rp_l[i + 1] - e0 : lo;
                    const int node = ns + i;
                    float4 pb[CV];
#pragma unroll
                    for (int v = 0; v < CV; ++v) pb[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                    [[maybe_unused]] int gq = 0;
                    if constexpr (CH) gq = have ? gq_pre : fh.node_graph[node];
                    if (fh.graph_term && hi > lo) {           // nodes without in-edges get no instruction term (empty softmax)
                        if constexpr (!CH) gq = have ? gq_pre : fh.node_graph[node];
#pragma unroll
                        for (int v = 0; v < CV; ++v)
                            if (c_ok[v]) pb[v] = *reinterpret_cast<const float4*>(fh.graph_term + (int64_t)gq * fh.t_ld + c + 4 * v);
                    }
                    float4 sq[CV];                            // skip row segment: on its way from here, consumed after the edge loop
                    // (CH) ... out of the packed input operand: channels [c, c + 8) of slot i are lane (i & 31) + 32 ((c >> 3) & 1) of
                    // k block c >> 4 -- 16 bytes per piece; (p1 + p2) / scale is the value the projection itself saw
                    [[maybe_unused]] uint4 sk1 = make_uint4(0u, 0u, 0u, 0u), sk2 = sk1;
                    [[maybe_unused]] float gsc_pre = 1.f;
                    if constexpr (CH) {
                        static_assert(!CH || CV == 2, "chained epilogue: 8 channels per thread");
                        if (c < fh.C) {
                            const uint16_t* pp = fh.ch_apk + ((int64_t)((grp * 4 + (i >> 5)) * fh.ch_KB + (c >> 4)) * 2) * 512 + ((i & 31) + 32 * ((c >> 3) & 1)) * 8;
                            sk1 = *reinterpret_cast<const uint4*>(pp);
                            sk2 = *reinterpret_cast<const uint4*>(pp + 512);
                        }
                        if (chain_out && fh.ch_gscale) gsc_pre = fh.ch_gscale[gq];
                    } else {
#pragma unroll
                        for (int v = 0; v < CV; ++v)
                            sq[v] = (fh.skip && c_ok[v]) ? *reinterpret_cast<const float4*>(fh.skip + (int64_t)node * fh.skip_ld + c + 4 * v)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    // EB edges per trip, every LDS read of a trip issued before its FMAs; the trip count is made wave-uniform
                    // (clamped index, zero weight past the end of the row): a divergent, dependent-load loop was 4x slower
                    constexpr int EB = (8 / (Hh * CV)) > 0 ? 8 / (Hh * CV) : 1;      // 8 row reads (32 VGPRs) in flight per trip
                    // the wave covers 64 / (q4 / CV) rows: largest in-degree among them
                    int maxdeg = hi - lo;
#pragma unroll
                    for (int o = 32; o >= (1 << lqv); o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o, 64));
                    const int trips = GVQA_FH_DBG(2) ? 0 : __builtin_amdgcn_readfirstlane((maxdeg + EB - 1) / EB);
                    float4 a4[CV], b4[CV];                     // two chains per quad: consecutive FMAs do not wait for each other
#pragma unroll
                    for (int v = 0; v < CV; ++v) a4[v] = b4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int tr = 0; tr < trips; ++tr) {
                        const int s0 = lo + tr * EB;
                        int se[EB];
                        float al[EB][Hh];
#pragma unroll
                        for (int e = 0; e < EB; ++e) {
                            // unconditional reads at a clamped (always mapped) slot, selected afterwards: no branch in the loop
                            const bool on = s0 + e < hi;
                            const int idx = max(min(s0 + e, hi - 1), 0);
                            // (always a valid slot of the group: its source row is used as it is, with weight zero when the slot is not
                            //  this row's; a select on `on` here makes hipcc predicate the read and drain the LDS queue behind it)
                            se[e] = min(max(src_l[idx] - ns, 0), 127);
                            if constexpr (Hh % 4 == 0) {
#pragma unroll
                                for (int h4 = 0; h4 < Hh / 4; ++h4) {
                                    const float4 t4 = *reinterpret_cast<const float4*>(al_l + idx * Hh + h4 * 4);
                                    al[e][h4 * 4] = on ? t4.x : 0.f; al[e][h4 * 4 + 1] = on ? t4.y : 0.f;
                                    al[e][h4 * 4 + 2] = on ? t4.z : 0.f; al[e][h4 * 4 + 3] = on ? t4.w : 0.f;
                                }
                            } else {
#pragma unroll
                                for (int h = 0; h < Hh; ++h) {
                                    const float t1 = al_l[idx * Hh + h];
                                    al[e][h] = on ? t1 : 0.f;
                                }
                            }
                        }
                        float4 v[EB][Hh][CV];
#pragma unroll
                        for (int e = 0; e < EB; ++e)
#pragma unroll
                            for (int h = 0; h < Hh; ++h)
#pragma unroll
                                for (int w = 0; w < CV; ++w) v[e][h][w] = xs4[se[e] * 64 + (((h << lq) + cq * CV + w) ^ (se[e] & 7))];
#pragma unroll
                        for (int e = 0; e < EB; ++e)
#pragma unroll
                            for (int h = 0; h < Hh; ++h)
#pragma unroll
                                for (int w = 0; w < CV; ++w) {
                                    float4& t = ((e * Hh + h) & 1) ? b4[w] : a4[w];
                                    t.x += al[e][h] * v[e][h][w].x; t.y += al[e][h] * v[e][h][w].y;
                                    t.z += al[e][h] * v[e][h][w].z; t.w += al[e][h] * v[e][h][w].w;
                                }
                    }
                    [[maybe_unused]] float4 rr[CV];
#pragma unroll
                    for (int w = 0; w < CV; ++w) {
                        float4 r = make_float4((a4[w].x + b4[w].x) * inv_h + pb[w].x, (a4[w].y + b4[w].y) * inv_h + pb[w].y,
                                               (a4[w].z + b4[w].z) * inv_h + pb[w].z, (a4[w].w + b4[w].w) * inv_h + pb[w].w);
                        r.x += bi[w].x; r.y += bi[w].y; r.z += bi[w].z; r.w += bi[w].w;
                        if constexpr (CH) {
                            const f16x8_t h1 = __builtin_bit_cast(f16x8_t, sk1), h2 = __builtin_bit_cast(f16x8_t, sk2);
                            const float rs = rinv_l[i];
                            r.x += ((float)h1[4 * w] + (float)h2[4 * w]) * rs; r.y += ((float)h1[4 * w + 1] + (float)h2[4 * w + 1]) * rs;
                            r.z += ((float)h1[4 * w + 2] + (float)h2[4 * w + 2]) * rs; r.w += ((float)h1[4 * w + 3] + (float)h2[4 * w + 3]) * rs;
                        } else if (fh.skip && c_ok[w]) {
                            const float4 s4 = sq[w];
                            r.x += s4.x; r.y += s4.y; r.z += s4.z; r.w += s4.w;
                        }
                        if (relu) {
                            r.x = fmaxf(r.x * sc[w].x + sh[w].x, 0.f); r.y = fmaxf(r.y * sc[w].y + sh[w].y, 0.f);
                            r.z = fmaxf(r.z * sc[w].z + sh[w].z, 0.f); r.w = fmaxf(r.w * sc[w].w + sh[w].w, 0.f);
                        }
                        if (CH && chain_out) rr[w] = (row_on && c_ok[w]) ? r : make_float4(0.f, 0.f, 0.f, 0.f);
                        else if (row_on && c_ok[w] && !GVQA_FH_DBG(4)) *reinterpret_cast<float4*>(fh.out + (int64_t)node * fh.out_ld + c + 4 * w) = r;
                    }
                    if constexpr (CH) {
                        if (chain_out && (c >> 4) < fh.ch_KB) {   // (channels >= C inside the last k block: the operand's zero padding)
                            // the segment leaves as its 16 bytes of the next hop's two operand fragments: slot (rows past the group's end
                            // write their zeros at their own slot), k block c >> 4, lane (slot & 31) + 32 ((c >> 3) & 1)
                            const int orow_i = row_on ? i : slot;
                            const float scl = row_on ? (fh.ch_gscale ? gsc_pre : gscl_l[gq - gf]) : 1.f;
                            const float vv[8] = {rr[0].x, rr[0].y, rr[0].z, rr[0].w, rr[1].x, rr[1].y, rr[1].z, rr[1].w};
                            f16x8_t p0, p1;
                            float mxv = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float xsc = vv[e] * scl;
                                const _Float16 hi16 = (_Float16)xsc;
                                p0[e] = hi16;
                                p1[e] = (_Float16)(xsc - (float)hi16);
                                mxv = fmaxf(mxv, fabsf(vv[e]));
                            }
                            uint16_t* dstp = fh.ch_pnext + ((int64_t)((grp * 4 + (orow_i >> 5)) * fh.ch_KB + (c >> 4)) * 2) * 512 + ((orow_i & 31) + 32 * ((c >> 3) & 1)) * 8;
                            *reinterpret_cast<uint4*>(dstp) = __builtin_bit_cast(uint4, p0);
                            *reinterpret_cast<uint4*>(dstp + 512) = __builtin_bit_cast(uint4, p1);
                            if (row_on) atomicMax(&gmax_l[gq - gf], __float_as_uint(mxv));      // (bit patterns of non-negative floats order like integers)
                        }
                    }
                    if constexpr (IC) {
                        if (fh.ic_lp_out && !GVQA_FH_DBG(16)) {                    // (block-uniform; whole waves are here: the shuffles below are safe)
                            // the next hop's node logits, this column block's share: a_node[node, j] += sum_c h_next[node, c] Vn_next[j, c]
                            // over the thread's 8 channels, then the row's 8 lanes meet by a transposed reduction (lane cq ends up with j = cq)
                            const float* vn_l = reinterpret_cast<const float*>(smem + IC_VN0) + cq * 8;
                            float pl[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const float4 va = *reinterpret_cast<const float4*>(vn_l + j * 64), vb = *reinterpret_cast<const float4*>(vn_l + j * 64 + 4);
                                pl[j] = (rr[0].x * va.x + rr[0].y * va.y) + (rr[0].z * va.z + rr[0].w * va.w) +
                                        ((rr[1].x * vb.x + rr[1].y * vb.y) + (rr[1].z * vb.z + rr[1].w * vb.w));
                            }
                            float s1[4], s2[2];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float mine = (cq & 4) ? pl[4 + k] : pl[k], other = (cq & 4) ? pl[k] : pl[4 + k];
                                s1[k] = mine + __shfl_xor(other, 4, 64);
                            }
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const float mine = (cq & 2) ? s1[2 + k] : s1[k], other = (cq & 2) ? s1[k] : s1[2 + k];
                                s2[k] = mine + __shfl_xor(other, 2, 64);
                            }
                            const float mine = (cq & 1) ? s2[1] : s2[0], other = (cq & 1) ? s2[0] : s2[1];
                            const float tot = mine + __shfl_xor(other, 1, 64);
                            if (row_on) fh.ic_lp_out[(int64_t)bn * fh.ic_lp_stride + (int64_t)node * 8 + cq] = tot;
                        }
                    }
                };
                // (whole waves enter `process`: its trip count is a wave-wide maximum; rows past the group's end are masked inside)
                if (pre) {
#pragma unroll
                    for (int k = 0; k < MAXIT; ++k) {          // static indices: the prefetched operands stay in registers
                        const int slot = (tid >> lqv) + k * (NTH >> lqv);
                        if (k < items && __builtin_amdgcn_readfirstlane(slot - (lane >> lqv)) < cnt) process(slot, ord[gi][k], true, gid[k]);
                        else if (CH && chain_out && k < items && (c >> 4) < fh.ch_KB) {       // a wave of padding slots: zero pieces
                            uint16_t* dstp = fh.ch_pnext + ((int64_t)((grp * 4 + (slot >> 5)) * fh.ch_KB + (c >> 4)) * 2) * 512 + ((slot & 31) + 32 * ((c >> 3) & 1)) * 8;
                            *reinterpret_cast<uint4*>(dstp) = make_uint4(0u, 0u, 0u, 0u);
                            *reinterpret_cast<uint4*>(dstp + 512) = make_uint4(0u, 0u, 0u, 0u);
                        }
                    }
                } else {
                    for (int idx0 = __builtin_amdgcn_readfirstlane(tid & ~63); idx0 < (cnt << lqv); idx0 += NTH)
                    {
                        const int slot = (idx0 + lane) >> lqv;
                        process(slot, (fh.row_order && slot < cnt) ? fh.row_order[ns + slot] : slot, false, 0);
                    }
                }
            }
            if constexpr (CH) {        // the group's per-graph output maxima of this column block: they anchor the next hop's scales
                if (live && chain_out) {    // (block-uniform)
                    __syncthreads();
                    if (tid < ngl) fh.ch_pmout[(int64_t)bn * fh.ch_B + gf + tid] = __uint_as_float(gmax_l[tid]);
                }
            }
        }
        return;
    }
    if constexpr (EPI == 3) {
        // ---- the finished rows leave as the NEXT product's packed A operand (two fp16 pieces, fragment-major, one power-of-two scale
        // per row: exactly what k_split2h_pack would make of them), optionally also as fp32 rows (C).  The workgroup owns whole rows
        // (gridDim.x == 1, N <= 32 FB), so a row's largest magnitude is known here: the 4 column waves of a row meet through LDS.
        // ep.pk_mul: the packed value is the finished value times a per-graph row (the pooling head's ques_nn(u)[batch] * x').
        // No pass over the result between two chained products: the pack kernel's read + write of [M, N] is gone.
        __builtin_amdgcn_s_barrier();                              // main loop done everywhere: the ring is free
        float* mx_l = reinterpret_cast<float*>(smem);              // [FA * 32 rows][WN]
        const int m = lane & 31, h = lane >> 5;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int lr = (wr * TM + i) * 32 + m, gr = bm * FA * 32 + lr;
            const bool on = gr < M;
            const float sa = on ? a_inv[gr] : 0.f;
            const float* mrow = (ep.pk_mul && on) ? ep.pk_mul + (int64_t)ep.pk_mul_idx[gr] * ep.pk_mul_ld : nullptr;
            float mx = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gc = (wc * TN + j) * 32 + 8 * q + 4 * h;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (on && gc < N) {
                        const float4 sb = *reinterpret_cast<const float4*>(b_inv + gc);
                        v = make_float4(acc[i][j][4 * q] * sa * sb.x, acc[i][j][4 * q + 1] * sa * sb.y, acc[i][j][4 * q + 2] * sa * sb.z,
                                        acc[i][j][4 * q + 3] * sa * sb.w);
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
                        if (C) *reinterpret_cast<float4*>(C + (int64_t)gr * ldc + gc) = v;
                        if (mrow) {
                            const float4 m4 = *reinterpret_cast<const float4*>(mrow + gc);
                            v.x *= m4.x; v.y *= m4.y; v.z *= m4.z; v.w *= m4.w;
                        }
                    }
                    acc[i][j][4 * q] = v.x; acc[i][j][4 * q + 1] = v.y; acc[i][j][4 * q + 2] = v.z; acc[i][j][4 * q + 3] = v.w;
                    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (h == 0) mx_l[lr * WN + wc] = mx;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int lr = (wr * TM + i) * 32 + m, rt = bm * FA + wr * TM + i;
            if (rt >= ep.pk_RT) continue;                          // (wave-uniform: a tile past the operand's last one)
            float full = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) full = fmaxf(full, mx_l[lr * WN + w]);
            const int ex = split2h_exponent(full);
            const float scale = pow2i(ex);
            if (wc == 0 && h == 0) ep.pk_inv[rt * 32 + m] = pow2i(-ex);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kb = (wc * TN + j) * 2 + (q >> 1);   // k block of the next product these 4 columns belong to
                    if (kb >= ep.pk_KB) continue;
                    f16x4_t p0, p1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = acc[i][j][4 * q + r] * scale;
                        const _Float16 a = (_Float16)x;
                        p0[r] = a;
                        p1[r] = (_Float16)(x - (float)a);
                    }
                    // lane (m, k half = q & 1) of the fragment holds 8 consecutive k: this thread's 4 are its half h
                    uint16_t* o = ep.pk_out + ((int64_t)(rt * ep.pk_KB + kb) * 2) * 512 + (m + 32 * (q & 1)) * 8 + 4 * h;
                    *reinterpret_cast<uint2*>(o) = __builtin_bit_cast(uint2, p0);
                    *reinterpret_cast<uint2*>(o + 512) = __builtin_bit_cast(uint2, p1);
                }
        }
        return;
    }
    // (pa / pm: the addend / mul values of this quad when the caller fetched them ahead -- the EPI 1 loop issues a tile's four passes'
    //  worth of them before it turns the tile through LDS: written with the loads inside, every quad's 16-byte read was issued, waited
    //  for and consumed alone -- J = XL + [prod | x_ctx] Wj^T of lcgn_seq took 369 us with its addend against 266 without: 183 MB at 1.8 TB/s)
    auto finish = [&](float4 v, int gr, int gc, const float4* pa = nullptr, const float4* pm = nullptr) -> float4 {      // bias / addend / mul / activation on 4 consecutive columns, then the store (C NULL: no store)
        if constexpr (NP == 2) {                       // undo the operands' power-of-two scales (exact)
            const float sa = a_inv[gr];
            const float4 sb = *reinterpret_cast<const float4*>(b_inv + gc);
            v.x = v.x * sa * sb.x; v.y = v.y * sa * sb.y; v.z = v.z * sa * sb.z; v.w = v.w * sa * sb.w;
        }
        if (ep.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(ep.bias + gc);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        }
        if (ep.addend) {
            const float4 a4 = pa ? *pa : *reinterpret_cast<const float4*>(ep.addend + (int64_t)gr * ep.ld_add + gc);
            v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
        }
        if (ep.mul) {
            const float4 m4 = pm ? *pm : *reinterpret_cast<const float4*>(ep.mul + (int64_t)gr * ep.ld_mul + gc);
            v.x *= m4.x; v.y *= m4.y; v.z *= m4.z; v.w *= m4.w;
        }
        if (ep.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (ep.relu == 2) {
            v.x = v.x > 0.f ? v.x : expf(v.x) - 1.f; v.y = v.y > 0.f ? v.y : expf(v.y) - 1.f;
            v.z = v.z > 0.f ? v.z : expf(v.z) - 1.f; v.w = v.w > 0.f ? v.w : expf(v.w) - 1.f;
        }
        if (C) *reinterpret_cast<float4*>(C + (int64_t)gr * ldc + gc) = v;
        return v;
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
        // ep.rowdot_w: the finished row's dot product with a vector, rd[i][ps] = this lane's 4-column share of row (i, ps * 8 + rr)
        // over the wave's TN tiles (the pooling head's 512 -> 1 gate Linear on top of gate_nn's first layer: the layer's output is
        // only ever that dot product's operand)
        float rd[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) rd[i][ps] = 0.f;
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
                float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ep.rowdot_w && gc < N) w4 = *reinterpret_cast<const float4*>(ep.rowdot_w + gc);
                // the tile's addend / mul quads, all four passes in flight together (they travel under the LDS turn below)
                float4 pa4[4], pm4[4];
                if (ep.addend || ep.mul) {
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int gr = (bm * FA + wr * TM + i) * 32 + ps * 8 + rr;
                        const bool ok = gr < M && gc < N;
                        pa4[ps] = (ep.addend && ok) ? *reinterpret_cast<const float4*>(ep.addend + (int64_t)gr * ep.ld_add + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
                        pm4[ps] = (ep.mul && ok) ? *reinterpret_cast<const float4*>(ep.mul + (int64_t)gr * ep.ld_mul + gc) : make_float4(1.f, 1.f, 1.f, 1.f);
                    }
                }
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int r = ps * 8 + rr;
                    const float4 v = *reinterpret_cast<const float4*>(t + r * 128 + ((cc ^ (r & 7)) << 4));
                    const int gr = (bm * FA + wr * TM + i) * 32 + r;
                    if (gr < M && gc < N) {
                        const float4 o = finish(v, gr, gc, ep.addend ? &pa4[ps] : nullptr, ep.mul ? &pm4[ps] : nullptr);
                        rd[i][ps] += (o.x * w4.x + o.y * w4.y) + (o.z * w4.z + o.w * w4.w);
                    }
                }
            }
        if (ep.rowdot_w) {       // (uniform) the 8 lanes of a row -> one partial per (row, wave column): slot bn WN + wc of the row's 16
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    float tsum = rd[i][ps];
                    tsum += __shfl_xor(tsum, 1, 64);
                    tsum += __shfl_xor(tsum, 2, 64);
                    tsum += __shfl_xor(tsum, 4, 64);
                    const int gr = (bm * FA + wr * TM + i) * 32 + ps * 8 + rr;
                    if (cc == 0 && gr < M) ep.rowdot_out[(int64_t)gr * 16 + bn * WN + wc] = tsum;
                }
        }
    }
}

// packed operand of `row_tiles` 32-row tiles: the fragments, then (two-piece form) one inverse scale per row
size_t split_packed_rows_bytes(int np, int64_t row_tiles, int64_t K) {
    return (size_t)row_tiles * (size_t)cdiv(K, 16) * (size_t)np * 1024 + (np == 2 ? (size_t)row_tiles * 32 * sizeof(float) : 0);
}
size_t split_packed_bytes(int np, int64_t rows, int64_t K) { return split_packed_rows_bytes(np, cdiv(rows, 32), K); }
static const float* split2h_inv_scales(const void* packed, int64_t row_tiles, int KB) {
    return reinterpret_cast<const float*>(static_cast<const char*>(packed) + (size_t)row_tiles * KB * 2048);
}


// ---- two-piece form ("split2h"): fp32 value x of a row with largest magnitude m is carried as two fp16 pieces of
// x 2^e, e = 13 - floor(log2 m) (the row's largest magnitude lands in [2^13, 2^14): no overflow, and 27 binades of fp16
// below it): p1 = RN16(x 2^e), p2 = RN16(x 2^e - p1).  |x 2^e - p1 - p2| <= 2^-22 |x 2^e| (2^-23 typical) while p2 is a
// normal fp16, <= 2^-25 in absolute terms (2^-38 of the row's largest magnitude) below that.  A dot product keeps
// p1 q1 + p1 q2 + p2 q1 (exact products, fp32 accumulation by `v_mfma_f32_32x32x16_f16`); the dropped p2 q2 is <= 2^-22 |x y|.
// Against fp64 the result is as close as the k-ordered fp32 chain of the f32 MFMA or three-piece bf16 (whose errors are the
// fp32 accumulation's, tests/test_gpu_split3.py) at half the matrix-core work of the latter.  The inverse scale 2^-e of every
// row is stored behind the fragments and applied (exactly) to the accumulators.
__device__ __forceinline__ void split2h_store(const float (&v)[8], float scale, uint16_t* o, f16x8_t& p0, f16x8_t& p1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e] * scale;
        const _Float16 a = (_Float16)x;
        p0[e] = a;
        p1[e] = (_Float16)(x - (float)a);
    }
    *reinterpret_cast<uint4*>(o) = __builtin_bit_cast(uint4, p0);
    *reinterpret_cast<uint4*>(o + 512) = __builtin_bit_cast(uint4, p1);
}

// which fp32 row a packed row (slot) holds
enum { PACK_PLAIN = 0, PACK_GROUPS = 1, PACK_HEADS = 2, PACK_HEADS2 = 3, PACK_GATHER = 4 };
struct PackRows {
    const float* X; int64_t ld;
    int64_t rows;                 // PLAIN: packed row r = X row r
    const int32_t* group_ptr;     // GROUPS: slot 128 g + i = node group_ptr[g] + i
    int H, C, cw;                 // HEADS: packed row 256 cb + h cw + cc = W row h C + cb cw + cc
    // GATHER (rows as PLAIN): the packed value is relu(X[r, k] + ga[gia[r], k] + gb[gib[r], k] + gbias[k]) -- the first Linear of an
    // edge-level MLP of the scene-graph encoder with its node-side column blocks gathered on the way into the operand (gb may be NULL)
    const float* ga; const int64_t* gia; int64_t glda;
    const float* gb; const int64_t* gib; int64_t gldb;
    const float* gbias;
    // GATHER, optional: row r of X is itself a gather, X[r, :] = (gneg[r] ? -1 : 1) * T[clamp(gtok[r], 0, gV - 1), :] with T = pr.X
    // (the one-token embedding "sum" of an edge against the projected table: k_embed_sum's pass folded into this one)
    const int64_t* gtok; int gV; const uint8_t* gneg;
    // GATHER, optional: gop = 1 -- the packed value is X[r, k] * ga[gi32[r], k] (rows scaled by a per-graph row: the pooling head's
    // ques_nn(u)[batch] * x', pipeline_model_gat.py:165, formed on the way into gate_nn's operand); gia NULL, gi32 the int32 index
    const int32_t* gi32; int gop;
    // optional: the rows' largest magnitudes leave as GVQA_ABSMAX_SLOTS slice maxima (slot = row tile mod slots; the caller zeroes them): a later
    // consumer that wants ONE scale for the whole operand -- the weight-gradient product of the backward -- then needs no pass of its own
    float* absmax;
};                                // HEADS2 (hop2.hip): packed row 256 cb + 64 w + 32 j + t = W row h C + cb cw + j hw + cc, (h, cc) = divmod(32 w + t, hw), hw = cw / 2
// head and channel of row `within` (0..255) of column block cb
template <int MAP>
__device__ __forceinline__ void heads_row(const PackRows& pr, int cb, int within, int& h, int& ch) {
    if constexpr (MAP == PACK_HEADS2) {
        const int hw = pr.cw >> 1, u = 32 * (within >> 6) + (within & 31);
        h = u / hw;
        ch = cb * pr.cw + ((within >> 5) & 1) * hw + (u - h * hw);
    } else {
        h = within / pr.cw;
        ch = cb * pr.cw + (within - h * pr.cw);
    }
}
template <int MAP>
__device__ __forceinline__ const float* pack_row(const PackRows& pr, int64_t rt, int m, bool& on, int64_t& src_row) {
    if constexpr (MAP == PACK_PLAIN || MAP == PACK_GATHER) {
        src_row = rt * 32 + m;
        on = src_row < pr.rows;
    } else if constexpr (MAP == PACK_GROUPS) {
        const int grp = (int)(rt >> 2), i = (int)(rt & 3) * 32 + m;
        const int ns = pr.group_ptr[grp], cnt = pr.group_ptr[grp + 1] - ns;
        on = i < cnt;
        src_row = ns + i;
    } else {
        const int r = (int)(rt * 32) + m;
        int h, ch;
        heads_row<MAP>(pr, r >> 8, r & 255, h, ch);
        on = ch < pr.C;
        src_row = (int64_t)h * pr.C + ch;
    }
    if (!on) src_row = 0;
    return pr.X + src_row * pr.ld;
}
__device__ __forceinline__ void load_row8(const float* row, bool on, int k0, int K, int vec, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (!on) return;
    if (vec && k0 + 8 <= K) {
        const float4 a = *reinterpret_cast<const float4*>(row + k0), b = *reinterpret_cast<const float4*>(row + k0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k0 + e < K) v[e] = row[k0 + e];
    }
}

// GATHER map: v = relu(v + a_row[k0..] + b_row[k0..] + bias[k0..]) on the 8 values just loaded (K % 4 == 0, 16-byte rows)
__device__ __forceinline__ void gather_add_relu8(const float* arow, const float* brow, const float* bias, bool on, int k0, int K, float (&v)[8]) {
    if (!on) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = k0 + 4 * h;
        if (k >= K) { v[4 * h] = v[4 * h + 1] = v[4 * h + 2] = v[4 * h + 3] = 0.f; continue; }
        const float4 a = *reinterpret_cast<const float4*>(arow + k), bi = *reinterpret_cast<const float4*>(bias + k);
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (brow) b = *reinterpret_cast<const float4*>(brow + k);
        v[4 * h + 0] = fmaxf(v[4 * h + 0] + a.x + b.x + bi.x, 0.f);
        v[4 * h + 1] = fmaxf(v[4 * h + 1] + a.y + b.y + bi.y, 0.f);
        v[4 * h + 2] = fmaxf(v[4 * h + 2] + a.z + b.z + bi.z, 0.f);
        v[4 * h + 3] = fmaxf(v[4 * h + 3] + a.w + b.w + bi.w, 0.f);
    }
}

// GATHER map, gop = 1: v *= a_row[k0..]
__device__ __forceinline__ void gather_mul8(const float* arow, bool on, int k0, int K, float (&v)[8]) {
    if (!on) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = k0 + 4 * h;
        if (k >= K) { v[4 * h] = v[4 * h + 1] = v[4 * h + 2] = v[4 * h + 3] = 0.f; continue; }
        const float4 a = *reinterpret_cast<const float4*>(arow + k);
        v[4 * h + 0] *= a.x; v[4 * h + 1] *= a.y; v[4 * h + 2] *= a.z; v[4 * h + 3] *= a.w;
    }
}

// One block (4 waves) per 32-row tile; wave w walks the k-block PAIRS 2w, 2w + 1, 2w + 8, 2w + 9, ... (a pair is one 128-byte
// line of every row: its four 16-byte-per-lane loads come from the same wave back to back).  Pass 1 finds the rows' largest magnitudes (two k
// halves of a wave by lane ^ 32, the four waves through LDS), pass 2 writes the scaled pieces.  NIT > 0: a thread's (at most)
// NIT k blocks stay in registers between the passes (rows read once); NIT == 0: any K, rows read twice.  J > 0 (GROUPS):
// a_node[node, j] = sum_k x[node, k] Vn[j, k] on the way.  LM = 0: fp32 FMAs against Vn in LDS, as in
// k_split3_pack_groups_logits (one LDS read per FMA: 20 us of a 70 us pass at config 3).  LM = 1: on the matrix cores -- the
// pieces this lane has just made ARE the A fragment of a 32x32x16 MFMA; against the two-piece image of Vn (`vn_pk`: one 32-row
// tile, rows >= J zero, packed like any weight operand and cached with the weights) three MFMAs per k block give the tile's
// logits in the projection's own arithmetic, rescaled exactly by the two inverse scales.
// NWV waves per workgroup (4, or 8 for the hop operand at K <= 512: half the k blocks -- and data registers -- per thread, twice
// the waves in flight per CU).
template <int MAP, int J, int NIT, int LM, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void k_split2h_pack(PackRows pr, int K, int KB, uint16_t* __restrict__ out, float* __restrict__ inv_scale,
                                                      int vec, const float* __restrict__ Vn, float* __restrict__ a_node,
                                                      const uint16_t* __restrict__ vn_pk, const float* __restrict__ vn_inv) {
    extern __shared__ __attribute__((aligned(16))) float pk_s[];      // [NWV][32] row maxima | LM 0: [J][Kp] Vn | [NWV][32][J] partial dots
    constexpr int NTHR = 64 * NWV;
    constexpr int JJ = (J > 0 && !LM) ? J : 1, NR = NIT > 0 ? NIT : 1;
    const int Kp = KB * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* mx_s = pk_s;
    float* vn_s = pk_s + 32 * NWV;
    if (J > 0 && !LM) {
        for (int idx = tid; idx < J * Kp; idx += NTHR) {
            const int j = idx / Kp, k = idx - j * Kp;
            vn_s[idx] = k < K ? Vn[(int64_t)j * K + k] : 0.f;
        }
    }
    const int64_t rt = blockIdx.x;
    bool row_on;
    int64_t src_row;
    const float* row = pack_row<MAP>(pr, rt, lane & 31, row_on, src_row);
    const int kh = (lane >> 5) * 8;
    float v[NR][8];
    float mx = 0.f;
    auto kb_of = [&](int it) { return (it >> 1) * (2 * NWV) + 2 * wave + (it & 1); };
    const int nit_all = ((KB + 2 * NWV - 1) / (2 * NWV)) * 2;         // iterations that cover every k block (NIT == 0 path)
    [[maybe_unused]] const float* g_arow = nullptr;
    [[maybe_unused]] const float* g_brow = nullptr;
    [[maybe_unused]] float g_sign = 1.f;
    if constexpr (MAP == PACK_GATHER) {                               // (launched with NIT > 0 only: the rows stay in registers)
        g_arow = pr.ga + (row_on ? (pr.gia ? pr.gia[src_row] : (int64_t)pr.gi32[src_row]) : 0) * pr.glda;
        g_brow = pr.gb ? pr.gb + (row_on ? pr.gib[src_row] : 0) * pr.gldb : nullptr;
        if (pr.gtok) {
            int64_t id = row_on ? pr.gtok[src_row] : 0;
            id = id < 0 ? 0 : (id >= pr.gV ? pr.gV - 1 : id);
            row = pr.X + id * pr.ld;
            g_sign = (pr.gneg && row_on && pr.gneg[src_row]) ? -1.f : 1.f;
        }
    }
    if (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NR; ++it) {
            const int kb = kb_of(it);
            load_row8(row, row_on && kb < KB, kb * 16 + kh, K, vec, v[it]);
            if constexpr (MAP == PACK_GATHER) {
                if (pr.gtok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[it][e] *= g_sign;
                }
                if (pr.gop == 1) gather_mul8(g_arow, row_on && kb < KB, kb * 16 + kh, K, v[it]);
                else gather_add_relu8(g_arow, g_brow, pr.gbias, row_on && kb < KB, kb * 16 + kh, K, v[it]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[it][e]));
        }
    } else {
        for (int it = 0; it < nit_all; ++it) {
            const int kb = kb_of(it);
            if (kb >= KB) continue;
            load_row8(row, row_on, kb * 16 + kh, K, vec, v[0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[0][e]));
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lane < 32) mx_s[wave * 32 + lane] = mx;
    __syncthreads();                                                  // (also: Vn is in LDS)
    const int m = lane & 31;
    float rowmax = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) rowmax = fmaxf(rowmax, mx_s[w * 32 + m]);
    if constexpr (MAP == PACK_HEADS || MAP == PACK_HEADS2) {
        // ONE scale per 256-row column block of the fused hop (its epilogue then needs a single factor for all its columns):
        // the largest magnitude of the block's rows, found by every tile of the block on its own (weights: packed once, cached).
        // Rows 2^-16 below their block's largest lose the 2^-22 guarantee (absolute error <= 2^-38 of the block's largest).
        float bmx = 0.f;
        const int cb = (int)(rt >> 3);
        for (int r = tid >> 6; r < 256; r += NWV) {                   // a wave per row, lanes over k
            int h, ch;
            heads_row<MAP>(pr, cb, r, h, ch);
            if (ch >= pr.C) continue;
            const float* wrow = pr.X + ((int64_t)h * pr.C + ch) * pr.ld;
            for (int k = lane; k < K; k += 64) bmx = fmaxf(bmx, fabsf(wrow[k]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bmx = fmaxf(bmx, __shfl_xor(bmx, o, 64));
        __syncthreads();                                              // every read of mx_s above is done
        if (lane == 0) mx_s[wave] = bmx;
        __syncthreads();
        rowmax = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) rowmax = fmaxf(rowmax, mx_s[w]);
    }
    if (pr.absmax && wave == 0) {                                     // one atomic per workgroup, GVQA_ABSMAX_SLOTS addresses
        float t = rowmax;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) t = fmaxf(t, __shfl_xor(t, o, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(pr.absmax) + (rt & (GVQA_ABSMAX_SLOTS - 1)), __float_as_uint(t));
    }
    const int ex = split2h_exponent(rowmax);
    const float scale = pow2i(ex);
    if (tid < 32) inv_scale[rt * 32 + tid] = pow2i(-ex);
    float acc[JJ];
#pragma unroll
    for (int j = 0; j < JJ; ++j) acc[j] = 0.f;
    f32x16 lacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
    auto emit = [&](int kb, const float (&w)[8]) {
        f16x8_t p0, p1;
        split2h_store(w, scale, out + ((rt * KB + kb) * 2) * 512 + lane * 8, p0, p1);
        if constexpr (J > 0 && LM) {
            const uint16_t* vb = vn_pk + (int64_t)kb * 1024 + lane * 8;
            const f16x8_t q0 = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(vb));
            const f16x8_t q1 = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(vb + 512));
            lacc = split_mfma(q1, p0, lacc);                      // transposed accumulators, as in the GEMM: lane (m, hh) holds
            lacc = split_mfma(q0, p1, lacc);                      // columns 8 q + 4 hh + 0..3 of row m in registers 4 q + 0..3
            lacc = split_mfma(q0, p0, lacc);
        } else if (J > 0) {
            const int k0 = kb * 16 + kh;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float4 w0 = *reinterpret_cast<const float4*>(vn_s + j * Kp + k0), w1 = *reinterpret_cast<const float4*>(vn_s + j * Kp + k0 + 4);
                acc[j] += w[0] * w0.x + w[1] * w0.y + w[2] * w0.z + w[3] * w0.w + w[4] * w1.x + w[5] * w1.y + w[6] * w1.z + w[7] * w1.w;
            }
        }
    };
    if (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NR; ++it)
            if (kb_of(it) < KB) emit(kb_of(it), v[it]);
    } else {
        for (int it = 0; it < nit_all; ++it) {
            const int kb = kb_of(it);
            if (kb >= KB) continue;
            load_row8(row, row_on, kb * 16 + kh, K, vec, v[0]);
            emit(kb, v[0]);
        }
    }
    if (J > 0) {
        float* part = LM ? vn_s : vn_s + J * Kp;                      // [4 waves][32 rows][J]
        if constexpr (LM) {
            const int hh = lane >> 5;
#pragma unroll
            for (int q = 0; q < (J + 7) / 8; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 8 * q + 4 * hh + r;
                    if (j < J) part[(wave * 32 + m) * J + j] = lacc[4 * q + r];
                }
            __syncthreads();                                          // every read of the row maxima is done, too
            if (tid < 32) mx_s[tid] = pow2i(-ex);                     // this row's inverse scale, for the threads that finish it
        } else {
#pragma unroll
            for (int j = 0; j < JJ; ++j) {
                const float t = acc[j] + __shfl_xor(acc[j], 32, 64);  // the two k halves of the row
                if (lane < 32) part[(wave * 32 + lane) * J + j] = t;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 32 * J; idx += NTHR) {
            const int r = idx / J, j = idx - r * J;
            bool on2;
            int64_t node;
            pack_row<MAP>(pr, rt, r, on2, node);
            if (on2) {
                float t = (part[(0 * 32 + r) * J + j] + part[(1 * 32 + r) * J + j]) + (part[(2 * 32 + r) * J + j] + part[(3 * 32 + r) * J + j]);
                if constexpr (NWV == 8)
                    t += (part[(4 * 32 + r) * J + j] + part[(5 * 32 + r) * J + j]) + (part[(6 * 32 + r) * J + j] + part[(7 * 32 + r) * J + j]);
                if constexpr (LM) t = t * mx_s[r] * vn_inv[j];
                a_node[node * J + j] = t;
            }
        }
    }
}

static size_t split2h_pack_lds(int J, int KB, bool mfma_logits = false, int nwv = 4) {
    return (32 * (size_t)nwv + (mfma_logits ? 0 : (size_t)J * KB * 16) + (size_t)nwv * 32 * (size_t)J) * sizeof(float);
}

// Vn_packed: NULL, or the two-piece image of Vn [J, K] (gvqa layout of one 32-row tile + its inverse scales): logits on the
// matrix cores (LM = 1) instead of fp32 FMAs against the fp32 Vn
template <int MAP>
static int launch_split2h_pack_tiles(const PackRows& pr, int64_t RT, int64_t K, void* packed, const float* Vn, int J, float* a_node,
                                     hipStream_t stream, const void* Vn_packed = nullptr) {
    const int KB = (int)cdiv(K, 16);
    const int vec = (reinterpret_cast<uintptr_t>(pr.X) & 15) == 0 && pr.ld % 4 == 0;
    uint16_t* o = static_cast<uint16_t*>(packed);
    float* inv = const_cast<float*>(split2h_inv_scales(packed, RT, KB));
    const bool lm = J > 0 && Vn_packed != nullptr;
    // eight waves per tile for the hop operand with MFMA logits at K <= 512 (GVQA_PACK_WAVES=4: the four-wave form, for the A/B)
    static const bool waves8_ok = []() { const char* v = getenv("GVQA_PACK_WAVES"); return !(v && v[0] == '4'); }();
    const bool w8 = lm && MAP == PACK_GROUPS && KB <= 32 && waves8_ok;
    const size_t lds = split2h_pack_lds(J, KB, lm, w8 ? 8 : 4);
    const dim3 grid((unsigned)RT), block(w8 ? 512 : 256);
    const uint16_t* vpk = static_cast<const uint16_t*>(Vn_packed);
    const float* vinv = lm ? split2h_inv_scales(Vn_packed, 1, KB) : nullptr;
#define GVQA_P2L(J_, NIT_, LM_) hipLaunchKernelGGL((k_split2h_pack<MAP, J_, NIT_, LM_>), grid, block, lds, stream, pr, (int)K, KB, o, inv, vec, Vn, \
                                                   a_node, vpk, vinv)
#define GVQA_P2W8(J_) hipLaunchKernelGGL((k_split2h_pack<MAP, J_, 4, 1, 8>), grid, block, lds, stream, pr, (int)K, KB, o, inv, vec, Vn, a_node, vpk, vinv)
#define GVQA_P2(J_, NIT_) do { if (w8) GVQA_P2W8(J_); else if (lm) GVQA_P2L(J_, NIT_, 1); else GVQA_P2L(J_, NIT_, 0); } while (0)
    if (J == 0) {
        if (KB <= 32) GVQA_P2L(0, 8, 0); else if (KB <= 64) GVQA_P2L(0, 16, 0); else GVQA_P2L(0, 0, 0);
    } else if constexpr (MAP == PACK_GROUPS) {
        const bool regs = KB <= 32;
        switch (J) {
            case 2: if (regs) GVQA_P2(2, 8); else GVQA_P2(2, 0); break;
            case 4: if (regs) GVQA_P2(4, 8); else GVQA_P2(4, 0); break;
            case 8: if (regs) GVQA_P2(8, 8); else GVQA_P2(8, 0); break;
            case 16: if (regs) GVQA_P2(16, 8); else GVQA_P2(16, 0); break;
            default: return GVQA_E_UNSUPPORTED;
        }
    } else if constexpr (MAP == PACK_PLAIN) {            // (plain rows + their J = 8 logits: the training forward's operand pack, gvqa_split2h_pack_logits)
        if (J != 8 || w8 || lm) return GVQA_E_UNSUPPORTED;
        if (KB <= 32) GVQA_P2L(8, 8, 0); else GVQA_P2L(8, 0, 0);
    } else {
        return GVQA_E_UNSUPPORTED;
    }
#undef GVQA_P2
#undef GVQA_P2L
#undef GVQA_P2W8
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// The two-piece pack of relu(Y + a[ia] + b[ib] + bias) (PackRows GATHER): K % 4 == 0, K <= 512, 16-byte aligned rows
int launch_split2h_pack_gather(int64_t rows, int64_t K, const float* Y, int64_t ld, const float* a, const int64_t* ia, int64_t lda,
                               const float* b, const int64_t* ib, int64_t ldb, const float* bias, void* packed, hipStream_t stream,
                               const int64_t* tok, int V, const uint8_t* neg) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    GVQA_REQUIRE(rows >= 0 && K > 0 && K % 4 == 0 && K <= 512 && ld % 4 == 0 && ld >= K && lda % 4 == 0 && (!b || ldb % 4 == 0), GVQA_E_UNSUPPORTED,
                 "split_pack_gather: K %% 4 == 0, K <= 512, row strides multiples of 4");
    if (rows == 0) return GVQA_OK;
    GVQA_REQUIRE(Y && a && ia && bias && packed && (!b || ib) && al(Y) && al(a) && al(b) && al(bias) && al(packed), GVQA_E_INVALID,
                 "split_pack_gather: null / unaligned operand");
    const int64_t RT = cdiv(rows, 32);
    GVQA_REQUIRE(RT < (1ll << 31), GVQA_E_INVALID, "split_pack_gather: too many rows");
    GVQA_REQUIRE(!tok || V > 0, GVQA_E_INVALID, "split_pack_gather: token rows need the table's row count");
    PackRows pr{Y, ld, rows, nullptr, 0, 0, 0, a, ia, lda, b, ib, ldb, bias, tok, V, neg};
    return launch_split2h_pack_tiles<PACK_GATHER>(pr, RT, K, packed, nullptr, 0, nullptr, stream);
}

// The two-piece pack of X[r, :] * R[idx[r], :] (PackRows GATHER, gop = 1): K % 4 == 0, K <= 512, 16-byte aligned rows
int launch_split2h_pack_rowmul(int64_t rows, int64_t K, const float* X, int64_t ld, const float* R, const int32_t* idx, int64_t ldr, void* packed,
                               hipStream_t stream) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    GVQA_REQUIRE(rows >= 0 && K > 0 && K % 4 == 0 && K <= 512 && ld % 4 == 0 && ld >= K && ldr % 4 == 0 && ldr >= K, GVQA_E_UNSUPPORTED,
                 "split_pack_rowmul: K %% 4 == 0, K <= 512, row strides multiples of 4");
    if (rows == 0) return GVQA_OK;
    GVQA_REQUIRE(X && R && idx && packed && al(X) && al(R) && al(packed), GVQA_E_INVALID, "split_pack_rowmul: null / unaligned operand");
    const int64_t RT = cdiv(rows, 32);
    GVQA_REQUIRE(RT < (1ll << 31), GVQA_E_INVALID, "split_pack_rowmul: too many rows");
    PackRows pr{X, ld, rows, nullptr, 0, 0, 0, R, nullptr, ldr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, idx, 1};
    return launch_split2h_pack_tiles<PACK_GATHER>(pr, RT, K, packed, nullptr, 0, nullptr, stream);
}

int launch_split_pack(int np, int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, hipStream_t stream, float* absmax, const float* Vn, int J,
                      float* a_node) {
    GVQA_REQUIRE(np == 2 || np == 3, GVQA_E_INVALID, "split_pack: 2 or 3 pieces");
    GVQA_REQUIRE(rows >= 0 && K >= 0 && K < (1ll << 30) && ld >= K, GVQA_E_INVALID, "split_pack: bad size");
    if (rows == 0 || K == 0) return GVQA_OK;
    GVQA_REQUIRE(X && packed, GVQA_E_INVALID, "split_pack: null operand");
    GVQA_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, GVQA_E_INVALID, "split_pack: packed buffer must be 16-byte aligned");
    const int KB = (int)cdiv(K, 16);
    const int64_t RT = cdiv(rows, 32);
    const int vec = (reinterpret_cast<uintptr_t>(X) & 15) == 0 && ld % 4 == 0;
    if (np == 2) {
        GVQA_REQUIRE(RT < (1ll << 31), GVQA_E_INVALID, "split_pack: too many rows");
        PackRows pr{X, ld, rows, nullptr, 0, 0, 0};
        pr.absmax = absmax;
        return launch_split2h_pack_tiles<PACK_PLAIN>(pr, RT, K, packed, Vn, Vn ? J : 0, a_node, stream);
    }
    GVQA_REQUIRE(!absmax && !Vn, GVQA_E_UNSUPPORTED, "split_pack: slice maxima / logits are a two-piece pack's by-products");
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

// batch > 1: gridDim.z batches with the strides of ep.zs_* (split-K chunks of the TN product); a_inv_batched / b_inv_batched:
// the operands' inverse scales when they do not sit behind the fragments (two-piece operands)
int launch_linear_split(int np, int64_t M, int64_t N, int64_t K, const void* Apk, const void* Bpk, LinearEpilogue ep, float* C,
                        int64_t ldc, hipStream_t stream, int batch, const float* a_inv_batched, const float* b_inv_batched) {
    GVQA_REQUIRE(np == 2 || np == 3, GVQA_E_INVALID, "linear_split: 2 or 3 pieces");
    GVQA_REQUIRE(M >= 0 && N >= 0 && K > 0, GVQA_E_INVALID, "linear_split3: bad size");
    GVQA_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 30), GVQA_E_INVALID, "linear_split3: size overflow");
    if (M == 0 || N == 0) return GVQA_OK;
    GVQA_REQUIRE(Apk && Bpk && (C || ep.rowdot_w || ep.pk_out), GVQA_E_INVALID, "linear_split3: null operand");
    if (ep.pk_out) {      // the result as the next product's packed operand: whole rows per workgroup (128 x 512 tile)
        GVQA_REQUIRE(np == 2 && batch == 1 && !ep.rowdot_w && ep.relu != 2 && M <= 65535ll * 128, GVQA_E_INVALID,
                     "linear_split: packed output takes two-piece operands, one batch, bias / addend / ReLU epilogues");
        if (N > 512) return GVQA_E_UNSUPPORTED;
        GVQA_REQUIRE((reinterpret_cast<uintptr_t>(ep.pk_out) & 15) == 0 && (!ep.pk_mul || (ep.pk_mul_idx && ep.pk_mul_ld % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(ep.pk_mul) & 15) == 0)), GVQA_E_INVALID, "linear_split: packed output / row multiplier alignment");
    }
    GVQA_REQUIRE(!ep.rowdot_w || (ep.rowdot_out && np == 2 && batch == 1 && (reinterpret_cast<uintptr_t>(ep.rowdot_w) & 15) == 0), GVQA_E_INVALID,
                 "linear_split: the row-dot epilogue needs its output, two-piece operands, one batch, a 16-byte aligned vector");
    GVQA_REQUIRE(ldc >= N && (!ep.addend || ep.ld_add >= N) && (!ep.mul || ep.ld_mul >= N), GVQA_E_INVALID,
                 "linear_split3: leading dimension too small");
    GVQA_REQUIRE(linear_split3_supported(N, ep, C, ldc), GVQA_E_UNSUPPORTED,
                 "linear_split3: N %% 4 == 0 and 16-byte aligned C / addend / mul / bias rows required");
    const int KB = (int)cdiv(K, 16), rtB = (int)cdiv(N, 32);
    const uint16_t* a = static_cast<const uint16_t*>(Apk);
    const uint16_t* b = static_cast<const uint16_t*>(Bpk);
    int variant = split3_variant(M, N, KB);            // (a forced variant >= 100 names a two-piece instantiation)
    const int forced_variant = ep.pk_out ? 140 : 0;
    if (np == 2 && variant < 100) variant = variant < 20 ? ((KB & 1) ? 118 : 112) : 134;      // (112: two K steps per barrier)
    // narrow results whose last 256-column tile would be at most half full (N = 300: 2 x 256 columns computed for 300) take
    // 128-column tiles instead (3 x 128): a quarter less matrix-core work at a lower arithmetic intensity per tile
    if (np == 2 && get_option(GVQA_OPT_SPLIT3_VARIANT) < 10 && N % 256 != 0 && N % 256 <= 128 && N <= 1024) variant = 124;
    if (forced_variant) variant = forced_variant;
    const int KBA = ep.a2 ? ep.a2_kb0 : KB;            // k blocks of the (first) A image
    if (ep.a2) {
        // two K segments of A: the plain and the two-steps-per-barrier loops know the switch (not the read-ahead loop)
        GVQA_REQUIRE(np == 2 && batch == 1 && ep.a2_inv && ep.a2_kb0 > 0 && ep.a2_KB > 0 && ep.a2_kb0 + ep.a2_KB == KB && M <= 65535ll * 128 &&
                     (reinterpret_cast<uintptr_t>(ep.a2) & 15) == 0, GVQA_E_INVALID, "linear_split: bad second K segment of A");
        if (!ep.pk_out) variant = ((KB & 1) == 0 && (ep.a2_kb0 & 1) == 0) ? 112 : 114;
    }
    GVQA_REQUIRE((variant >= 100) == (np == 2), GVQA_E_INVALID, "linear_split: variant %d does not take %d-piece operands", variant, np);
    const int64_t bm = variant % 100 < 20 ? 256 : 128;     // (rows per tile: 1x = 256, 2x / 3x / 4x = 128)
    GVQA_REQUIRE(batch >= 1 && batch <= 65535 && (batch == 1 || M <= 65535 * bm), GVQA_E_INVALID, "linear_split: bad batch count");
    const float* a_inv = np == 2 ? (a_inv_batched ? a_inv_batched : split2h_inv_scales(Apk, cdiv(M, 32), KBA)) : nullptr;
    const float* b_inv = np == 2 ? (b_inv_batched ? b_inv_batched : split2h_inv_scales(Bpk, rtB, KB)) : nullptr;
    {   // measurement switch GVQA_PK_DIRECT (default 0; 1 = widths whose 256-column tiles are more than half full, 2 = every width): the plain
        // product in the direct kernels' step layout (tn_direct.hip: three-slot ring of DMA'd fragment pairs, barrier mid-step, epilogue through
        // LDS).  Stand-alone it is 2-10 % faster than k_linear_split3 on this path's shapes (scripts/bench_pk_direct.py), inside the training step
        // and the LCGN forward it is not (12.77 vs 12.63 ms, 2.97 vs 2.97 ms: profiles/r05_pk_direct_ab.txt) -- off; GPU tier green with it on
        static const int pk_direct = [] { const char* v = getenv("GVQA_PK_DIRECT"); return v ? atoi(v) : 0; }();
        if (pk_direct && np == 2 && batch == 1 && !ep.pk_out && !ep.rowdot_w && !ep.a2 && C && M >= 1024 && N >= 128 &&
            (pk_direct == 2 || N % 256 == 0 || N % 256 > 128) && get_option(GVQA_OPT_SPLIT3_VARIANT) == 0 &&
            cdiv(M, 32) * (int64_t)KB * 2048 < (1ll << 32))
            return launch_linear_pk_direct(M, N, KB, Apk, a_inv, Bpk, b_inv, ep, C, ldc, stream);
    }
#ifdef GVQA_PROBES
    const char* ssv = getenv("GVQA_SPLIT3_STAGGER");      // quarter units of the default start offset (tuning aid)
    const int stag_scale = ssv ? atoi(ssv) : 0;
    const char* ldv = getenv("GVQA_SPLIT3_LOOP_DEBUG");   // measurement aid: see the main loop
    const int loop_dbg = ldv ? atoi(ldv) : 0;
#else
    constexpr int stag_scale = 0, loop_dbg = 0;
#endif
    const int64_t rows_per_launch = 65535 * bm;        // grid.y limit: row chunks (rows are independent)
    for (int64_t m0 = 0; m0 < M; m0 += rows_per_launch) {
        const int64_t m = std::min(rows_per_launch, M - m0);
        LinearEpilogue e2 = ep;
        if (ep.addend) e2.addend = ep.addend + m0 * ep.ld_add;
        if (ep.mul) e2.mul = ep.mul + m0 * ep.ld_mul;
        if (ep.rowdot_out) e2.rowdot_out = ep.rowdot_out + m0 * 16;
        if (ep.pk_out) {                                   // (one launch: M <= 65535 x 128 checked above)
            e2.pk_KB = (int)cdiv(N, 16);
            e2.pk_RT = (int)cdiv(M, 32);
            e2.pk_inv = const_cast<float*>(split2h_inv_scales(ep.pk_out, e2.pk_RT, e2.pk_KB));
        }
        const uint16_t* a2 = a + (m0 / 32) * (int64_t)KBA * (np * 512);
        const float* a_inv2 = a_inv ? a_inv + m0 : nullptr;
        const int rt2 = (int)cdiv(m, 32);
#define GVQA_S3_LAUNCH(WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_) GVQA_SP_LAUNCH(WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_, 3, 0)
#define GVQA_SN_LAUNCH(WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_, NP_) GVQA_SP_LAUNCH(WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_, NP_, 0)
#define GVQA_SP_LAUNCH(WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_, NP_, PIPE_) GVQA_SK_LAUNCH(WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_, NP_, PIPE_, 1)
#define GVQA_SK_LAUNCH(WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_, NP_, PIPE_, KS_)                           \
        do {                                                                                                             \
            dim3 grid((unsigned)cdiv(N, 32 * WN_ * TN_), (unsigned)cdiv(m, 32 * WM_ * TM_), (unsigned)batch);             \
            if (e2.rowdot_w && (EPI_ != 1 || grid.x * WN_ > 16)) return GVQA_E_UNSUPPORTED;   /* 16 partial slots per row */     \
            /* a block's MFMA issue time x the STAG_ blocks sharing the SIMDs, split into STAG_ start offsets */         \
            const int stag = STAG_ > 0 ? (int)((int64_t)KB * 6 * TM_ * TN_ * 32 / 8128) : 0;                              \
            hipLaunchKernelGGL((k_linear_split3<WM_, WN_, TM_, TN_, NBUF_, ILV_, PRIO_, NOST_, STAG_, EPI_, 0, NP_, PIPE_, KS_>), grid, \
                               dim3(64 * WM_ * WN_), 0, stream, (int)m, (int)N, KB, a2, rt2, b, rtB, e2, C ? C + m0 * ldc : nullptr, ldc, \
                               STAG_ == 0 ? loop_dbg : (stag_scale > 0 ? stag * stag_scale / 4 : stag), FusedHopArgs{},   \
                               a_inv2, b_inv);                                                                           \
        } while (0)
        switch (variant) {
            case 10: GVQA_S3_LAUNCH(2, 4, 4, 2, 3, false, false, false, 0, 0); break;
            case 11: GVQA_S3_LAUNCH(2, 4, 4, 2, 3, true, false, false, 0, 0); break;
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 13: GVQA_S3_LAUNCH(2, 4, 4, 2, 3, true, false, true, 0, 0); break;
#endif
            case 20: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, false, false, false, 0, 0); break;
            case 21: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, true, false, false, 0, 0); break;
            case 22: GVQA_S3_LAUNCH(2, 2, 2, 2, 2, true, false, false, 0, 0); break;
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 23: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, true, false, true, 0, 0); break;
#endif
            case 26: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, true, false, false, 2, 0); break;
            case 27: GVQA_S3_LAUNCH(2, 2, 2, 2, 2, true, false, false, 3, 0); break;
            case 14: GVQA_S3_LAUNCH(2, 4, 4, 2, 3, true, false, false, 0, 1); break;
            case 28: GVQA_S3_LAUNCH(2, 2, 2, 2, 3, true, false, false, 0, 1); break;
            case 29: GVQA_S3_LAUNCH(2, 2, 2, 2, 2, true, false, false, 0, 1); break;
            case 30: GVQA_S3_LAUNCH(2, 2, 2, 4, 2, true, false, false, 0, 0); break;
            case 34: GVQA_S3_LAUNCH(2, 2, 2, 4, 2, true, false, false, 0, 1); break;
            case 31: GVQA_S3_LAUNCH(2, 2, 2, 4, 2, true, false, false, 2, 0); break;
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 33: GVQA_S3_LAUNCH(2, 2, 2, 4, 2, true, false, true, 0, 0); break;
#endif
            // two fp16 pieces: a K step is 2 KiB per tile, so the rings are one step deeper in the same LDS
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 113: GVQA_SN_LAUNCH(2, 4, 4, 2, 4, true, false, true, 0, 0, 2); break;
#endif
            case 114: GVQA_SN_LAUNCH(2, 4, 4, 2, 4, true, false, false, 0, 1, 2); break;
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 115: GVQA_SN_LAUNCH(2, 4, 4, 2, 3, true, false, true, 0, 0, 2); break;
#endif
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 117: GVQA_SP_LAUNCH(2, 4, 4, 2, 4, true, false, true, 0, 0, 2, 1); break;
#endif
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 119: GVQA_SP_LAUNCH(2, 2, 4, 4, 4, true, false, true, 0, 0, 2, 0); break;     // four waves of 128 x 128 (256 AGPRs)
#endif
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 111: if (KB & 1) return GVQA_E_UNSUPPORTED; GVQA_SK_LAUNCH(2, 4, 4, 2, 2, true, false, true, 0, 0, 2, 0, 2); break;   // two K steps per barrier
#endif
            case 112: if (KB & 1) return GVQA_E_UNSUPPORTED; GVQA_SK_LAUNCH(2, 4, 4, 2, 2, true, false, false, 0, 1, 2, 0, 2); break;
            case 118: GVQA_SP_LAUNCH(2, 4, 4, 2, 4, true, false, false, 0, 1, 2, 1); break;
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 116: GVQA_SN_LAUNCH(2, 4, 4, 2, 4, false, false, true, 0, 0, 2); break;
#endif
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 123: GVQA_SN_LAUNCH(2, 2, 2, 2, 3, true, false, true, 0, 0, 2); break;
#endif
#ifdef GVQA_PROBES      /* main-loop-only measurement variant: leaves C unwritten */
            case 133: GVQA_SN_LAUNCH(2, 2, 2, 4, 3, true, false, true, 0, 0, 2); break;
#endif
            case 134: GVQA_SN_LAUNCH(2, 2, 2, 4, 3, true, false, false, 0, 1, 2); break;
            case 124: GVQA_SN_LAUNCH(2, 2, 2, 2, 3, true, false, false, 0, 1, 2); break;       // 128 x 128 tile
            case 140: GVQA_SN_LAUNCH(2, 4, 2, 4, 3, true, false, false, 0, 3, 2); break;       // 128 x 512 tile, result as a packed operand
            default: return GVQA_E_INVALID;
        }
#undef GVQA_S3_LAUNCH
#undef GVQA_SN_LAUNCH
#undef GVQA_SP_LAUNCH
#undef GVQA_SK_LAUNCH
    }
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// ---- operands of the fused hop ----------------------------------------------------------------------------------
// A: node rows gathered into row groups -- slot s = 128 r + i holds node group_ptr[r] + i (zeros past the group's end).
// grid (ceil(KB / 4), 4 G): blockIdx.y = row tile (4 per group)
__global__ __launch_bounds__(256) void k_split3_pack_groups(const int32_t* __restrict__ group_ptr, int K, int KB,
                                                            const float* __restrict__ X, int64_t ld, uint16_t* __restrict__ out, int vec) {
    const int lane = threadIdx.x & 63, kb = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (kb >= KB) return;
    const int64_t rt = blockIdx.y;
    const int grp = (int)(rt >> 2), i = (int)(rt & 3) * 32 + (lane & 31);
    const int ns = group_ptr[grp], cnt = group_ptr[grp + 1] - ns;
    const int k0 = kb * 16 + (lane >> 5) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (i < cnt) {
        const float* src = X + (int64_t)(ns + i) * ld + k0;
        if (vec && k0 + 8 <= K) {
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (k0 + e < K) v[e] = src[e];
        }
    }
    split3_store(v, out + ((rt * KB + kb) * 3) * 512 + lane * 8);
}

// The same pack with the attention logits of the node rows computed on the way (the rows are read from HBM once instead
// of twice): a_node[node, j] = sum_k h[node, k] Vn[j, k], j < J = 2 H (the folded a_l | a_r vectors, gat_skip.py:134-135).
// One block per 32-slot row tile; wave w walks k blocks w, w + 4, ...; Vn lives in LDS ([J][Kp] floats, zero padded);
// partial dots are combined across the two k halves of a wave (lane ^ 32) and the four waves in a fixed order.
template <int J>
__global__ __launch_bounds__(256) void k_split3_pack_groups_logits(const int32_t* __restrict__ group_ptr, int K, int KB,
                                                                   const float* __restrict__ X, int64_t ld,
                                                                   uint16_t* __restrict__ out, int vec,
                                                                   const float* __restrict__ Vn, float* __restrict__ a_node) {
    extern __shared__ __attribute__((aligned(16))) float vn_s[];      // [J][Kp] + [4][32][J] partials
    const int Kp = KB * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < J * Kp; idx += 256) {
        const int j = idx / Kp, k = idx - j * Kp;
        vn_s[idx] = k < K ? Vn[(int64_t)j * K + k] : 0.f;
    }
    __syncthreads();
    const int64_t rt = blockIdx.x;
    const int grp = (int)(rt >> 2), i = (int)(rt & 3) * 32 + (lane & 31);
    const int ns = group_ptr[grp], cnt = group_ptr[grp + 1] - ns;
    const bool row_on = i < cnt;
    const float* row = X + (int64_t)(ns + (row_on ? i : 0)) * ld;
    float acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = 0.f;
    for (int kb = wave; kb < KB; kb += 4) {
        const int k0 = kb * 16 + (lane >> 5) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (row_on) {
            if (vec && k0 + 8 <= K) {
                const float4 a = *reinterpret_cast<const float4*>(row + k0), b = *reinterpret_cast<const float4*>(row + k0 + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k0 + e < K) v[e] = row[k0 + e];
            }
        }
        split3_store(v, out + ((rt * KB + kb) * 3) * 512 + lane * 8);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float4 w0 = *reinterpret_cast<const float4*>(vn_s + j * Kp + k0), w1 = *reinterpret_cast<const float4*>(vn_s + j * Kp + k0 + 4);
            acc[j] += v[0] * w0.x + v[1] * w0.y + v[2] * w0.z + v[3] * w0.w + v[4] * w1.x + v[5] * w1.y + v[6] * w1.z + v[7] * w1.w;
        }
    }
    float* part = vn_s + J * Kp;                                    // [4 waves][32 rows][J]
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const float t = acc[j] + __shfl_xor(acc[j], 32, 64);        // the two k halves of the row
        if (lane < 32) part[(wave * 32 + lane) * J + j] = t;
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * J; idx += 256) {
        const int r = idx / J, j = idx - r * J;
        const int ii = (int)(rt & 3) * 32 + r;
        if (ii < cnt)
            a_node[(int64_t)(ns + ii) * J + j] = (part[(0 * 32 + r) * J + j] + part[(1 * 32 + r) * J + j]) +
                                                 (part[(2 * 32 + r) * J + j] + part[(3 * 32 + r) * J + j]);
    }
}

// B: weight rows head-interleaved -- packed row 256 cb + h cw + cc holds W[h C + cb cw + cc, :] (zeros for channels >= C),
// so that column block cb of the product carries channels [cb cw, cb cw + cw) of every head.  grid (ceil(KB / 4), 8 ncb)
__global__ __launch_bounds__(256) void k_split3_pack_heads(int H, int C, int cw, int K, int KB, const float* __restrict__ W,
                                                           int64_t ldw, uint16_t* __restrict__ out, int vec) {
    const int lane = threadIdx.x & 63, kb = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (kb >= KB) return;
    const int64_t rt = blockIdx.y;
    const int r = (int)(rt * 32) + (lane & 31);
    const int cb = r >> 8, within = r & 255, h = within / cw, ch = cb * cw + (within - h * cw);
    const int k0 = kb * 16 + (lane >> 5) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (ch < C) {
        const float* src = W + (int64_t)(h * C + ch) * ldw + k0;
        if (vec && k0 + 8 <= K) {
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (k0 + e < K) v[e] = src[e];
        }
    }
    split3_store(v, out + ((rt * KB + kb) * 3) * 512 + lane * 8);
}

bool split_pack_groups_logits_supported(int np, int J, int64_t K) {
    const size_t lds = np == 2 ? split2h_pack_lds(J, (int)cdiv(K, 16)) : ((size_t)J * cdiv(K, 16) * 16 + 4 * 32 * (size_t)J) * sizeof(float);   // (fp32-Vn form)
    return (J == 2 || J == 4 || J == 8 || J == 16) && lds <= 64 * 1024;
}

int launch_split_pack_groups(int np, int num_groups, const int32_t* group_ptr, int64_t K, const float* X, int64_t ld, void* packed,
                             const float* Vn, int J, float* a_node, hipStream_t stream, const void* Vn_packed) {
    GVQA_REQUIRE(np == 2 || np == 3, GVQA_E_INVALID, "split_pack_groups: 2 or 3 pieces");
    GVQA_REQUIRE(num_groups >= 0 && K > 0 && K < (1ll << 30) && ld >= K, GVQA_E_INVALID, "split3_pack_groups: bad size");
    if (num_groups == 0) return GVQA_OK;
    GVQA_REQUIRE(group_ptr && X && packed, GVQA_E_INVALID, "split3_pack_groups: null operand");
    const int KB = (int)cdiv(K, 16);
    const int vec = (reinterpret_cast<uintptr_t>(X) & 15) == 0 && ld % 4 == 0;
    const int64_t RT = (int64_t)num_groups * 4;
    if (np == 2) {
        const bool with_logits = Vn && a_node;
        GVQA_REQUIRE(!a_node || (with_logits && split_pack_groups_logits_supported(2, J, K)), GVQA_E_UNSUPPORTED,
                     "split_pack_groups: logits on the way need 2 H in {2, 4, 8, 16} and [2 H, K] within 64 KiB of LDS");
        PackRows pr{X, ld, 0, group_ptr, 0, 0, 0};
        return launch_split2h_pack_tiles<PACK_GROUPS>(pr, RT, K, packed, with_logits ? Vn : nullptr, with_logits ? J : 0,
                                                      with_logits ? a_node : nullptr, stream, with_logits ? Vn_packed : nullptr);
    }
    const size_t lds = ((size_t)J * KB * 16 + 4 * 32 * (size_t)J) * sizeof(float);
    if (Vn && a_node && (J == 2 || J == 4 || J == 8 || J == 16) && lds <= 64 * 1024) {     // logits on the way (rows read once)
        uint16_t* o = static_cast<uint16_t*>(packed);
#define GVQA_PGL(J_) hipLaunchKernelGGL((k_split3_pack_groups_logits<J_>), dim3((unsigned)RT), dim3(256), lds, stream, group_ptr, (int)K, KB, \
                                        X, ld, o, vec, Vn, a_node)
        switch (J) {
            case 2: GVQA_PGL(2); break;
            case 4: GVQA_PGL(4); break;
            case 8: GVQA_PGL(8); break;
            default: GVQA_PGL(16); break;
        }
#undef GVQA_PGL
        GVQA_LAUNCH_CHECK();
        return GVQA_OK;
    }
    GVQA_REQUIRE(!a_node, GVQA_E_UNSUPPORTED, "split3_pack_groups: logits on the way need 2 H in {2, 4, 8, 16} and [2 H, K] within 64 KiB of LDS");
    for (int64_t r0 = 0; r0 < RT; r0 += 65532) {       // grid.y limit, whole groups per launch
        const int64_t n = std::min<int64_t>(65532, RT - r0);
        hipLaunchKernelGGL(k_split3_pack_groups, dim3((unsigned)cdiv(KB, 4), (unsigned)n), dim3(256), 0, stream, group_ptr + r0 / 4,
                           (int)K, KB, X, ld, static_cast<uint16_t*>(packed) + r0 * KB * 1536, vec);
    }
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

int launch_split_pack_heads(int np, int H, int C, int cw, int64_t K, const float* W, int64_t ldw, void* packed, hipStream_t stream) {
    GVQA_REQUIRE(np == 2 || np == 3, GVQA_E_INVALID, "split_pack_heads: 2 or 3 pieces");
    GVQA_REQUIRE(H > 0 && C > 0 && cw > 0 && H * cw == 256 && K > 0 && ldw >= K, GVQA_E_INVALID, "split3_pack_heads: bad size");
    GVQA_REQUIRE(W && packed, GVQA_E_INVALID, "split3_pack_heads: null operand");
    const int KB = (int)cdiv(K, 16), ncb = (int)cdiv(C, cw);
    const int vec = (reinterpret_cast<uintptr_t>(W) & 15) == 0 && ldw % 4 == 0;
    if (np == 2) {
        PackRows pr{W, ldw, 0, nullptr, H, C, cw};
        return launch_split2h_pack_tiles<PACK_HEADS>(pr, (int64_t)ncb * 8, K, packed, nullptr, 0, nullptr, stream);
    }
    hipLaunchKernelGGL(k_split3_pack_heads, dim3((unsigned)cdiv(KB, 4), (unsigned)(ncb * 8)), dim3(256), 0, stream, H, C, cw, (int)K, KB, W,
                       ldw, static_cast<uint16_t*>(packed), vec);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// the half-interleaved weight layout of the two-workgroups-per-CU hop kernel (hop2.hip); two fp16 pieces only
int launch_split_pack_heads2(int H, int C, int cw, int64_t K, const float* W, int64_t ldw, void* packed, hipStream_t stream) {
    GVQA_REQUIRE(H > 0 && C > 0 && cw >= 2 && H * cw == 256 && K > 0 && ldw >= K, GVQA_E_INVALID, "split_pack_heads2: bad size");
    GVQA_REQUIRE(W && packed, GVQA_E_INVALID, "split_pack_heads2: null operand");
    PackRows pr{W, ldw, 0, nullptr, H, C, cw};
    return launch_split2h_pack_tiles<PACK_HEADS2>(pr, (int64_t)cdiv(C, cw) * 8, K, packed, nullptr, 0, nullptr, stream);
}

// edges of one row group the fused epilogue can hold in LDS beside the 128 KiB row image
size_t hop_fused_lds_edge_capacity(int H) { return (size_t)(8192 - 192 - 128) / (size_t)(H + 1); }      // 32 KiB of words, padded sub-arrays
size_t hop_fused_chain_lds_edge_capacity(int H) { return (size_t)(8192 - 768 - 192 - 128) / (size_t)(H + 1); }      // (3 KiB of the 32 hold the chain's per-group arrays)
// in-kernel coefficients (CHN = 2): BOTH groups' regions [rowptr | src | raw logits -> alpha | eid] resident at once, 13 KiB each (sub-arrays padded to 64 words)
size_t hop_fused_ic_lds_edge_capacity(int H) {
    size_t e = (size_t)(13 * 256 - 192) / (size_t)(H + 2);
    while (e > 0 && 192 + 2 * ((e + 63) & ~(size_t)63) + ((e * H + 63) & ~(size_t)63) > 13 * 256) --e;
    return e;
}

int launch_hop_fused_split(int np, int64_t K, const void* Apk, const void* Bpk, const FusedHopArgs& f, hipStream_t stream, const Hop2ChainDesc* cd) {
    GVQA_REQUIRE(np == 2 || np == 3, GVQA_E_INVALID, "hop_fused: 2 or 3 pieces");
    GVQA_REQUIRE(Apk && Bpk && (f.out || (cd && cd->Pnext)) && f.group_ptr && f.rowptr && f.csr_src && (f.alpha_csr || f.ic_a_edge) && f.node_graph, GVQA_E_INVALID,
                 "hop_fused: null operand");
    GVQA_REQUIRE(f.H * f.cw == 256 && f.C % 4 == 0 && f.cw % 4 == 0 && (f.cw & (f.cw - 1)) == 0, GVQA_E_UNSUPPORTED,
                 "hop_fused: needs H in {1,2,4,8} and C %% 4 == 0");
    GVQA_REQUIRE((size_t)f.e_cap <= hop_fused_lds_edge_capacity(f.H), GVQA_E_UNSUPPORTED, "hop_fused: row group has too many edges for LDS");
    if (f.num_groups == 0) return GVQA_OK;
    const int KB = (int)cdiv(K, 16), ncb = (int)cdiv(f.C, f.cw);
    const int rtA = f.num_groups * 4, rtB = ncb * 8;
    GVQA_REQUIRE(cdiv(f.num_groups, 2) <= 65535, GVQA_E_UNSUPPORTED, "hop_fused: too many row groups for one launch");
    // Row blocks: pairs of row groups (256-row tiles), and -- when the pairs leave a last round of workgroups at most half full -- that
    // round's groups one per block (half tiles: twice the workgroups, each about 0.6 of a full tile's time).
    int pair_blocks = (int)cdiv(f.num_groups, 2), single_blocks = 0;
    if (get_option(GVQA_OPT_HOP_HALF_TILES) != 0 && f.num_groups >= 2) {
        const int64_t cus = device_cu_count(), P = (int64_t)pair_blocks * ncb;
        const int64_t rem_wg = P % cus;                                   // pair workgroups of the last partial round
        if (get_option(GVQA_OPT_HOP_HALF_TILES) == 2) {                   // (measurement: every block one row group)
            pair_blocks = 0;
            single_blocks = f.num_groups;
        } else if (rem_wg > 0 && 2 * rem_wg <= cus) {
            const int rem_blocks = (int)std::min<int64_t>(cdiv(rem_wg, ncb), pair_blocks);
            pair_blocks -= rem_blocks;
            single_blocks = f.num_groups - 2 * pair_blocks;
        }
        if ((int64_t)pair_blocks + single_blocks > 65535) {               // (grid.y: the split adds up to rem_blocks rows to cdiv(groups, 2) -- near the limit keep pairs only)
            pair_blocks = (int)cdiv(f.num_groups, 2);
            single_blocks = 0;
        }
    }
    dim3 grid((unsigned)ncb, (unsigned)(pair_blocks + single_blocks));
    FusedHopArgs f2 = f;
    f2.pair_blocks = single_blocks > 0 ? pair_blocks : (int)cdiv(f.num_groups, 2);
    f2.debug = 0;
    {
#ifdef GVQA_PROBES
        if (const char* dbg = getenv("GVQA_FUSED_DEBUG")) f2.debug = atoi(dbg);
        const char* xv = getenv("GVQA_FUSED_XCD");
        const int want = xv ? atoi(xv) : 4;
#else
        constexpr int want = 4;                               // measured at config 3: 458 (plain order) / 452 / 449 / 447 us for 1 / 2 / 4 / 8
#endif
        const int rows = single_blocks > 0 ? 1 : (int)cdiv(f.num_groups, 2);          // (half tiles change the block -> group map: plain launch order)
        f2.xcd_cols = (ncb == 8 && (want == 2 || want == 4 || want == 8) && rows % (want) == 0) ? want : 1;
    }
    const float* a_inv = np == 2 ? split2h_inv_scales(Apk, rtA, KB) : nullptr;
    const float* b_inv = np == 2 ? split2h_inv_scales(Bpk, rtB, KB) : nullptr;
    if (cd) {       // chained hop (hop2.hip's protocol on this kernel): skip rows out of the packed input, output as the next hop's operand
        GVQA_REQUIRE(np == 2 && f.H == 4 && K == f.C && cd->bc && cd->graph_ptr && (!cd->Pnext || cd->PMout) && (cd->Pnext || f.out), GVQA_E_INVALID,
                     "hop_fused: a chained hop needs two-piece operands, H = 4, node_dim == out_channels and its side arrays");
        GVQA_REQUIRE((size_t)f.e_cap <= hop_fused_chain_lds_edge_capacity(f.H), GVQA_E_UNSUPPORTED, "hop_fused: row group has too many edges for a chained hop");
        f2.ch_apk = static_cast<const uint16_t*>(Apk);
        f2.ch_a_inv = a_inv;
        f2.ch_pnext = static_cast<uint16_t*>(cd->Pnext);
        f2.ch_a_inv_next = cd->Pnext ? reinterpret_cast<float*>(static_cast<char*>(cd->Pnext) + (size_t)f.num_groups * 4 * KB * 2048) : nullptr;
        f2.ch_gscale = cd->gscale; f2.ch_pmout = cd->PMout; f2.ch_tmax = cd->Tmax; f2.ch_bc = cd->bc; f2.ch_graph_ptr = cd->graph_ptr;
        f2.ch_B = cd->B; f2.ch_KB = KB;
#define GVQA_FUSED_CHAIN(NBUF_, KS_, CHN_)                                                                                      \
        hipLaunchKernelGGL((k_linear_split3<2, 4, 4, 2, NBUF_, true, false, false, 0, 2, 4, 2, 0, KS_, CHN_>), grid, dim3(512), 0, stream, \
                           f.num_groups * 128, ncb * 256, KB, static_cast<const uint16_t*>(Apk), rtA, static_cast<const uint16_t*>(Bpk), \
                           rtB, LinearEpilogue{}, nullptr, (int64_t)0, 0, f2, a_inv, b_inv)
        if (f.ic_a_edge) {      // coefficients inside the kernel (CHN = 2)
            GVQA_REQUIRE(f.ic_csr_eid && f.ic_lp_in && f.ic_parts_in >= 1 && (!f.ic_lp_out || f.ic_vn_next) && (size_t)f.e_cap <= hop_fused_ic_lds_edge_capacity(f.H) &&
                         (f.ic_a_edge_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(f.ic_a_edge) & 15) == 0 && f.C >= 4, GVQA_E_INVALID,
                         "hop_fused: in-kernel coefficients need the slots' edge ids, 16-byte aligned edge halves, partial node logits and a row group within %zu edges",
                         hop_fused_ic_lds_edge_capacity(f.H));
            if ((KB & 1) == 0) GVQA_FUSED_CHAIN(2, 2, 2);
            else GVQA_FUSED_CHAIN(4, 1, 2);
        } else if ((KB & 1) == 0) GVQA_FUSED_CHAIN(2, 2, 1);
        else GVQA_FUSED_CHAIN(4, 1, 1);
#undef GVQA_FUSED_CHAIN
        GVQA_LAUNCH_CHECK();
        return GVQA_OK;
    }
    // (the read-ahead main loop, PIPE, gains 3 % in the plain GEMM and nothing here: 317 us either way for the main loop alone)
#define GVQA_FUSED_LAUNCH(H_, NBUF_, NP_) GVQA_FUSED_LAUNCH_K(H_, NBUF_, NP_, 1)
#define GVQA_FUSED_LAUNCH_K(H_, NBUF_, NP_, KS_)                                                                                \
    hipLaunchKernelGGL((k_linear_split3<2, 4, 4, 2, NBUF_, true, false, false, 0, 2, H_, NP_, 0, KS_>), grid, dim3(512), 0, stream, \
                       f.num_groups * 128, ncb * 256, KB, static_cast<const uint16_t*>(Apk), rtA, static_cast<const uint16_t*>(Bpk), \
                       rtB, LinearEpilogue{}, nullptr, (int64_t)0, 0, f2, a_inv, b_inv)
    if (np == 3) {
        switch (f.H) {
            case 1: GVQA_FUSED_LAUNCH(1, 3, 3); break;
            case 2: GVQA_FUSED_LAUNCH(2, 3, 3); break;
            case 4: GVQA_FUSED_LAUNCH(4, 3, 3); break;
            default: GVQA_FUSED_LAUNCH(8, 3, 3); break;
        }
    } else if ((KB & 1) == 0
#ifdef GVQA_PROBES
               && !getenv("GVQA_FUSED_KS1")
#endif
    ) {
        // two fp16 pieces, even number of k blocks: two K steps per 64 KiB stage and per barrier, two stages (59.6 k vs 65.3 k
        // cycles per tile in the main loop, 285 vs 298 us at config 3: part of the saving comes back as a lower clock)
        switch (f.H) {
            case 1: GVQA_FUSED_LAUNCH_K(1, 2, 2, 2); break;
            case 2: GVQA_FUSED_LAUNCH_K(2, 2, 2, 2); break;
            case 4: GVQA_FUSED_LAUNCH_K(4, 2, 2, 2); break;
            default: GVQA_FUSED_LAUNCH_K(8, 2, 2, 2); break;
        }
    } else {        // two fp16 pieces: 32 KiB per K step, four steps in the 128 KiB below the epilogue's CSR regions
        switch (f.H) {
            case 1: GVQA_FUSED_LAUNCH(1, 4, 2); break;
            case 2: GVQA_FUSED_LAUNCH(2, 4, 2); break;
            case 4: GVQA_FUSED_LAUNCH(4, 4, 2); break;
            default: GVQA_FUSED_LAUNCH(8, 4, 2); break;
        }
    }
#undef GVQA_FUSED_LAUNCH
#undef GVQA_FUSED_LAUNCH_K
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // namespace gvqa

namespace gvqa {
// A plain fp32 Linear, C = A W^T (+ epilogue), for callers that own a scratch buffer: on the two-piece kernels (A and W packed
// into the scratch per call) when the product is large enough and GVQA_OPT_PROJECTION allows it, on the f32-input MFMA kernels
// otherwise.  Used by the pooling head's node MLPs ([N, 512] x [512, 512]: 264 -> 150 us per layer at config 3).
size_t linear_auto_scratch_bytes(int64_t M, int64_t N, int64_t K) {
    return align_up(split_packed_bytes(2, M, K), 256) + align_up(split_packed_bytes(2, N, K), 256);
}
int launch_linear_auto(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw, LinearEpilogue ep,
                       float* C, int64_t ldc, void* scratch, size_t scratch_bytes, hipStream_t stream) {
    const bool split = get_option(GVQA_OPT_PROJECTION) != GVQA_PROJECTION_F32 && scratch && M > 0 &&
                       scratch_bytes >= linear_auto_scratch_bytes(M, N, K) && linear_split3_supported(N, ep, C, ldc) &&
                       (reinterpret_cast<uintptr_t>(scratch) & 255) == 0 &&
                       2.0 * (double)M * (double)N * (double)K >= 1e6 * (double)get_option(GVQA_OPT_SPLIT3_MIN_MFLOP);
    if (!split) return launch_linear_t(M, N, K, A, lda, W, ldw, ep, C, ldc, 1, 0, 0, 0, 0, stream);
    char* apk = static_cast<char*>(scratch);
    char* wpk = apk + align_up(split_packed_bytes(2, M, K), 256);
    int rc = launch_split_pack(2, M, K, A, lda, apk, stream);
    if (rc) return rc;
    rc = launch_split_pack(2, N, K, W, ldw, wpk, stream);
    if (rc) return rc;
    return launch_linear_split(2, M, N, K, apk, wpk, ep, C, ldc, stream);
}
}  // namespace gvqa

extern "C" size_t gvqa_split3_packed_bytes(int64_t rows, int64_t K) {
    if (rows <= 0 || K <= 0) return 0;
    return gvqa::split_packed_bytes(3, rows, K);
}

extern "C" int gvqa_split3_pack(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, void* stream) {
    return gvqa::launch_split_pack(3, rows, K, X, ld, packed, static_cast<hipStream_t>(stream));
}

extern "C" size_t gvqa_split2h_packed_bytes(int64_t rows, int64_t K) {
    if (rows <= 0 || K <= 0) return 0;
    return gvqa::split_packed_bytes(2, rows, K);
}

extern "C" int gvqa_split2h_pack_logits(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, float* absmax, const float* Vn, int32_t J,
                                        float* a_node, void* stream) {
    GVQA_REQUIRE(Vn && a_node && J == 8 && K % 4 == 0 && K <= 1024, GVQA_E_UNSUPPORTED, "split2h_pack_logits: J = 8 vectors, K %% 4 == 0, K <= 1024");
    if (absmax) GVQA_HIP_CHECK(hipMemsetAsync(absmax, 0, GVQA_ABSMAX_SLOTS * sizeof(float), static_cast<hipStream_t>(stream)));
    return gvqa::launch_split_pack(2, rows, K, X, ld, packed, static_cast<hipStream_t>(stream), absmax, Vn, J, a_node);
}
extern "C" int gvqa_split2h_pack_absmax(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, float* absmax, void* stream) {
    GVQA_REQUIRE(absmax, GVQA_E_INVALID, "split2h_pack_absmax: null maxima");
    GVQA_HIP_CHECK(hipMemsetAsync(absmax, 0, GVQA_ABSMAX_SLOTS * sizeof(float), static_cast<hipStream_t>(stream)));
    return gvqa::launch_split_pack(2, rows, K, X, ld, packed, static_cast<hipStream_t>(stream), absmax);
}
extern "C" int gvqa_split2h_pack(int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, void* stream) {
    return gvqa::launch_split_pack(2, rows, K, X, ld, packed, static_cast<hipStream_t>(stream));
}

extern "C" int gvqa_linear_split2h(int64_t M, int64_t N, int64_t K, const void* Apk, const void* Bpk, const float* bias,
                                   const float* addend, int64_t ld_add, const float* mul, int64_t ld_mul, int relu, float* C,
                                   int64_t ldc, void* stream) {
    gvqa::LinearEpilogue ep{bias, addend, ld_add, mul, ld_mul, relu};
    return gvqa::launch_linear_split(2, M, N, K, Apk, Bpk, ep, C, ldc, static_cast<hipStream_t>(stream));
}

extern "C" int gvqa_linear_split2h_chain(int64_t M, int64_t N, int64_t K1, const void* Apk, int64_t K2, const void* A2pk, const void* Bpk,
                                         const float* bias, const float* addend, int64_t ld_add, const float* mul, int64_t ld_mul, int relu,
                                         float* C, int64_t ldc, void* pk_out, void* stream) {
    using namespace gvqa;
    GVQA_REQUIRE(M >= 0 && N > 0 && K1 > 0 && K2 >= 0 && (A2pk != nullptr) == (K2 > 0) && (!A2pk || K1 % 16 == 0), GVQA_E_INVALID,
                 "linear_split2h_chain: bad segment sizes (K1 must be a multiple of 16 when a second segment follows)");
    LinearEpilogue ep{bias, addend, ld_add, mul, ld_mul, relu};
    if (A2pk) {
        ep.a2 = static_cast<const uint16_t*>(A2pk);
        ep.a2_kb0 = (int)(K1 / 16);
        ep.a2_KB = (int)cdiv(K2, 16);
        ep.a2_inv = reinterpret_cast<const float*>(static_cast<const char*>(A2pk) + (size_t)cdiv(M, 32) * ep.a2_KB * 2048);
    }
    ep.pk_out = static_cast<uint16_t*>(pk_out);
    return launch_linear_split(2, M, N, K1 + (A2pk ? (int64_t)ep.a2_KB * 16 : 0), Apk, Bpk, ep, C, ldc, static_cast<hipStream_t>(stream));
}

extern "C" int gvqa_linear_split3(int64_t M, int64_t N, int64_t K, const void* Apk, const void* Bpk, const float* bias,
                                  const float* addend, int64_t ld_add, const float* mul, int64_t ld_mul, int relu, float* C,
                                  int64_t ldc, void* stream) {
    gvqa::LinearEpilogue ep{bias, addend, ld_add, mul, ld_mul, relu};
    return gvqa::launch_linear_split(3, M, N, K, Apk, Bpk, ep, C, ldc, static_cast<hipStream_t>(stream));
}
