// C = X^T Y straight from the row-major fp32 operands (gfx950): the weight gradient dW = dy^T x of a projection (the reference gets it from
// autograd through torch.nn.Linear, /root/reference gat_skip.py:133), a reduction over ALL rows of the batch.
//
// train.hip's first form packs both operands TRANSPOSED (k_split2h_pack_t: 1.6 GB of traffic per hop for dy alone) so that the split GEMM of
// split3.hip can contract over a contiguous dimension.  Here the transposition happens on the way from global memory to the MFMA fragment
// image in LDS and nothing is packed in HBM:
//
//   * workgroup = a 256 x 256 tile of C over one chunk of KC rows (split-K: partial results added in a fixed order by k_splitk_reduce);
//     8 waves as 2 (128 columns of X) x 4 (64 columns of Y), 4 x 2 accumulators of 32 x 32 per wave;
//   * K step = 16 rows.  Thread (column pair cp, row half g) of an operand's 256 loaders reads X[r0 + 8 g + j, 2 cp .. 2 cp + 1], j = 0..7 -- a
//     wave's load instruction is 128 consecutive floats of ONE row, 512 contiguous bytes; waves 0..3 serve X, 4..7 serve Y.  The eight values
//     of a column ARE the 8 consecutive k of lane (c mod 32, g) of an MFMA operand fragment: scaled by the operand's power of two, split into two fp16 pieces (v_fma_mix, as the
//     aggregate-first hop kernel does), written as 2 x 16 bytes into the fragment image of the NEXT step (double-buffered, 64 KiB);
//   * the loads of step s + 2 are issued when step s + 1's registers have been converted: a full step (~1.4 us) to land;
//   * one power-of-two scale per operand (train.hip's argument: a piece pair carries 22 bits below each element's own exponent while
//     the element is within 2^18 of the operand's largest), from the producer's slice maxima or k_absmax.
//
// Same arithmetic as the packed form: same scales, same pieces, the same three piece products (in another order), fp32 accumulation, the same
// split-K chunks.
#include "common.h"
#include "gemm_tile.h"

namespace gvqa {

typedef _Float16 tnd_f16x8 __attribute__((ext_vector_type(8)));

#if defined(__HIP_DEVICE_COMPILE__)
#define GVQA_TND_SPLIT2(hi_, lo_, p_, a_, b_)                                                                                         \
    asm("v_fma_mixlo_f16 %0, %2, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %4, 0\n\t"                                                          \
        "v_fma_mixlo_f16 %1, %2, %3, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %2, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"       \
        : "=&v"(hi_), "=&v"(lo_) : "v"(p_), "v"(a_), "v"(b_))
#else
#define GVQA_TND_SPLIT2(hi_, lo_, p_, a_, b_) do { (void)(p_); (void)(a_); (void)(b_); (hi_) = 0u; (lo_) = 0u; } while (0)     /* (host pass) */
#endif

constexpr int TND_TILE = 256, TND_STEP = 16, TND_IMG = 16 * 1024;      // one operand's fragment image of a step: 8 tiles x 2 pieces x 1 KiB

struct TndArgs {
    int64_t R;
    int M, N;
    const float* X; int64_t ldx;
    const float* Y; int64_t ldy;
    const float* xmax; int nxmax;
    const float* ymax; int nymax;
    int KC, S, tiles_n, ntiles;
    float* C; int64_t ldc, zs_c;
};

__global__ __launch_bounds__(512) void k_linear_tn_direct(TndArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TND_IMG];
    __shared__ float mx_s[2][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // chunk / tile of this workgroup.  The tiles of one chunk read the same rows: consecutive workgroup ids go round the 8 XCDs, so a chunk's
    // tiles take ids that are 8 apart (one XCD, one L2) when the chunk count allows it
    int z, tile;
    {
        const int L = blockIdx.x;
        if (a.S % 8 == 0) { const int q = L >> 3; z = (L & 7) + 8 * (q / a.ntiles); tile = q % a.ntiles; }
        else { z = L / a.ntiles; tile = L % a.ntiles; }
    }
    const int m0 = (tile / a.tiles_n) * TND_TILE, n0 = (tile % a.tiles_n) * TND_TILE;
    const int64_t rbeg = (int64_t)z * a.KC, rend = rbeg + a.KC < a.R ? rbeg + a.KC : a.R;
    const int rows = (int)(rend - rbeg);
    const int ns = (rows + TND_STEP - 1) / TND_STEP;

    // the operands' scales
    {
        float mx = tid < a.nxmax ? a.xmax[tid] : 0.f, my = tid < a.nymax ? a.ymax[tid] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); my = fmaxf(my, __shfl_xor(my, o, 64)); }
        if (lane == 0) { mx_s[0][wave] = mx; mx_s[1][wave] = my; }
    }
    __syncthreads();
    float vx = 0.f, vy = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { vx = fmaxf(vx, mx_s[0][w]); vy = fmaxf(vy, mx_s[1][w]); }
    const int ex = split2h_exponent(vx), ey = split2h_exponent(vy);

    // loader role: waves 0..3 load X, waves 4..7 load Y; thread = (column PAIR cp, row half g) of its operand's tile: 8 rows x 2 adjacent columns
    // per step as eight 8-byte buffer loads (a wave instruction = 512 contiguous bytes of one row; half the load instructions of one column
    // per thread), i.e. two fragment units.  Buffer loads: a wave-uniform descriptor (base = the step's first row, size = what is left of the
    // chunk: rows past its end read as zero, no guards anywhere) + eight 32-bit per-lane byte offsets (row j of the half, column 2 cp)
    const int op = __builtin_amdgcn_readfirstlane(tid >> 8);           // 0: X, 1: Y
    const int cp = tid & 127, g = (tid >> 7) & 1, c = 2 * cp;
    const int col0 = op ? n0 : m0, ncols = op ? a.N : a.M;
    const int64_t ld = op ? a.ldy : a.ldx;
    const bool okc = col0 + c < ncols;                                  // (M, N multiples of 4: a pair is inside or outside)
    const float sc = okc ? pow2i(op ? ey : ex) : 0.f;
    const float* bo = (op ? a.Y : a.X) + rbeg * ld;
    const int64_t bytes_o = ((int64_t)(rows - 1) * ld + ncols) * 4;
    const int64_t step_o = (int64_t)TND_STEP * ld * 4;
    unsigned oo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) oo[j] = 4u * (unsigned)((8 * g + j) * ld + (okc ? col0 + c : 0));
    auto rsrc_of = [](const float* base, int64_t first, int64_t total) {
        // (wave-uniform by construction -- kernel arguments, blockIdx, the wave's operand -- and made PROVABLY so for the compiler: a
        //  descriptor it takes for divergent turns every load into a waterfall loop)
        const int64_t left = total - first;
        const uint64_t addr = reinterpret_cast<uint64_t>(base) + (uint64_t)(left > 0 ? first : 0);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr), hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
        const int n = __builtin_amdgcn_readfirstlane((int)(left > 0 ? left : 0));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)hi << 32) | lo), (short)0, n, 0x00020000);
    };
#ifndef GVQA_TND_DBG        /* A/B build switch (scripts/ab_tn.sh): bit 1 no loads inside the loop, 2 no split / image writes, 4 no MFMAs */
#define GVQA_TND_DBG 0
#endif
    typedef float tnd_f32x2 __attribute__((ext_vector_type(2)));
#define GVQA_TND_LD(rs_, off_) __builtin_bit_cast(tnd_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_, (int)(off_), 0, 0))
#define GVQA_TND_LDL(dst_, rs_, off_) do { if (!(GVQA_TND_DBG & 1)) (dst_) = GVQA_TND_LD(rs_, off_); } while (0)
#define GVQA_TND_SPL(h_, l_, p_, a_, b_) do { if (!(GVQA_TND_DBG & 2)) GVQA_TND_SPLIT2(h_, l_, p_, a_, b_); } while (0)
    const unsigned wr_off = (unsigned)(op * TND_IMG + (c >> 5) * 2048 + ((c & 31) + 32 * g) * 16);      // unit of column c; column c + 1: + 16
    tnd_f32x2 fa[8], fb[8];                      // two register sets of loaded rows (row j: columns c, c + 1): one being converted, one in flight
    uint4 h0, l0, h1, l1;                        // hi / lo fragment units of columns c and c + 1

    // consumer role: wave (wr, wc) = rows [128 wr, +128) of the tile (4 fragments of X) x columns [64 wc, +64) (2 fragments of Y)
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned a_off = (unsigned)(wr * 4 * 2048 + lane * 16), b_off = (unsigned)(TND_IMG + wc * 2 * 2048 + lane * 16);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto rd = [&](const unsigned char* p) { return __builtin_bit_cast(tnd_f16x8, *reinterpret_cast<const uint4*>(p)); };
    tnd_f16x8 ah[4], al[4], bh[2], bl[2];
    // product n of a step: piece pair n / 8 -- (lo, hi), (hi, hi), (hi, lo) -- of accumulator n % 8: the three products of one accumulator are 8 apart
#define GVQA_TND_MF(n_) do { constexpr int q_ = (n_) / 8, t_ = (n_) % 8, i_ = t_ >> 1, j_ = t_ & 1;                                       \
        if (!(GVQA_TND_DBG & 4)) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q_ == 0 ? al[i_] : ah[i_], q_ == 2 ? bl[j_] : bh[j_], acc[i_][j_], 0, 0, 0); } while (0)
#define GVQA_TND_FENCE() __builtin_amdgcn_sched_barrier(0)

    {   // step 0's rows -> image 0; step 1's rows in flight in set A
        const auto r0 = rsrc_of(bo, 0, bytes_o);
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[j] = GVQA_TND_LD(r0, oo[j]);
        const auto r1 = rsrc_of(bo, step_o, bytes_o);
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[j] = GVQA_TND_LD(r1, oo[j]);
        GVQA_TND_SPLIT2(h0.x, l0.x, sc, fb[0].x, fb[1].x); GVQA_TND_SPLIT2(h0.y, l0.y, sc, fb[2].x, fb[3].x);
        GVQA_TND_SPLIT2(h0.z, l0.z, sc, fb[4].x, fb[5].x); GVQA_TND_SPLIT2(h0.w, l0.w, sc, fb[6].x, fb[7].x);
        GVQA_TND_SPLIT2(h1.x, l1.x, sc, fb[0].y, fb[1].y); GVQA_TND_SPLIT2(h1.y, l1.y, sc, fb[2].y, fb[3].y);
        GVQA_TND_SPLIT2(h1.z, l1.z, sc, fb[4].y, fb[5].y); GVQA_TND_SPLIT2(h1.w, l1.w, sc, fb[6].y, fb[7].y);
        unsigned char* d = smem + wr_off;
        *reinterpret_cast<uint4*>(d) = h0; *reinterpret_cast<uint4*>(d + 1024) = l0;
        *reinterpret_cast<uint4*>(d + 16) = h1; *reinterpret_cast<uint4*>(d + 16 + 1024) = l1;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) al[i] = rd(smem + a_off + i * 2048 + 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) bh[j] = rd(smem + b_off + j * 2048);
    // One step, laid out by hand: an MFMA holds the SIMD's matrix pipe for 32 cycles and the wave issues in order, so the step's other work --
    // 8 loads (rows of step s + 2 into set LD_), 32 VALU of the split (rows of step s + 1 out of set CV_), 4 LDS writes -- sits in the shadow of
    // the first ten MFMAs, a few instructions behind each; the first product's fragments (a lo, b hi) were read under the PREVIOUS step's last MFMAs.
    // The step's barrier comes right after the image writes (MFMAs queued in front of it
    // and behind it), not at the end of the step: what it orders is the image of step s + 1 (written above it by everybody, read at the top of the
    // next step) and the image of step s (read at the top of this step by everybody, overwritten below the NEXT barrier).  Loads have a whole
    // step to land.  Steps past the chunk's end: the descriptor is empty, the rows read as zero, the products add zeros.
#define GVQA_TND_STEP(s_, CV_, LD_)                                                                                                           \
    {                                                                                                                                         \
        const int st_ = (s_);                                                                                                                 \
        const unsigned char* img = smem + (st_ & 1) * (2 * TND_IMG);                                                                          \
        unsigned char* d = smem + ((st_ + 1) & 1) * (2 * TND_IMG) + wr_off;                                                                   \
        const auto rs_ = rsrc_of(bo, (int64_t)(st_ + 2) * step_o, bytes_o);                                                                   \
        const unsigned char* imn = smem + ((st_ + 1) & 1) * (2 * TND_IMG);                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) ah[i] = rd(img + a_off + i * 2048);                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) bl[j] = rd(img + b_off + j * 2048 + 1024);                                              \
        GVQA_TND_FENCE();                                                                                                                     \
        GVQA_TND_MF(0); GVQA_TND_FENCE(); GVQA_TND_SPL(h0.x, l0.x, sc, CV_[0].x, CV_[1].x); GVQA_TND_FENCE();                                  \
        GVQA_TND_MF(1); GVQA_TND_FENCE(); GVQA_TND_SPL(h1.x, l1.x, sc, CV_[0].y, CV_[1].y); GVQA_TND_LDL(LD_[0], rs_, oo[0]); GVQA_TND_LDL(LD_[1], rs_, oo[1]); GVQA_TND_FENCE(); \
        GVQA_TND_MF(2); GVQA_TND_FENCE(); GVQA_TND_SPL(h0.y, l0.y, sc, CV_[2].x, CV_[3].x); GVQA_TND_FENCE();                                  \
        GVQA_TND_MF(3); GVQA_TND_FENCE(); GVQA_TND_SPL(h1.y, l1.y, sc, CV_[2].y, CV_[3].y); GVQA_TND_LDL(LD_[2], rs_, oo[2]); GVQA_TND_LDL(LD_[3], rs_, oo[3]); GVQA_TND_FENCE(); \
        GVQA_TND_MF(4); GVQA_TND_FENCE(); GVQA_TND_SPL(h0.z, l0.z, sc, CV_[4].x, CV_[5].x); GVQA_TND_FENCE();                                  \
        GVQA_TND_MF(5); GVQA_TND_FENCE(); GVQA_TND_SPL(h1.z, l1.z, sc, CV_[4].y, CV_[5].y); GVQA_TND_LDL(LD_[4], rs_, oo[4]); GVQA_TND_LDL(LD_[5], rs_, oo[5]); GVQA_TND_FENCE(); \
        GVQA_TND_MF(6); GVQA_TND_FENCE(); GVQA_TND_SPL(h0.w, l0.w, sc, CV_[6].x, CV_[7].x); GVQA_TND_FENCE();                                  \
        GVQA_TND_MF(7); GVQA_TND_FENCE(); GVQA_TND_SPL(h1.w, l1.w, sc, CV_[6].y, CV_[7].y); GVQA_TND_LDL(LD_[6], rs_, oo[6]); GVQA_TND_LDL(LD_[7], rs_, oo[7]); GVQA_TND_FENCE(); \
        GVQA_TND_MF(8); GVQA_TND_FENCE();                                                                                                     \
        if (!(GVQA_TND_DBG & 2)) { *reinterpret_cast<uint4*>(d) = h0; *reinterpret_cast<uint4*>(d + 1024) = l0; }                            \
        GVQA_TND_FENCE();                                                                                                                     \
        GVQA_TND_MF(9); GVQA_TND_FENCE();                                                                                                     \
        if (!(GVQA_TND_DBG & 2)) { *reinterpret_cast<uint4*>(d + 16) = h1; *reinterpret_cast<uint4*>(d + 16 + 1024) = l1; }                  \
        GVQA_TND_FENCE();                                                                                                                     \
        GVQA_TND_MF(10); GVQA_TND_MF(11); GVQA_TND_FENCE();                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                                    \
        __builtin_amdgcn_s_barrier();                                                                                                         \
        GVQA_TND_FENCE();                                                                                                                     \
        GVQA_TND_MF(12); GVQA_TND_MF(13); GVQA_TND_MF(14); GVQA_TND_MF(15);                                                                   \
        GVQA_TND_FENCE();                                                                                                                     \
        /* a lo and b hi are dead (products 0 .. 15): the NEXT step's come in under this step's last eight MFMAs, so that step starts on registers */ \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) al[i] = rd(imn + a_off + i * 2048 + 1024);                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) bh[j] = rd(imn + b_off + j * 2048);                                                     \
        GVQA_TND_FENCE();                                                                                                                     \
        GVQA_TND_MF(16); GVQA_TND_MF(17); GVQA_TND_MF(18); GVQA_TND_MF(19); GVQA_TND_MF(20); GVQA_TND_MF(21); GVQA_TND_MF(22); GVQA_TND_MF(23); \
        GVQA_TND_FENCE();                                                                                                                     \
    }
    for (int s = 0; s < ns; s += 2) {          // (an odd count runs one more step on zero rows)
        GVQA_TND_STEP(s, fa, fb)
        GVQA_TND_STEP(s + 1, fb, fa)
    }
#undef GVQA_TND_STEP
#undef GVQA_TND_MF
#undef GVQA_TND_FENCE
#undef GVQA_TND_LD
#undef GVQA_TND_LDL
#undef GVQA_TND_SPL

    // partial result (or C itself when there is one chunk): C/D map of the 32 x 32 MFMA -- column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const float inv = pow2i(-ex) * pow2i(-ey);
    float* dst = a.C + (int64_t)z * a.zs_c;
    const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gc = n0 + wc * 64 + j * 32 + ccol;
        if (gc >= a.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gr0 = m0 + wr * 128 + i * 32 + crow0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = gr0 + (r & 3) + 8 * (r >> 2);
                if (gr < a.M) dst[(int64_t)gr * a.ldc + gc] = acc[i][j][r] * inv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// C = A B^T with A taken as it is -- row-major fp32 rows, e.g. dy of dx = dy W (the input gradient of the projection) -- and B packed (the small,
// weight-sized operand: two-piece fragments [tile][k block][piece][lane][8] + per-row inverse scales, what k_split2h_pack_t / gvqa_split2h_pack
// write).  The row-form pack of dy (537 MB read, 537 MB written per hop at config 3, 263 us) goes away: thread (row, k half) of the 512 loads its
// 8 consecutive floats of the step (2 x 16 bytes), scales by the operand's ONE power of two (as the pack it replaces does for gradients:
// train.hip), splits, and writes the two 16-byte fragment units; B's 16 KiB of the step are copied global -> registers -> image.  Step layout,
// barrier placement and fragment rotation as in k_linear_tn_direct.  K % 16 == 0.
struct NndArgs {
    int64_t R;
    int N, K;
    const float* A; int64_t lda;
    const float* amax; int namax;
    const uint16_t* Bpk; int KBb, TB;         // k blocks per tile in the pack, tiles
    const float* b_inv;
    float* C; int64_t ldc;
    int accumulate, tiles_n, row_tiles;
    const float* lr_g; const float* lr_v; int J;      // optional rank-J term of the epilogue: C[r, n] += sum_j lr_g[r, j] lr_v[n, j]  (fp32 FMAs; J % 4 == 0, <= 16)
    const float* addend; int64_t ld_add;               // optional [R, N] added to the result
};

// LDS-DMA of a packed operand's fragment pair (hi at +0, lo at +1024 on both sides): global = scalar base + 32-bit lane offset, LDS = M0 base +
// lane * 16.  Inline asm: invisible to the compiler's wait-count bookkeeping, ordered by the counted s_waitcnt in front of the step's barrier.
typedef __attribute__((address_space(3))) unsigned char* tnd_lds_ptr;
__device__ __forceinline__ void tnd_dma16_x2(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory", "m0");
}

constexpr int NND_A0 = 0, NND_B0 = 2 * TND_IMG, NND_LDS = 5 * TND_IMG;     // A images x 2 | B ring x 3

__global__ __launch_bounds__(512) void k_linear_nn_direct(NndArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NND_LDS];
    __shared__ float mx_s[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the column tiles of one row tile read the same rows of A: they take workgroup ids 8 apart (one XCD, one L2)
    const int q = blockIdx.x >> 3, rt = (q / a.tiles_n) * 8 + (blockIdx.x & 7), ct = q % a.tiles_n;
    if (rt >= a.row_tiles) return;
    const int m0 = rt * TND_TILE, n0 = ct * TND_TILE;
    const int ns = a.K / TND_STEP;
    {
        float m = tid < a.namax ? a.amax[tid] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) mx_s[wave] = m;
    }
    __syncthreads();
    float vm = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) vm = fmaxf(vm, mx_s[w]);
    const int ea = split2h_exponent(vm);
    const float sa = pow2i(ea);

    auto rsrc_of = [](const void* base, int64_t first, int64_t total) {
        const int64_t left = total - first;
        const uint64_t addr = reinterpret_cast<uint64_t>(base) + (uint64_t)(left > 0 ? first : 0);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr), hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
        const int n = __builtin_amdgcn_readfirstlane((int)(left > 0 ? left : 0));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)hi << 32) | lo), (short)0, n, 0x00020000);
    };
    // loader roles.  A: row tid / 2 of the tile, k half tid & 1 (rows past R: beyond the descriptor, read as zero).  B: tile tid / 64, lane tid & 63
    const int arow = tid >> 1, ag = tid & 1;
    const int64_t bytes_a = ((a.R - 1) * a.lda + a.K) * 4;
    const unsigned oa = (unsigned)(((int64_t)(m0 + arow) * a.lda + 8 * ag) * 4);
    // B: wave w copies tile w of the column block (tiles past the operand's last: the last one again -- columns that are never stored), by
    // LDS-DMA, no register round trip (through registers the copy cost 40 us of a 409 us launch: profiles/r05_tn_direct_ab.txt)
    const int btile = min(ct * 8 + wave, a.TB - 1);
    const unsigned ob = (unsigned)((int64_t)btile * a.KBb * 2048 + lane * 16);
    const unsigned lds_base = (unsigned)(size_t)(tnd_lds_ptr)smem;
    const unsigned wa_off = (unsigned)((arow >> 5) * 2048 + ((arow & 31) + 32 * ag) * 16);
    auto dma_b = [&](int step, int slot) {                  // the step's 2 KiB of this wave's tile -> ring slot (steps past the pack's last block: the last)
        const int kb = min(step, a.KBb - 1);
        tnd_dma16_x2(a.Bpk, ob + (unsigned)kb * 2048u, __builtin_amdgcn_readfirstlane(lds_base + NND_B0 + (unsigned)slot * TND_IMG + (unsigned)wave * 2048u));
    };
    typedef float nnd_f32x4 __attribute__((ext_vector_type(4)));
    nnd_f32x4 pa0, pa1, qa0, qa1;                           // two register sets of A rows (8 floats), one converted, one in flight
#define GVQA_NND_LD4(rs_, off_) __builtin_bit_cast(nnd_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)(off_), 0, 0))
    uint4 hx, lx;

    const int wr = wave >> 2, wc = wave & 3;
    const unsigned a_off = (unsigned)(NND_A0 + wr * 4 * 2048 + lane * 16), b_off = (unsigned)(NND_B0 + wc * 2 * 2048 + lane * 16);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto rd = [&](const unsigned char* p) { return __builtin_bit_cast(tnd_f16x8, *reinterpret_cast<const uint4*>(p)); };
    tnd_f16x8 ah[4], al[4], bh[2], bl[2];
#ifndef GVQA_NND_DBG        /* A/B build switch: bit 1 no A loads in the loop, 2 no split / A image writes, 4 no MFMAs (timing only: results wrong) */
#define GVQA_NND_DBG 0
#endif
#define GVQA_NND_MF(n_) do { constexpr int q_ = (n_) / 8, t_ = (n_) % 8, i_ = t_ >> 1, j_ = t_ & 1;                                       \
        if (!(GVQA_NND_DBG & 4)) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q_ == 0 ? al[i_] : ah[i_], q_ == 2 ? bl[j_] : bh[j_], acc[i_][j_], 0, 0, 0); } while (0)
#define GVQA_NND_FENCE() __builtin_amdgcn_sched_barrier(0)
    {
        dma_b(0, 0);
        dma_b(1, 1);
        const auto r0a = rsrc_of(a.A, 0, bytes_a);
        qa0 = GVQA_NND_LD4(r0a, oa); qa1 = GVQA_NND_LD4(r0a, oa + 16);
        const auto r1a = rsrc_of(a.A, ns > 1 ? 64 : bytes_a, bytes_a);      // (steps past the last: an empty descriptor, zeros)
        pa0 = GVQA_NND_LD4(r1a, oa); pa1 = GVQA_NND_LD4(r1a, oa + 16);
        GVQA_TND_SPLIT2(hx.x, lx.x, sa, qa0.x, qa0.y); GVQA_TND_SPLIT2(hx.y, lx.y, sa, qa0.z, qa0.w);
        GVQA_TND_SPLIT2(hx.z, lx.z, sa, qa1.x, qa1.y); GVQA_TND_SPLIT2(hx.w, lx.w, sa, qa1.z, qa1.w);
        *reinterpret_cast<uint4*>(smem + NND_A0 + wa_off) = hx; *reinterpret_cast<uint4*>(smem + NND_A0 + wa_off + 1024) = lx;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) al[i] = rd(smem + a_off + i * 2048 + 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) bh[j] = rd(smem + b_off + j * 2048);
    int bcur = 0;                                           // ring slot of the current step's B tiles
    // step s_: A rows of step s + 1 (set CA_) -> image, A rows of step s + 2 -> set LA_; B tiles of step s + 2 -> ring by DMA.  At the barrier the
    // DMAs of step s + 1 (issued one step ago) must have landed.  Round 5 let "the four younger memory operations -- this step's two A loads and two
    // DMAs --" stay in flight, on the assumption that memory reads retire in issue order.  They do among LDS-DMAs and among register loads, NOT across
    // the two kinds (gine_mlp.hip, round 6: a register load counted together with younger DMAs came back after the count said "landed" -- whole rows
    // wrong in one launch of three): with four allowed, two early A loads could stand in for the two DMAs the barrier is for.  Now only this step's two
    // DMAs (the youngest operations, in order among themselves) may stay in flight; the A loads have landed -- they are converted at the top of the
    // next step anyway, so the stricter wait costs nothing measurable
#define GVQA_NND_STEP(s_, CA0_, CA1_, LA0_, LA1_)                                                                                             \
    {                                                                                                                                         \
        const int st_ = (s_);                                                                                                                 \
        const int bnext = bcur == 2 ? 0 : bcur + 1, bload = bnext == 2 ? 0 : bnext + 1;                                                       \
        const unsigned char* img = smem + (st_ & 1) * TND_IMG;                                                                                \
        const unsigned char* imn = smem + ((st_ + 1) & 1) * TND_IMG;                                                                          \
        unsigned char* dn = smem + NND_A0 + ((st_ + 1) & 1) * TND_IMG;                                                                        \
        const auto ra_ = rsrc_of(a.A, st_ + 2 < ns ? (int64_t)(st_ + 2) * 64 : bytes_a, bytes_a);                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) ah[i] = rd(img + a_off + i * 2048);                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) bl[j] = rd(smem + bcur * TND_IMG + b_off + j * 2048 + 1024);                            \
        GVQA_NND_FENCE();                                                                                                                     \
        GVQA_NND_MF(0); GVQA_NND_FENCE(); if (!(GVQA_NND_DBG & 2)) GVQA_TND_SPLIT2(hx.x, lx.x, sa, CA0_.x, CA0_.y); if (!(GVQA_NND_DBG & 1)) LA0_ = GVQA_NND_LD4(ra_, oa); GVQA_NND_FENCE();      \
        GVQA_NND_MF(1); GVQA_NND_FENCE(); if (!(GVQA_NND_DBG & 2)) GVQA_TND_SPLIT2(hx.y, lx.y, sa, CA0_.z, CA0_.w); if (!(GVQA_NND_DBG & 1)) LA1_ = GVQA_NND_LD4(ra_, oa + 16); GVQA_NND_FENCE(); \
        GVQA_NND_MF(2); GVQA_NND_FENCE(); if (!(GVQA_NND_DBG & 2)) GVQA_TND_SPLIT2(hx.z, lx.z, sa, CA1_.x, CA1_.y); GVQA_NND_FENCE();                                   \
        GVQA_NND_MF(3); GVQA_NND_FENCE(); if (!(GVQA_NND_DBG & 2)) GVQA_TND_SPLIT2(hx.w, lx.w, sa, CA1_.z, CA1_.w); GVQA_NND_FENCE();                                   \
        GVQA_NND_MF(4); GVQA_NND_FENCE();                                                                                                     \
        if (!(GVQA_NND_DBG & 2)) { *reinterpret_cast<uint4*>(dn + wa_off) = hx; *reinterpret_cast<uint4*>(dn + wa_off + 1024) = lx; }        \
        GVQA_NND_FENCE();                                                                                                                     \
        GVQA_NND_MF(5); GVQA_NND_FENCE();                                                                                                     \
        dma_b(st_ + 2, bload);                                                                                                                \
        GVQA_NND_FENCE();                                                                                                                     \
        GVQA_NND_MF(6); GVQA_NND_MF(7); GVQA_NND_FENCE();                                                                                     \
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");      /* (round 6: was vmcnt(4), see the note above the step) */         \
        __builtin_amdgcn_s_barrier();                                                                                                         \
        GVQA_NND_FENCE();                                                                                                                     \
        GVQA_NND_MF(8); GVQA_NND_MF(9); GVQA_NND_MF(10); GVQA_NND_MF(11); GVQA_NND_MF(12); GVQA_NND_MF(13); GVQA_NND_MF(14); GVQA_NND_MF(15);  \
        GVQA_NND_FENCE();                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) al[i] = rd(imn + a_off + i * 2048 + 1024);                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) bh[j] = rd(smem + bnext * TND_IMG + b_off + j * 2048);                                  \
        GVQA_NND_FENCE();                                                                                                                     \
        GVQA_NND_MF(16); GVQA_NND_MF(17); GVQA_NND_MF(18); GVQA_NND_MF(19); GVQA_NND_MF(20); GVQA_NND_MF(21); GVQA_NND_MF(22); GVQA_NND_MF(23); \
        GVQA_NND_FENCE();                                                                                                                     \
        bcur = bnext;                                                                                                                         \
    }
    for (int s = 0; s < ns; s += 2) {          // (an odd count runs one more step: A zeros -- an empty descriptor -- against B's last block again)
        GVQA_NND_STEP(s, pa0, pa1, qa0, qa1)
        GVQA_NND_STEP(s + 1, qa0, qa1, pa0, pa1)
    }
#undef GVQA_NND_STEP
#undef GVQA_NND_MF
#undef GVQA_NND_FENCE
#undef GVQA_NND_LD4

    // Epilogue through LDS, 32 rows of the tile at a time (the waves that hold them -- one row half, accumulator row i -- write their scaled
    // accumulators as a [32][256] fp32 image; then thread (row, 4 consecutive columns) adds the optional terms and moves 16 bytes): the result,
    // the addend and an accumulated-into dx all travel as 1 KiB row segments per wave instead of 4 bytes per lane, and a row of lr_g is
    // read once per four output columns.
    constexpr int EP_LD = 260;
    float* stage = reinterpret_cast<float*>(smem);
    const float ainv = pow2i(-ea);
    const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
    float binv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int gc = n0 + wc * 64 + j * 32 + ccol; binv[j] = gc < a.N ? ainv * a.b_inv[gc] : 0.f; }
    const int ecol = (tid & 63) * 4, erow0 = tid >> 6;             // this thread's 4 columns of the tile, rows erow0 + 8 k
    const int gcol = n0 + ecol;
    const bool col_on = gcol < a.N;                                // (N % 4 == 0: a column quad is inside or outside)
    // (the lr_v rows of this thread's 4 columns -- J <= 16 floats each -- are re-read from L1 per output row: kept in a register array indexed by the
    //  runtime J they were 272 bytes of scratch per lane, i.e. memory loads all the same, plus the stores and the dispatch's scratch set-up)
    const float* const vbase = a.lr_v + (int64_t)gcol * a.J;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the last steps' DMAs -- invisible to the compiler -- have landed: the stage below overlaps the ring)
    __syncthreads();                                               // every wave is done with the operand images
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        const int cwr = ch >> 2, ci = ch & 3;
        if (wr == cwr) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(crow0 + (r & 3) + 8 * (r >> 2)) * EP_LD + wc * 64 + jj * 32 + ccol] = acc[ci][jj][r] * binv[jj];
        }
        __syncthreads();
        if (col_on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lr = erow0 + 8 * k;
                const int64_t gr = m0 + cwr * 128 + ci * 32 + lr;
                if (gr >= a.R) continue;
                float4 v = *reinterpret_cast<const float4*>(&stage[lr * EP_LD + ecol]);
                if (a.lr_g) {
                    const float* gp = a.lr_g + gr * a.J;
                    for (int u = 0; 4 * u < a.J; ++u) {
                        const float4 gq = *reinterpret_cast<const float4*>(gp + 4 * u);
                        const float4 v0 = *reinterpret_cast<const float4*>(vbase + 4 * u), v1 = *reinterpret_cast<const float4*>(vbase + a.J + 4 * u);
                        const float4 v2 = *reinterpret_cast<const float4*>(vbase + 2 * a.J + 4 * u), v3 = *reinterpret_cast<const float4*>(vbase + 3 * a.J + 4 * u);
                        v.x += gq.x * v0.x + gq.y * v0.y + gq.z * v0.z + gq.w * v0.w;
                        v.y += gq.x * v1.x + gq.y * v1.y + gq.z * v1.z + gq.w * v1.w;
                        v.z += gq.x * v2.x + gq.y * v2.y + gq.z * v2.z + gq.w * v2.w;
                        v.w += gq.x * v3.x + gq.y * v3.y + gq.z * v3.z + gq.w * v3.w;
                    }
                }
                if (a.addend) {
                    const float4 t = *reinterpret_cast<const float4*>(a.addend + gr * a.ld_add + gcol);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                float* cp = a.C + gr * a.ldc + gcol;
                if (a.accumulate) {
                    const float4 t = *reinterpret_cast<const float4*>(cp);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                *reinterpret_cast<float4*>(cp) = v;
            }
        }
        __syncthreads();
    }
}

bool linear_nn_direct_applies(int64_t R, int64_t N, int64_t K, int64_t lda) {
    return R > 0 && N > 0 && N % 4 == 0 && K > 0 && K % TND_STEP == 0 && lda % 4 == 0 && (R + TND_TILE) * lda < (1ll << 29);
}

// C [R, N] (+)= A [R, K] (fp32 rows, lda, one scale from amax) x packed B [N rows, K] (TB tiles of KBb k blocks, b_inv [N])
int launch_linear_nn_direct(int64_t R, int64_t N, int64_t K, const float* A, int64_t lda, const float* amax, int namax, const void* Bpk, int KBb, int TB,
                            const float* b_inv, float* C, int64_t ldc, int accumulate, hipStream_t stream, const float* lr_g, const float* lr_v, int J,
                            const float* addend, int64_t ld_add) {
    GVQA_REQUIRE(!lr_g || (lr_v && J > 0 && J <= 16 && J % 4 == 0 && ((reinterpret_cast<uintptr_t>(lr_g) | reinterpret_cast<uintptr_t>(lr_v)) & 15) == 0),
                 GVQA_E_INVALID, "linear_nn_direct: rank-J term needs J in {4, 8, 12, 16} and 16-byte aligned operands");
    GVQA_REQUIRE(linear_nn_direct_applies(R, N, K, lda) && A && amax && Bpk && b_inv && C && namax >= 1 && namax <= 512 && KBb * TND_STEP >= K &&
                     (int64_t)TB * 32 >= N && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && ldc % 4 == 0 &&
                     (!addend || ((reinterpret_cast<uintptr_t>(addend) & 15) == 0 && ld_add % 4 == 0)),
                 GVQA_E_INVALID, "linear_nn_direct: bad argument (16-byte aligned rows)");
    NndArgs a;
    a.R = R; a.N = (int)N; a.K = (int)K; a.A = A; a.lda = lda; a.amax = amax; a.namax = namax; a.Bpk = static_cast<const uint16_t*>(Bpk); a.KBb = KBb; a.TB = TB;
    a.b_inv = b_inv; a.C = C; a.ldc = ldc; a.accumulate = accumulate; a.tiles_n = (int)cdiv(N, TND_TILE); a.row_tiles = (int)cdiv(R, TND_TILE);
    a.lr_g = lr_g; a.lr_v = lr_v; a.J = lr_g ? J : 0; a.addend = addend; a.ld_add = ld_add;
    const int64_t groups = cdiv(a.row_tiles, 8) * a.tiles_n;
    hipLaunchKernelGGL(k_linear_nn_direct, dim3((unsigned)(groups * 8)), dim3(512), 0, stream, a);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// C = A B^T from two PACKED two-piece operands (what gvqa_split2h_pack writes), in the step layout of the two kernels above: both operands'
// fragment pairs reach a three-slot ring by LDS-DMA (wave w: row tile w of A and column tile w of B, 4 DMA instructions per step, no VALU),
// the step's barrier sits behind the DMA issue with MFMAs queued on both sides, the first product's fragments are read under the previous
// step's last MFMAs; the epilogue goes through LDS in 32-row slabs (16-byte row segments; bias, addend, elementwise multiplier, ReLU / ELU).
struct PkdArgs {
    int M, N, KB;
    const uint16_t* Apk; const uint16_t* Bpk;
    const float* a_inv; const float* b_inv;
    int RTa, RTb;                                   // 32-row tiles in the packs
    LinearEpilogue ep;
    float* C; int64_t ldc;
    int tiles_n, row_tiles;
};

__global__ __launch_bounds__(512) void k_linear_pk_direct(PkdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[6 * TND_IMG];          // ring of 3 x (A image | B image)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x >> 3, rt = (q / a.tiles_n) * 8 + (blockIdx.x & 7), ct = q % a.tiles_n;    // column tiles of a row tile: one XCD
    if (rt >= a.row_tiles) return;
    const int m0 = rt * TND_TILE, n0 = ct * TND_TILE;
    const int ns = a.KB;
    const unsigned lds_base = (unsigned)(size_t)(tnd_lds_ptr)smem;
    const int atile = min(rt * 8 + wave, a.RTa - 1), btile = min(ct * 8 + wave, a.RTb - 1);      // (tiles past the packs' last: re-read it; never stored)
    const unsigned oa = (unsigned)((int64_t)atile * a.KB * 2048 + lane * 16), ob = (unsigned)((int64_t)btile * a.KB * 2048 + lane * 16);
    auto dma = [&](int step, int slot) {
        const unsigned kb = (unsigned)min(step, a.KB - 1) * 2048u;
        const unsigned dst = lds_base + (unsigned)slot * (2 * TND_IMG) + (unsigned)wave * 2048u;
        tnd_dma16_x2(a.Apk, oa + kb, __builtin_amdgcn_readfirstlane(dst));
        tnd_dma16_x2(a.Bpk, ob + kb, __builtin_amdgcn_readfirstlane(dst + TND_IMG));
    };
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned a_off = (unsigned)(wr * 4 * 2048 + lane * 16), b_off = (unsigned)(TND_IMG + wc * 2 * 2048 + lane * 16);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto rd = [&](const unsigned char* p) { return __builtin_bit_cast(tnd_f16x8, *reinterpret_cast<const uint4*>(p)); };
    tnd_f16x8 ah[4], al[4], bh[2], bl[2];
#define GVQA_PKD_MF(n_) do { constexpr int q_ = (n_) / 8, t_ = (n_) % 8, i_ = t_ >> 1, j_ = t_ & 1;                                       \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q_ == 0 ? al[i_] : ah[i_], q_ == 2 ? bl[j_] : bh[j_], acc[i_][j_], 0, 0, 0); } while (0)
#define GVQA_PKD_FENCE() __builtin_amdgcn_sched_barrier(0)
    dma(0, 0);
    dma(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) al[i] = rd(smem + a_off + i * 2048 + 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) bh[j] = rd(smem + b_off + j * 2048);
    int cur = 0;
    for (int s = 0; s < ns; ++s) {
        const int nxt = cur == 2 ? 0 : cur + 1, ld = nxt == 2 ? 0 : nxt + 1;
        const unsigned char* img = smem + cur * (2 * TND_IMG);
        const unsigned char* imn = smem + nxt * (2 * TND_IMG);
#pragma unroll
        for (int i = 0; i < 4; ++i) ah[i] = rd(img + a_off + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) bl[j] = rd(img + b_off + j * 2048 + 1024);
        GVQA_PKD_FENCE();
        GVQA_PKD_MF(0); GVQA_PKD_MF(1); GVQA_PKD_FENCE();
        dma(s + 2, ld);                                   // (slot of step s - 1: every wave read it before the previous barrier)
        GVQA_PKD_FENCE();
        GVQA_PKD_MF(2); GVQA_PKD_MF(3); GVQA_PKD_MF(4); GVQA_PKD_MF(5); GVQA_PKD_MF(6); GVQA_PKD_MF(7); GVQA_PKD_FENCE();
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");          // step s + 1's DMAs (issued one step ago) have landed; this step's four stay in flight
        __builtin_amdgcn_s_barrier();
        GVQA_PKD_FENCE();
        GVQA_PKD_MF(8); GVQA_PKD_MF(9); GVQA_PKD_MF(10); GVQA_PKD_MF(11); GVQA_PKD_MF(12); GVQA_PKD_MF(13); GVQA_PKD_MF(14); GVQA_PKD_MF(15);
        GVQA_PKD_FENCE();
#pragma unroll
        for (int i = 0; i < 4; ++i) al[i] = rd(imn + a_off + i * 2048 + 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) bh[j] = rd(imn + b_off + j * 2048);
        GVQA_PKD_FENCE();
        GVQA_PKD_MF(16); GVQA_PKD_MF(17); GVQA_PKD_MF(18); GVQA_PKD_MF(19); GVQA_PKD_MF(20); GVQA_PKD_MF(21); GVQA_PKD_MF(22); GVQA_PKD_MF(23);
        GVQA_PKD_FENCE();
        cur = nxt;
    }
#undef GVQA_PKD_MF
#undef GVQA_PKD_FENCE
    // epilogue through LDS, 32 rows at a time
    constexpr int EP_LD = 260;
    float* stage = reinterpret_cast<float*>(smem);
    const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
    float binv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int gc = n0 + wc * 64 + j * 32 + ccol; binv[j] = gc < a.N ? a.b_inv[gc] : 0.f; }
    const int ecol = (tid & 63) * 4, erow0 = tid >> 6;
    const int gcol = n0 + ecol;
    const bool col_on = gcol < a.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.ep.bias && col_on) bias4 = *reinterpret_cast<const float4*>(a.ep.bias + gcol);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        const int cwr = ch >> 2, ci = ch & 3;
        if (wr == cwr) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(crow0 + (r & 3) + 8 * (r >> 2)) * EP_LD + wc * 64 + jj * 32 + ccol] = acc[ci][jj][r] * binv[jj];
        }
        __syncthreads();
        if (col_on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lr = erow0 + 8 * k;
                const int64_t gr = m0 + cwr * 128 + ci * 32 + lr;
                if (gr >= a.M) continue;
                float4 v = *reinterpret_cast<const float4*>(&stage[lr * EP_LD + ecol]);
                const float ai = a.a_inv[gr];
                v.x = v.x * ai + bias4.x; v.y = v.y * ai + bias4.y; v.z = v.z * ai + bias4.z; v.w = v.w * ai + bias4.w;
                if (a.ep.addend) {
                    const float4 t = *reinterpret_cast<const float4*>(a.ep.addend + gr * a.ep.ld_add + gcol);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                if (a.ep.mul) {
                    const float4 t = *reinterpret_cast<const float4*>(a.ep.mul + gr * a.ep.ld_mul + gcol);
                    v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w;
                }
                if (a.ep.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                else if (a.ep.relu == 2) {
                    v.x = v.x > 0.f ? v.x : expf(v.x) - 1.f; v.y = v.y > 0.f ? v.y : expf(v.y) - 1.f;
                    v.z = v.z > 0.f ? v.z : expf(v.z) - 1.f; v.w = v.w > 0.f ? v.w : expf(v.w) - 1.f;
                }
                *reinterpret_cast<float4*>(a.C + gr * a.ldc + gcol) = v;
            }
        }
        __syncthreads();
    }
}

// plain two-piece product (no batch, chain, packed output or row-dot epilogue); KB = cdiv(K, 16) k blocks per tile in both packs
int launch_linear_pk_direct(int64_t M, int64_t N, int KB, const void* Apk, const float* a_inv, const void* Bpk, const float* b_inv, const LinearEpilogue& ep,
                            float* C, int64_t ldc, hipStream_t stream) {
    GVQA_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && KB > 0 && Apk && Bpk && a_inv && b_inv && C && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && ldc % 4 == 0 &&
                     cdiv(M, 32) * (int64_t)KB * 2048 < (1ll << 32) && cdiv(N, 32) * (int64_t)KB * 2048 < (1ll << 32),
                 GVQA_E_INVALID, "linear_pk_direct: bad argument");
    PkdArgs a;
    a.M = (int)M; a.N = (int)N; a.KB = KB; a.Apk = static_cast<const uint16_t*>(Apk); a.Bpk = static_cast<const uint16_t*>(Bpk); a.a_inv = a_inv; a.b_inv = b_inv;
    a.RTa = (int)cdiv(M, 32); a.RTb = (int)cdiv(N, 32); a.ep = ep; a.C = C; a.ldc = ldc;
    a.tiles_n = (int)cdiv(N, TND_TILE); a.row_tiles = (int)cdiv(M, TND_TILE);
    hipLaunchKernelGGL(k_linear_pk_direct, dim3((unsigned)(cdiv(a.row_tiles, 8) * a.tiles_n * 8)), dim3(512), 0, stream, a);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

bool linear_tn_direct_applies(int KC, int64_t ldx, int64_t ldy) {      // (+ 8-byte aligned operands with even leading dimensions: the launcher)
    return KC > 0 && KC % TND_STEP == 0 && ldx % 2 == 0 && ldy % 2 == 0 && (int64_t)(KC + 16) * ldx < (1ll << 29) && (int64_t)(KC + 16) * ldy < (1ll << 29);
}

// X [R, M] (ldx), Y [R, N] (ldy), chunks of KC rows (multiple of 16) -> S partial results C + z zs_c (ldc); maxima as in train.hip
int launch_linear_tn_direct(int64_t R, int64_t M, int64_t N, const float* X, int64_t ldx, const float* Y, int64_t ldy, const float* xmax, int nxmax,
                            const float* ymax, int nymax, int KC, int S, float* C, int64_t ldc, int64_t zs_c, hipStream_t stream) {
    GVQA_REQUIRE(R > 0 && M > 0 && N > 0 && X && Y && C && xmax && ymax && KC > 0 && KC % TND_STEP == 0 && S >= 1 && (int64_t)S * KC >= R && (int64_t)(KC + 16) * ldx < (1ll << 29) &&
                     (int64_t)(KC + 16) * ldy < (1ll << 29) &&
                     nxmax >= 1 && nxmax <= 512 && nymax >= 1 && nymax <= 512,
                 GVQA_E_INVALID, "linear_tn_direct: bad argument");
    TndArgs a;
    a.R = R; a.M = (int)M; a.N = (int)N; a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.xmax = xmax; a.nxmax = nxmax; a.ymax = ymax; a.nymax = nymax;
    a.KC = KC; a.S = S; a.tiles_n = (int)cdiv(N, TND_TILE); a.ntiles = (int)cdiv(M, TND_TILE) * a.tiles_n;
    a.C = C; a.ldc = ldc; a.zs_c = zs_c;
    hipLaunchKernelGGL(k_linear_tn_direct, dim3((unsigned)((int64_t)a.ntiles * S)), dim3(512), 0, stream, a);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // namespace gvqa
