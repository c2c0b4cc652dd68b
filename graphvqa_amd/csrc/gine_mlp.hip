// GINEConv's nn = Lin -> ReLU -> Lin as ONE kernel (gfx950): the hidden rows never leave the CU.
//
// Reference being replaced: baseline_and_test_models/pipeline_model_gine.py:628 (`GINEConv(Seq(Lin(812, 300), ReLU(), Lin(300, 300)))`), applied
// at :665 to z = (1 + eps) x_i + sum_{j -> i} relu(x_j + e_ji).  Round 5 ran it as pack pass + product + `k_gine_mid` + pack pass + product:
// 0.87 ms of config 4's 1.18 ms for 5 x 10.7 GFLOP (K = 300 is 19 K steps, N = 300 is 2.3 column tiles of 128 --
// prologues, epilogues and tile padding dominated), with the hidden [N, 300] rows written, read, packed and read again in between.
//
// Here a workgroup (4 waves, one per SIMD, 512 registers each) owns 128 rows and ALL <= 320 columns of both layers; a wave owns 32 rows:
//   * layer 1: the A operand is read straight from the fp32 rows of z (lane (m, khalf): 8 consecutive k of row m = 32 bytes, one step
//     ahead), scaled by the row's power of two (the aggregate kernel left max |z| per row), split into two fp16 pieces in registers
//     (4 v_fma_mix per register pair, as hopagg.hip) -- no pack pass, no LDS round trip for A;
//   * the accumulators of layer 1 (lane (m, hh): columns 8 q + 4 hh + 0..3 of row m per 32-column tile) ARE the A-fragment layout of
//     layer 2 up to a permutation of k inside every K step -- so layer 2's weights are packed with that permutation (k_gine_pack_w) and
//     y = relu(acc + b1 + (1 + eps) P1[g] + deg P2[g]) goes from the accumulator registers to layer 2's A fragments without touching LDS:
//     row maximum (own 160 values + the partner lane's), power-of-two scale, two fp16 pieces;
//   * both layers' weights (fragment-major two-piece images, one power-of-two scale per output column) stream through ONE LDS ring
//     (4 stages x 20 KiB, LDS-DMA three steps ahead, shared by the four waves); the ring runs across the seam, so layer 2's first
//     stages land during layer 1's epilogue.
// Products: three of the four piece products (hi x hi, lo x hi, hi x lo), fp32 accumulate -- the arithmetic of every other fp32-class
// product in this library (split3.hip).
#include <algorithm>

#include "common.h"
#include "gemm_tile.h"

namespace gvqa {

typedef _Float16 gm_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gm_f16x2 __attribute__((ext_vector_type(2)));
typedef float gm_f32x4 __attribute__((ext_vector_type(4)));

constexpr int GM_ROWS = 128;          // rows of a workgroup
constexpr int GM_TN = 10;             // 32-column tiles: C <= 320
constexpr int GM_CMAX = GM_TN * 32;
#ifndef GVQA_GM_DEEP
#define GVQA_GM_DEEP 0      /* (measured: 0.656 vs 0.664 ms per five layers -- the DMA lead is not what a step waits for; the shallow ring leaves 40 KiB of LDS free) */
#endif
constexpr int GM_NST = GVQA_GM_DEEP ? 6 : 4, GM_PD = GVQA_GM_DEEP ? 5 : 3;  // weight ring stages; weight DMAs run GM_PD steps ahead (layer 1's rows: 3)
constexpr int GM_ZD = 3;
constexpr unsigned GM_BST = GM_TN * 2048u;                        // bytes per stage: 10 tiles x two pieces x 1 KiB
constexpr unsigned GM_CC0 = GM_NST * GM_BST, GM_Z0 = GM_CC0 + 4 * GM_CMAX * 4, GM_LDS = GM_Z0 + 4 * 8192;     // + per-column constants [4][320]: 1/scale 1 | b1 | 1/scale 2 | b2; + layer 1's rows in flight (4 slots x 2 KiB per wave)
constexpr int GM_MAXQ2 = GM_CMAX / 16;

// Both weights -> fragment-major two-piece images in ONE launch.  Block (ct, which): Wk[ct][s][piece][lane][8], lane (m, khalf) holding
// W[32 ct + m, 16 s + kmap(khalf, e)] scaled by the output channel's power of two; binv[c] = its inverse.  which = 0 (layer 1): natural
// order kmap = 8 khalf + e.  which = 1 (layer 2): the order layer 1's accumulators have -- e < 4: 4 khalf + e, else 8 + 4 khalf + e - 4.
__global__ __launch_bounds__(256) void k_gine_pack_w(int C, int K1, const float* __restrict__ W1, int64_t ld1, const float* __restrict__ W2, int64_t ld2,
                                                     uint16_t* __restrict__ out1, float* __restrict__ binv1, uint16_t* __restrict__ out2,
                                                     float* __restrict__ binv2) {
    // (first version: 20 blocks, a wave per row for the maxima, then four or five K steps per wave of eight scalar loads per lane -- a chain of
    //  ~15 dependent memory round trips: 20 us per layer for 0.7 MB.  Now (ct, which, z) blocks: eight threads per row for the maximum -- every
    //  block of a column tile repeats it, 38 KB out of L2 --, then ONE K step per wave as two 16-byte loads per lane)
    __shared__ float mx_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ct = blockIdx.x, which = blockIdx.y;
    const float* W = which ? W2 : W1;
    const int64_t ldw = which ? ld2 : ld1;
    const int K = which ? C : K1, NQ = (K + 15) / 16;
    uint16_t* out = which ? out2 : out1;
    float* binv = which ? binv2 : binv1;
    {
        const int r = tid >> 3, sub = tid & 7, c = ct * 32 + r;
        float v = 0.f;
        if (c < C) {
            const float* wr = W + (int64_t)c * ldw;
            for (int k = sub * 4; k < K; k += 32) {           // (K % 4 == 0, rows 16-byte aligned)
                const float4 q = *reinterpret_cast<const float4*>(wr + k);
                v = fmaxf(v, fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w))));
            }
        }
        v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
        if (sub == 0) mx_s[r] = v;
    }
    __syncthreads();
    const int m = lane & 31, khalf = lane >> 5, c = ct * 32 + m;
    const float scale = pow2i(split2h_exponent(mx_s[m]));
    if (tid < 32 && blockIdx.z == 0) binv[ct * 32 + tid] = pow2i(-split2h_exponent(mx_s[tid]));
    for (int s = blockIdx.z * 4 + wave; s < NQ; s += 4 * gridDim.z) {
        // the lane's 8 values: k = 16 s + (layer 1: 8 khalf + 0..7 | layer 2: 4 khalf + 0..3 and 8 + 4 khalf + 0..3)
        const int ka = 16 * s + (which ? 4 * khalf : 8 * khalf), kb = which ? ka + 8 : ka + 4;
        float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;
        if (c < C) {
            if (ka + 4 <= K) qa = *reinterpret_cast<const float4*>(W + (int64_t)c * ldw + ka);
            if (kb + 4 <= K) qb = *reinterpret_cast<const float4*>(W + (int64_t)c * ldw + kb);
        }
        const float w[8] = {qa.x * scale, qa.y * scale, qa.z * scale, qa.w * scale, qb.x * scale, qb.y * scale, qb.z * scale, qb.w * scale};
        gm_f16x8 p0, p1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 hi = (_Float16)w[e];
            p0[e] = hi;
            p1[e] = (_Float16)(w[e] - (float)hi);
        }
        uint16_t* dst = out + ((int64_t)(ct * NQ + s) * 2) * 512 + lane * 8;
        *reinterpret_cast<uint4*>(dst) = __builtin_bit_cast(uint4, p0);
        *reinterpret_cast<uint4*>(dst + 512) = __builtin_bit_cast(uint4, p1);
    }
}

struct GineMlpArgs {
    const float* z; int64_t ldz;        // [N, Dn] fp32 rows (16-byte aligned, ldz % 4 == 0)
    const float* zmax;                  // [N] largest |z| per row
    int N, Dn, C, NQ1, NQ2;
    const uint16_t *W1pk, *W2pk;
    const float *binv1, *binv2, *b1, *b2;
    const float *P1, *P2; int64_t ldp;  // NULL or [B, ldp]: per-graph instruction shares of layer 1
    const int32_t *node_graph, *rowptr;
    float eps;
    float* out; int64_t ldo;
};

// LDS-DMA, global address = SGPR base + 32-bit VGPR offset: 2 x 16 bytes per lane, 1 KiB apart on both sides (as hopagg.hip's)
__device__ __forceinline__ void gm_dma16_x2(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory", "m0");
}
template <typename T>
__device__ __forceinline__ const T* gm_uniform(const T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}
// hi = f16(p a), f16(p b); lo = f16(p a - hi), ...: one v_fma_mix per half (hopagg.hip's GVQA_HA_SPLIT2)
#if defined(__HIP_DEVICE_COMPILE__)
#define GVQA_GM_SPLIT2(hi_, lo_, p_, a_, b_)                                                                                          \
    asm("v_fma_mixlo_f16 %0, %2, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %4, 0\n\t"                                                          \
        "v_fma_mixlo_f16 %1, %2, %3, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %2, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"       \
        : "=&v"(hi_), "=&v"(lo_) : "v"(p_), "v"(a_), "v"(b_))
#else
#define GVQA_GM_SPLIT2(hi_, lo_, p_, a_, b_) do { const float ta_ = (a_) * (p_), tb_ = (b_) * (p_); gm_f16x2 h_, l_; h_[0] = (_Float16)ta_; h_[1] = (_Float16)tb_; \
        l_[0] = (_Float16)(ta_ - (float)h_[0]); l_[1] = (_Float16)(tb_ - (float)h_[1]); (hi_) = __builtin_bit_cast(unsigned, h_); (lo_) = __builtin_bit_cast(unsigned, l_); } while (0)
#endif

__global__ __launch_bounds__(256, 1) void k_gine_mlp(GineMlpArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[GM_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;
    const int m = lane & 31, hh = lane >> 5;
    const int row = blockIdx.x * GM_ROWS + wave * 32 + m;
    const bool on = row < a.N;
    const int rowc = min(row, a.N - 1);
    const int NQ1 = a.NQ1, NQ2 = a.NQ2, NS = NQ1 + NQ2, C = a.C, Dn = a.Dn;
    float* cc_l = reinterpret_cast<float*>(smem + GM_CC0);
    for (int i = tid; i < 4 * GM_CMAX; i += 256) {
        const int k = i / GM_CMAX, c = i - k * GM_CMAX;
        const float* src = k == 0 ? a.binv1 : k == 1 ? a.b1 : k == 2 ? a.binv2 : a.b2;
        cc_l[i] = (k & 1) ? (c < C ? src[c] : 0.f) : src[c];       // (the pack pass wrote a scale for every column of the 10 tiles; biases are zero past C)
    }
    // ---- weight ring: unit (tile u, step g) = 2 KiB (both pieces); wave w issues tiles w, w + 4 and (w < 2) w + 8 of every step.  Steps
    // 0 .. NQ1 - 1 come from layer 1's image, NQ1 .. NS - 1 from layer 2's; steps past the end re-load the last one into a free slot, so
    // that every step issues the same number of DMA instructions and the counted waits hold everywhere
    const int nd = wave < 2 ? 3 : 2;                   // (wave-uniform)
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_b = [&](int g) {
#ifdef GVQA_GM_NODMA            /* (timing variant, results wrong: no weight DMAs after the priming ones -- what their issue costs a step) */
        if (g >= GM_PD) return;
#endif
        const int gs = min(g, NS - 1);
        const bool l2 = gs >= NQ1;
        const uint16_t* img = l2 ? a.W2pk : a.W1pk;
        const int nq = l2 ? NQ2 : NQ1, s = l2 ? gs - NQ1 : gs;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = wave + 4 * i;
            if (i < nd)
                gm_dma16_x2(gm_uniform(img + ((int64_t)u * nq + s) * 1024), lane16,
                            __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(g % GM_NST) * GM_BST + (unsigned)u * 2048u));
        }
    };
    // ---- layer 1's A operand: lane (m, hh) needs k = 16 s + 8 hh + 0..7 of its row, one step ahead.  The two quads travel by LDS-DMA into
    // the lane's own 2 x 16 bytes of a wave-private double buffer and are read back after the step's counted wait.  (First version: plain
    // global loads into registers, counted together with the weight DMAs -- whole rows came out wrong now and then: an LDS-DMA and a
    // register load do NOT retire in issue order, so `vmcnt(n)` said "landed" while the rows were still on their way.  DMAs among
    // themselves do retire in order -- every ring in this library counts on it.)  Quads past the row's end are fetched from the slack
    // behind it and zeroed at use.
    const float* zrow = a.z + (int64_t)rowc * a.ldz + 8 * hh;
    const unsigned zl0 = GM_Z0 + (unsigned)wave * 8192u;        // this wave's [4 slots][2 quads][64 lanes][16 bytes]
    auto zload = [&](int s) {                          // (steps past the last: the last step's rows again, into a free slot)
#ifdef GVQA_GM_NODMA
        if (s >= GM_ZD) return;
#endif
        const float* p_ = zrow + 16 * min(s, NQ1 - 1);
        const unsigned d_ = __builtin_amdgcn_readfirstlane(lds_base + zl0 + (unsigned)(s & 3) * 2048u);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:16"
                     :
                     : "v"(p_), "s"(d_), "s"(d_ + 1024u - 16u)
                     : "memory", "m0");
    };
#ifdef GVQA_GM_STAMPS          /* measurement variant: phase stamps (100 MHz) of wave 0 overwrite the first 12 floats of the workgroup's first output row */
    unsigned long long st_[6];
#define GVQA_GM_STAMP(k_) st_[k_] = __builtin_amdgcn_s_memrealtime()
#else
#define GVQA_GM_STAMP(k_) do { } while (0)
#endif
    GVQA_GM_STAMP(0);
    const int ex1 = split2h_exponent(a.zmax[rowc]);
    const float ps1 = pow2i(ex1);
    f32x16 acc[GM_TN];
#pragma unroll
    for (int j = 0; j < GM_TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int g = 0; g < GM_PD; ++g) issue_b(g);
#pragma unroll
    for (int g = 0; g < GM_ZD; ++g) zload(g);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto rd = [&](const unsigned char* p_) { return __builtin_bit_cast(gm_f16x8, *reinterpret_cast<const uint4*>(p_)); };
    // one K step of either layer: 10 column tiles x three piece products, B fragments from ring stage g % GM_NST in two groups of five
    // tiles (product-major inside a group: five independent MFMAs between two on the same accumulator).
    // (Tried, round 6: the step ROTATED around its barrier -- the second group's fragments read before the barrier, multiplied behind it
    //  under the next step's first reads, so that no step opens with an exposed LDS round trip: layer 1 19.1 -> 20.3 us, layer 2 14.7 ->
    //  16.1 us.  The read latency is not what a step waits for.  What the ISA and the guide's price list point at instead: the step's
    //  6-8 LDS-DMA instructions per wave -- 60-185 cycles of issue each beside MFMAs -- are ~500 cycles per SIMD and step, the size of
    //  the gap between the step's 0.46 us of MFMAs and its 0.75-1.0 us.  A K step moves 28 KiB per CU whichever wave asks for it.)
    auto mma_step = [&](int g, const gm_f16x8& afh, const gm_f16x8& afl) {
        const unsigned char* sb = smem + (g % GM_NST) * GM_BST + lane * 16;
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            gm_f16x8 bh[5], bl[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                bh[j] = rd(sb + (grp * 5 + j) * 2048);
                bl[j] = rd(sb + (grp * 5 + j) * 2048 + 1024);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[grp * 5 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], afl, acc[grp * 5 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[grp * 5 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], afh, acc[grp * 5 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[grp * 5 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], afh, acc[grp * 5 + j], 0, 0, 0);
        }
    };
    GVQA_GM_STAMP(1);
    // ---- layer 1 ----
    // step s: rows of step s + 1 requested; rows of step s -> two pieces; DMAs of step s + GM_PD; products.  End of the step: only the DMAs
    // just issued may stay in flight (loads retire in order: the rows of step s + 1 and the DMAs of step s + 2, issued a step ago, have landed)
    for (int s = 0; s < NQ1; ++s) {
        const gm_f32x4 zc0 = *reinterpret_cast<const gm_f32x4*>(smem + zl0 + (s & 3) * 2048 + lane * 16);
        const gm_f32x4 zc1 = *reinterpret_cast<const gm_f32x4*>(smem + zl0 + (s & 3) * 2048 + 1024 + lane * 16);
        zload(s + GM_ZD);
        const int k0 = 16 * s + 8 * hh;
        const gm_f32x4 zero4 = gm_f32x4{0.f, 0.f, 0.f, 0.f};
        const gm_f32x4 v0 = k0 + 4 <= Dn ? zc0 : zero4, v1 = k0 + 8 <= Dn ? zc1 : zero4;      // (k >= Dn contributes nothing -- and must not meet the fp16 range with another row's values)
        uint4 ah, al;
        GVQA_GM_SPLIT2(ah.x, al.x, ps1, v0[0], v0[1]); GVQA_GM_SPLIT2(ah.y, al.y, ps1, v0[2], v0[3]);
        GVQA_GM_SPLIT2(ah.z, al.z, ps1, v1[0], v1[1]); GVQA_GM_SPLIT2(ah.w, al.w, ps1, v1[2], v1[3]);
        issue_b(s + GM_PD);
        mma_step(s, __builtin_bit_cast(gm_f16x8, ah), __builtin_bit_cast(gm_f16x8, al));
        // rows and weights of step s + 1 (issued two steps ago) have landed; what this step and the previous one issued -- 2 x (2 row DMAs
        // + 2 per weight tile) -- may stay in flight: two full steps of lead (one step of lead measured 114 us per launch: every step
        // waited out an L2 round trip)
        // (deep ring, GM_PD = 5: + the weight DMAs of step s - 2, issued behind the rows this wait is for: 2 x (2 + 2 nd) + 2 nd)
        if (GM_PD == 5) { if (nd == 3) asm volatile("s_waitcnt vmcnt(22) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); }
        else if (nd == 3) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    GVQA_GM_STAMP(2);
    // ---- between the layers: y = relu(acc / (row scale x column scale) + b1 + (1 + eps) P1[g] + deg P2[g])  (k_gine_mid's arithmetic,
    // pipeline_model_gine.py:628's ReLU), its row maximum, two fp16 pieces per value -> layer 2's A fragments, all in registers
    const float rinv1 = pow2i(-ex1);
    const int gidx = a.P1 ? a.node_graph[rowc] : 0;
    const float deg = a.P1 ? (float)(a.rowptr[rowc + 1] - a.rowptr[rowc]) : 0.f;
    const float e1 = 1.0f + a.eps;
    const float* p1row = a.P1 ? a.P1 + (int64_t)gidx * a.ldp : cc_l;      // (no instruction shares: a mapped address, the factors below are 0)
    const float* p2row = a.P1 ? a.P2 + (int64_t)gidx * a.ldp : cc_l;
    const float e1p = a.P1 ? e1 : 0.f, degp = a.P1 ? deg : 0.f;
    auto yquad = [&](int j, int q, float (&y)[4], const float* cc_, const float4& p1, const float4& p2) {
        const int c0 = 32 * j + 8 * q + 4 * hh;
        const bool live = c0 < C;
        const float4 bv = *reinterpret_cast<const float4*>(cc_ + c0);
        const float4 bb = *reinterpret_cast<const float4*>(cc_ + GM_CMAX + c0);
        // (each accumulator element is read by an explicit, volatile v_accvgpr_read: the tile stays in its accumulator registers --
        //  hipcc otherwise copies whole 16-register tiles to VGPRs, keeps the first pass's copies for the second, and spills 270 .. 600 registers)
        float a0, a1, a2, a3;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a0) : "a"(acc[j][4 * q]));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a1) : "a"(acc[j][4 * q + 1]));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a2) : "a"(acc[j][4 * q + 2]));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a3) : "a"(acc[j][4 * q + 3]));
#else
        a0 = acc[j][4 * q]; a1 = acc[j][4 * q + 1]; a2 = acc[j][4 * q + 2]; a3 = acc[j][4 * q + 3];
#endif
        y[0] = a0 * (rinv1 * bv.x) + bb.x + e1p * p1.x + degp * p2.x;
        y[1] = a1 * (rinv1 * bv.y) + bb.y + e1p * p1.y + degp * p2.y;
        y[2] = a2 * (rinv1 * bv.z) + bb.z + e1p * p1.z + degp * p2.z;
        y[3] = a3 * (rinv1 * bv.w) + bb.w + e1p * p1.w + degp * p2.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = live ? fmaxf(y[i], 0.f) : 0.f;
    };
    // pass 1: y of every column, the row maximum, y back into the accumulator registers it came from (explicit v_accvgpr_write, element
    // by element) -- the second pass needs no loads.  The per-graph rows are fetched for FIVE tiles at a time (40 x 16 bytes per lane in
    // flight: one tile at a time was ten exposed L2 round trips, 12 of the kernel's 56 us)
    float ymax = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float4 p1v[5][4], p2v[5][4];
#pragma unroll
        for (int jj = 0; jj < 5; ++jj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * (5 * half + jj) + 8 * q + 4 * hh, cl = c0 < C ? c0 : 0;
                p1v[jj][q] = *reinterpret_cast<const float4*>(p1row + cl);
                p2v[jj][q] = *reinterpret_cast<const float4*>(p2row + cl);
            }
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int j = 5 * half + jj;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float y[4];
                yquad(j, q, y, cc_l, p1v[jj][q], p2v[jj][q]);
                ymax = fmaxf(ymax, fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])));
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[j][4 * q]) : "v"(y[0]));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[j][4 * q + 1]) : "v"(y[1]));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[j][4 * q + 2]) : "v"(y[2]));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[j][4 * q + 3]) : "v"(y[3]));
#else
                acc[j][4 * q] = y[0]; acc[j][4 * q + 1] = y[1]; acc[j][4 * q + 2] = y[2]; acc[j][4 * q + 3] = y[3];
#endif
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    ymax = fmaxf(ymax, __shfl_xor(ymax, 32, 64));
    const int ex2 = split2h_exponent(ymax);
    const float ps2 = pow2i(ex2), rinv2 = pow2i(-ex2);
    unsigned a2h[GM_TN][8], a2l[GM_TN][8];             // [tile][2 q + pair]: K step s of layer 2 = tile s / 2, registers 4 (s & 1) .. + 3
#pragma unroll
    for (int j = 0; j < GM_TN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y[4];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y[0]) : "a"(acc[j][4 * q]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y[1]) : "a"(acc[j][4 * q + 1]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y[2]) : "a"(acc[j][4 * q + 2]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y[3]) : "a"(acc[j][4 * q + 3]));
#else
            y[0] = acc[j][4 * q]; y[1] = acc[j][4 * q + 1]; y[2] = acc[j][4 * q + 2]; y[3] = acc[j][4 * q + 3];
#endif
            GVQA_GM_SPLIT2(a2h[j][2 * q], a2l[j][2 * q], ps2, y[0], y[1]);
            GVQA_GM_SPLIT2(a2h[j][2 * q + 1], a2l[j][2 * q + 1], ps2, y[2], y[3]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
    }
    GVQA_GM_STAMP(3);
    // ---- layer 2: A from registers ----
#pragma unroll
    for (int s = 0; s < GM_MAXQ2; ++s) {
        if (s < NQ2) {                                  // (block-uniform)
            const int j = s >> 1, o = 4 * (s & 1);
            const uint4 ah = make_uint4(a2h[j][o], a2h[j][o + 1], a2h[j][o + 2], a2h[j][o + 3]);
            const uint4 al = make_uint4(a2l[j][o], a2l[j][o + 1], a2l[j][o + 2], a2l[j][o + 3]);
            issue_b(NQ1 + s + GM_PD);
            mma_step(NQ1 + s, __builtin_bit_cast(gm_f16x8, ah), __builtin_bit_cast(gm_f16x8, al));
            if (GM_PD == 5) { if (nd == 3) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); }     // (deep ring: three steps' weight DMAs)
            else if (nd == 3) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");       // (the weight DMAs of this step and the previous one may stay in flight)
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    GVQA_GM_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the clamped re-loads of the last steps: nothing may land in LDS after the workgroup has gone)
    // ---- out = acc / (row scale x column scale) + b2 ----
    if (on) {
        float* orow = a.out + (int64_t)row * a.ldo;
#pragma unroll
        for (int j = 0; j < GM_TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * j + 8 * q + 4 * hh;
                if (c0 >= C) continue;
                const float4 bv = *reinterpret_cast<const float4*>(cc_l + 2 * GM_CMAX + c0);
                const float4 bb = *reinterpret_cast<const float4*>(cc_l + 3 * GM_CMAX + c0);
                float4 v;
                v.x = acc[j][4 * q] * (rinv2 * bv.x) + bb.x;
                v.y = acc[j][4 * q + 1] * (rinv2 * bv.y) + bb.y;
                v.z = acc[j][4 * q + 2] * (rinv2 * bv.z) + bb.z;
                v.w = acc[j][4 * q + 3] * (rinv2 * bv.w) + bb.w;
                *reinterpret_cast<float4*>(orow + c0) = v;
            }
    }
#ifdef GVQA_GM_STAMPS
    GVQA_GM_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out + (int64_t)(blockIdx.x * GM_ROWS) * a.ldo);
        for (int k = 0; k < 6; ++k) dst[k] = st_[k];
    }
#endif
#undef GVQA_GM_STAMP
}

size_t gine_mlp_packed_bytes(int C, int Dn) {
    const size_t nq1 = (size_t)cdiv(Dn, 16), nq2 = (size_t)cdiv(C, 16);
    return GM_TN * (nq1 + nq2) * 2048 + 2 * GM_CMAX * sizeof(float) + 256;
}

bool gine_mlp_supported(int64_t N, int C, int Dn, const float* z, int64_t ldz, const float* out, int64_t ldo, int64_t ldp) {
    return N > 0 && C >= 8 && C <= GM_CMAX && C % 4 == 0 && Dn >= 8 && Dn % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 && ldp % 4 == 0 &&
           ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
}

// z [N, Dn] (+ zmax [N]) -> out [N, C] = Lin2(relu(Lin1(z) + instruction shares)); `packed`: gine_mlp_packed_bytes of scratch (256-byte aligned)
int launch_gine_mlp(int64_t N, int C, int Dn, const float* z, int64_t ldz, const float* zmax, const float* W1, int64_t ld1, const float* b1,
                    const float* W2, int64_t ld2, const float* b2, const float* P1, const float* P2, int64_t ldp, const int32_t* node_graph,
                    const int32_t* rowptr, float eps, float* out, int64_t ldo, void* packed, hipStream_t stream) {
    GVQA_REQUIRE(gine_mlp_supported(N, C, Dn, z, ldz, out, ldo, ldp) && zmax && W1 && W2 && b1 && b2 && packed && (!P1 == !P2) && (!P1 || (node_graph && rowptr)),
                 GVQA_E_INVALID, "gine_mlp: bad argument");
    GVQA_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && ((reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(W2)) & 15) == 0, GVQA_E_INVALID,
                 "gine_mlp: weight rows must be 16-byte aligned");
    const int nq1 = (int)cdiv(Dn, 16), nq2 = (int)cdiv(C, 16);
    char* base = static_cast<char*>(packed);
    uint16_t* w1pk = reinterpret_cast<uint16_t*>(base);
    uint16_t* w2pk = reinterpret_cast<uint16_t*>(base + (size_t)GM_TN * nq1 * 2048);
    float* binv1 = reinterpret_cast<float*>(base + (size_t)GM_TN * (nq1 + nq2) * 2048);
    float* binv2 = binv1 + GM_CMAX;
    hipLaunchKernelGGL(k_gine_pack_w, dim3(GM_TN, 2, (unsigned)cdiv(std::max(nq1, nq2), 4)), dim3(256), 0, stream, C, Dn, W1, ld1, W2, ld2, w1pk, binv1, w2pk, binv2);
    GVQA_LAUNCH_CHECK();
    GineMlpArgs a;
    a.z = z; a.ldz = ldz; a.zmax = zmax; a.N = (int)N; a.Dn = Dn; a.C = C; a.NQ1 = nq1; a.NQ2 = nq2;
    a.W1pk = w1pk; a.W2pk = w2pk; a.binv1 = binv1; a.binv2 = binv2; a.b1 = b1; a.b2 = b2;
    a.P1 = P1; a.P2 = P2; a.ldp = ldp; a.node_graph = node_graph; a.rowptr = rowptr; a.eps = eps; a.out = out; a.ldo = ldo;
    hipLaunchKernelGGL(k_gine_mlp, dim3((unsigned)cdiv(N, GM_ROWS)), dim3(256), 0, stream, a);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // namespace gvqa
