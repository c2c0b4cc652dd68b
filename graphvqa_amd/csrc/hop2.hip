// The GAT hop as ONE persistent kernel with two workgroups per CU (gfx950).
//
// Same math as the fused hop of split3.hip (k_linear_split3<..., EPI = 2>): xp = h . W_h^T on the fp16 matrix cores from
// two-piece operands (split2h), then -- without xp ever reaching HBM -- the attention-weighted aggregation over incoming
// edges, head mean, per-graph instruction term, bias, skip, BatchNorm(eval), ReLU  (/root/reference gat_skip.py:133,
// 155-168, 270-275).  What is different is the shape of the execution:
//
//   * that kernel runs one 8-wave workgroup per CU (160 KiB of LDS: the 128 KiB row image of a row group aliases the
//     operand ring), so its aggregation epilogue (27 % of its time) runs with the matrix cores idle;
//   * here a workgroup is 4 waves (one per SIMD) and owns ONE row group x one 256-column block at a time: tile 128 x 256,
//     wave w holds the 128 rows x 64 columns [64 w, 64 w + 64) in 8 accumulators of 32 x 32 (the same per-wave shape).
//     Its LDS is 80 KiB -- operand ring 2 x 24 KiB, aliased after the main loop by a HALF row image (128 rows x 128
//     columns fp32 = 64 KiB: the columns of MFMA tile j of every wave, i.e. hw = 128 / H channels of every head), plus
//     16 KiB for the group's CSR slice and attention coefficients -- so TWO workgroups are resident per CU, each with its
//     own barrier.  They are started half a period apart and stay that way (equal periods): while one aggregates out of
//     LDS (VALU + LDS, no MFMA), the other has the matrix cores to itself, and when both are in their main loops they
//     share them.  The epilogue is processed in two halves (write the j = 0 tiles, aggregate, write j = 1, aggregate).
//   * workgroups are persistent (grid = 2 x CUs): workgroup w (XCD w % 8) walks a fixed list of (row group, column block)
//     items chosen so that the 64 workgroups of an XCD work on 16 row groups x half of the column blocks at any time
//     (A operand 4 MiB + weights 2 MiB through that XCD's L2, like the XCD-aware map of the 8-wave kernel).
//
// Weight rows are packed "half-interleaved": packed row 256 cb + 64 w + 32 j + t holds W[h C + cb cw + j hw + cc, :]
// with (h, cc) = divmod(32 w + t, hw), cw = 256 / H, hw = cw / 2 -- for H = 4 this IS the head-interleaved layout of
// split3.hip (k_split2h_pack<PACK_HEADS>); launch_split_pack_heads2 produces it for the other head counts.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#include "gemm_tile.h"

namespace gvqa {

typedef _Float16 h2_f16x8 __attribute__((ext_vector_type(8)));

// Chaining (CHAIN kernels): instead of fp32 rows, a hop leaves what the NEXT hop's launch consumes -- its input rows as packed
// two-piece operands, partial attention logits, per-graph maxima -- so that no pack pass runs between hops (gat.hip).
struct Hop2Chain {
    uint16_t* Pnext;          // packed output rows (slot layout of Apk; K = C) or NULL: fp32 rows to f.out (last hop)
    float* a_inv_next;        // [128 G] inverse scales of Pnext's rows
    const float* gscale;      // NULL, or [B]: the output rows' power-of-two scale per graph, decided ahead of the launch (by the coefficient
                              // kernel, which runs between the hops anyway: k_gat_alpha_groups_packed) -- then a_inv_next is written there too
    float* PMout;             // [ncb][B]: per graph, largest |output| over each column block's channels
    const float* PMin;        // [ncb][B] of the input rows (the previous hop's PMout) or NULL: bound from the rows' scales (first hop)
    const float* Tmax;        // NULL or [B]: largest |instruction term| of this hop per graph
    const float* bc;          // [4]: largest L1 norm of a weight row | largest |BN scale| | largest |BN shift| | largest |bias| of this hop
    const int32_t* graph_ptr; // [B + 1]
    int B, N;
};

struct Hop2Args {
    FusedHopArgs f;
    Hop2Chain ch;
    const uint16_t* Apk;      // packed node rows by row-group slot (k_split2h_pack<PACK_GROUPS>)
    const uint16_t* Bpk;      // packed weights of this hop, half-interleaved rows
    const float* a_inv;       // [128 G] inverse scales of the node rows
    const float* b_inv;       // [256 ncb] inverse scales of the weight rows (one value per column block)
 const float* epc;         // [3][epc_ld] bias | BatchNorm scale | shift per output channel (k_hop2_consts)
    int epc_ld;
    int KB, ncb;
    int stagger;              // start delay of the second workgroup of every CU, in s_sleep(127) units (~8128 cycles each)
#ifdef GVQA_PROBES
    int item_map;                // 0 the XCD-aware item list, 1 row-group-major (scripts/bench_hop2.py MAP=1)
    int dbg;                     // ablation switches (wrong results): 1 no epilogue, 2 no DMA in the loop, 4 no fragment reads after step 0,
                                 // 8 no waits / barriers in the loop, 16 (launcher) one workgroup per CU
    unsigned long long* probe;   // NULL or [workgroups][32 items][4 waves][8]: 100 MHz stamps 0 item start, 1 main loop end, 2/4 image of half 0/1 in place, 3/5 half aggregated, 6 item end (scripts/probe_hop2.py)
#endif
};
#ifdef GVQA_PROBES
static unsigned long long* g_hop2_probe = nullptr;
static int g_hop2_dbg = 0;
static int g_hop2_sel_every = 1, g_hop2_sel_which = 0, g_hop2_calls = 0;      // stamps from launches with call number % every == which
#define GVQA_H2_DBG(bit_) (a.dbg & (bit_))
#define GVQA_H2_STAMP(slot_) do { if (a.probe && (tid & 63) == 0 && tid < 256 && item_no < 32) a.probe[((size_t)blockIdx.x * 32 + item_no) * 32 + (tid >> 6) * 8 + (slot_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GVQA_H2_STAMP(slot_) do { } while (0)
#define GVQA_H2_DBG(bit_) false
#endif

// (device pass only: on the host pass a "v" constraint on a value of dependent type silently voids the kernel's stub)
__device__ __forceinline__ void h2_keep_live(const f32x16& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#endif
}

// largest value of v over the 64 lanes, in an SGPR: four DPP rotations inside the 16-lane rows, then the four rows by v_readlane
// (__shfl_xor is ds_bpermute: four dependent trips through the LDS crossbar)
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));      // row_ror:1
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));      // row_ror:2
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));      // row_ror:4
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));      // row_ror:8
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// NW = 4: four waves, wave w holds all 128 rows x columns [64 w, +64) (8 accumulators; <= 256 VGPRs, one wave of each workgroup
// per SIMD).  NW = 8: eight waves as 2 x 4, wave (wr, wc) holds rows [64 wr, +64) x columns [64 wc, +64) (4 accumulators; <= 128
// VGPRs, TWO waves of each workgroup per SIMD -- so a workgroup whose partner is in its epilogue still has two matrix-core waves
// on every SIMD to hide each other's fragment reads, DMA issue and barrier waits; a lone wave reaches ~80 % of the pair's rate).
template <int H, int NBUF, bool CHAIN, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void k_hop2(Hop2Args a) {
    static_assert(NW == 4 || NW == 8, "hop2: four or eight waves");
    constexpr int NT = 64 * NW;
    constexpr int TM = 16 / NW;                      // 32-row A tiles per wave
    constexpr bool PIPE = NW == 4;                   // hand-pipelined edge loop (needs the registers of the four-wave form)
    static_assert(H == 1 || H == 2 || H == 4 || H == 8, "hop2: H must be 1, 2, 4 or 8");
    static_assert(NBUF == 2 || NBUF == 3, "hop2: two or three ring stages");
    constexpr int STAGE = 12 * 2048;                 // one K step: 4 A tiles + 8 B tiles, two 1 KiB pieces each
    constexpr int CW = 256 / H;                      // channels of every head per column block
    constexpr int HW = 128 / H;                      // channels of every head in one image half
    constexpr int HC4 = HW / 4;                      // 16-byte chunks per head in an image row (32 chunks per row)
    constexpr int LPR = HW / 8;                      // lanes per row in the aggregation (8 channels each)
    constexpr int RPP = NT / LPR;                    // rows per pass of the workgroup's threads
    constexpr int ITEMS = 128 / RPP;                 // rows per thread and half
    static_assert(ITEMS >= 1, "hop2: more threads than row segments (H = 8 runs on four waves)");
    constexpr int EB = (4 / H) > 0 ? 4 / H : 1;      // edges per trip of the edge loop (8 row reads in flight, 16 at H = 8)
    constexpr int CPAD = CW < 64 ? 64 : CW;          // constants sub-arrays padded to whole 64-lane DMA instructions
    constexpr bool REGION_EARLY = NBUF * STAGE <= 64 * 1024;   // the CSR region is outside the ring: filled under the main loop
    constexpr unsigned REGION = 64 * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[80 * 1024];
    const FusedHopArgs& fh = a.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;          // wave row / column in the tile (four waves: wr = 0)
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;

    // ---- this workgroup's items: XCD x = blockIdx.x % 8 owns the row groups g = gres (mod GS) x one of CS column-block
    // ranges; its workgroups take every per_x-th item of that list, column blocks fastest
    const int x = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
#ifdef GVQA_PROBES
    const bool rgm = a.item_map == 1;            // (measurement: row-group-major -- workgroup w walks ALL column blocks of the row groups w, w + grid, ...)
#else
    constexpr bool rgm = false;
#endif
    const int CS = rgm ? 1 : (a.ncb >= 2 ? 2 : 1), GS = rgm ? (int)gridDim.x : 8 / CS;
    const int cpart = rgm ? 0 : x % CS, gres = rgm ? (int)blockIdx.x : x / CS;
    const int s_begin = rgm ? 0 : jx, s_step = rgm ? 1 : per_x;
    const int n0 = (a.ncb + CS - 1) / CS;
    const int ncp = cpart == 0 ? n0 : a.ncb - n0, cb0 = cpart == 0 ? 0 : n0;
    const int ng = fh.num_groups > gres ? (fh.num_groups - gres + GS - 1) / GS : 0;
    const int count = ng * ncp;
    if (2 * jx >= per_x)
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);

    // LDS region [64, 80) KiB, in words: rowptr | csr_src | alpha | epilogue constants bias, scale, shift of the column block
    // (CHAIN) | inverse scales of the group's input rows | scales of its graphs' output rows | per-graph output maxima
    const int src_off = 192, al_off = src_off + ((fh.e_cap + 63) & ~63), cst_off = al_off + ((fh.e_cap * H + 63) & ~63);
    // (the per-graph maxima are cleared BEFORE the main loop and flushed at the top of the NEXT item, while the operand ring --
    //  three 24 KiB stages: 8 KiB more than the 64 KiB below the region -- is live: they must sit past the ring's last byte.
    //  Small row groups (few edges: short sub-arrays) would otherwise place them inside it.)
    constexpr int RING_SPILL = REGION_EARLY ? 0 : (NBUF * STAGE - 64 * 1024) / 4;
    const int rinv_off = cst_off + 3 * CPAD, gscl_off = rinv_off + 128, gmax_off = max(gscl_off + 128, RING_SPILL);
    const Hop2Chain& ch = a.ch;
    const bool chain_out = CHAIN && ch.Pnext != nullptr;
    float* xs = reinterpret_cast<float*>(smem);
    const float inv_h = 1.0f / H;
    const bool relu = fh.bn_w != nullptr;
    const int orow = tid / LPR, oct = tid % LPR;      // first row slot / channel octet of this thread in the aggregation

    // item metadata travels one item ahead (two dependent scalar loads: off the critical path of the item that uses them)
    auto item_of = [&](int s, int& grp, int& cb) { const int gi = s / ncp; grp = gi * GS + gres; cb = cb0 + (s - gi * ncp); };
    int grp = 0, cb = 0, ns = 0, cnt = 0, e0 = 0, ne = 0;
    if (s_begin < count) {
        item_of(s_begin, grp, cb);
        ns = fh.group_ptr[grp]; cnt = fh.group_ptr[grp + 1] - ns;
        e0 = fh.rowptr[ns]; ne = min(fh.rowptr[ns + cnt] - e0, fh.e_cap);     // (e_cap: a wrong loader-side layout must not overrun the region)
    }
    [[maybe_unused]] int item_no = 0;
    [[maybe_unused]] int pm_cb = 0, pm_gf = 0, pm_n = 0, pm_par = 1;
    for (int s = s_begin; s < count; s += s_step, ++item_no) {
        __syncthreads();                              // the previous item's image and region are free
        GVQA_H2_STAMP(0);
        if constexpr (CHAIN) {
            // per-graph maxima of the previous item's column block: they anchor the next hop's scales (complete behind the barrier)
            // (two arrays, alternating by item: this item clears and fills the other one)
            if (tid < pm_n) ch.PMout[(int64_t)pm_cb * ch.B + pm_gf + tid] = __uint_as_float(reinterpret_cast<const unsigned*>(smem + REGION)[gmax_off + pm_par * 128 + tid]);
            pm_par ^= 1;
        }
        // CSR slice + coefficients of the group and the column block's constants -> LDS [REGION, ...) by LDS-DMA, 4 bytes per lane
        // (RAW global values, rebased where they are used; every sub-array padded to whole 64-lane DMA instructions, lanes past
        // the end re-load the last word)
        auto dma_region = [&]() {
            const unsigned base = lds_base + REGION;
            const int n_rp = cnt + 1, n_al = ne * H;
            const int32_t* rp_g = fh.rowptr + ns;
            const int32_t* src_g = fh.csr_src + e0;
            const float* al_g = fh.alpha_csr + (int64_t)e0 * H;
            const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
            for (int u = wbase; u < n_rp; u += NT)
                lds_dma4_b(rp_g + min(u + lane, n_rp - 1), __builtin_amdgcn_readfirstlane(base + (unsigned)u * 4u));
            for (int u = wbase; u < ne; u += NT)
                lds_dma4_b(src_g + min(u + lane, ne - 1), __builtin_amdgcn_readfirstlane(base + (unsigned)(src_off + u) * 4u));
            for (int u = wbase; u < n_al; u += NT)
                lds_dma4_b(al_g + min(u + lane, n_al - 1), __builtin_amdgcn_readfirstlane(base + (unsigned)(al_off + u) * 4u));
            // constants: one 64-lane instruction covers 64 channels of one array; wave w takes instructions w, w + 4, ...
            for (int u = wave; u < 3 * (CPAD / 64); u += NW) {
                const int arr = u / (CPAD / 64), c0 = (u % (CPAD / 64)) * 64;
                lds_dma4_b(a.epc + (int64_t)arr * a.epc_ld + cb * CW + min(c0 + lane, CW - 1),
                           __builtin_amdgcn_readfirstlane(base + (unsigned)(cst_off + arr * CPAD + c0) * 4u));
            }
            if (CHAIN && wave < 2)                    // inverse scales of the group's 128 input slots (the skip rows are read from the packed operand)
                lds_dma4_b(a.a_inv + grp * 128 + wave * 64 + lane, __builtin_amdgcn_readfirstlane(base + (unsigned)(rinv_off + wave * 64) * 4u));
        };
        if (REGION_EARLY) dma_region();               // older than every ring DMA: the first wait of the main loop covers it
        // rows of the group are aggregated in the order fh.row_order gives (most in-edges first): slot -> row and the row's
        // graph for this thread's slots, and the exact power-of-two factors of its accumulator rows -- all on their way now
        int ord[ITEMS], gid[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int slot = orow + k * RPP;
            const bool on = slot < cnt;
            ord[k] = on ? (fh.row_order ? fh.row_order[ns + slot] : slot) : 0;
            gid[k] = (CHAIN || fh.graph_term) ? fh.node_graph[ns + ord[k]] : 0;
        }
        // (CHAIN) first / last graph of the group, the graph of local row tid; the per-graph maxima start at zero
        int gf = 0, ngl = 0, my_graph = 0;
        if constexpr (CHAIN) {
            gf = fh.node_graph[ns];
            ngl = fh.node_graph[ns + cnt - 1] - gf + 1;
            my_graph = (tid < cnt ? fh.node_graph[ns + tid] : gf) - gf;
            if (tid < 128) reinterpret_cast<unsigned*>(smem + REGION)[gmax_off + pm_par * 128 + tid] = 0u;
        }
        float sab[TM];
        {
            const float sbu = a.b_inv[cb * 256];
#pragma unroll
            for (int i = 0; i < TM; ++i) sab[i] = a.a_inv[(grp * 4 + wr * TM + i) * 32 + (lane & 31)] * sbu;
        }

        // ---- main loop: acc[i][j] (+)= A tile i (rows 32 i ..) x B tile 2 wave + j over K, three piece products per K step
        f32x16 acc[TM][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const uint16_t* src[3];
        unsigned dst[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            // DMA duty of this wave, three units per K step.  Four waves: unit = operand tile wave + 4 q of the stage (both pieces, 2 KiB
            // contiguous on both sides).  Eight waves: unit = fragment wave + 8 q (tile = unit / 2, piece = unit % 2, 1 KiB).
            const int u = wave + q * NW;
            const int t = NW == 4 ? u : (u >> 1), pc = NW == 4 ? 0 : (u & 1);
            const bool isA = t < 4;
            const int tile = isA ? grp * 4 + t : cb * 8 + (t - 4);
            src[q] = (isA ? a.Apk : a.Bpk) + (int64_t)tile * a.KB * 1024 + pc * 512 + lane * 8;
            dst[q] = lds_base + t * 2048 + pc * 1024;
        }
        auto issue_pair = [&](int buf, int q) {       // unit q of the next K step (consecutive K steps of a tile are 2 KiB apart)
            if constexpr (NW == 4) lds_dma16_x2(src[q], __builtin_amdgcn_readfirstlane(dst[q] + buf * STAGE));
            else lds_dma16_b(src[q], __builtin_amdgcn_readfirstlane(dst[q] + buf * STAGE));
            src[q] += 1024;
        };
        const unsigned a_off = (unsigned)(wr * TM * 2048 + lane * 16);
        const unsigned b_off = (unsigned)((4 + wc * 2) * 2048 + lane * 16);
        const int KB = a.KB;
#pragma unroll
        for (int st = 0; st < NBUF - 1; ++st)
            if (st < KB)
#pragma unroll
                for (int q = 0; q < 3; ++q) issue_pair(st, q);
        int rb = 0, wb = NBUF - 1;                    // ring slot read in step s / filled in step s
        h2_f16x8 af[TM][2], bfr[2][2];
        for (int ks = 0; ks < KB; ++ks) {
            if (!GVQA_H2_DBG(8)) {
                if (NBUF == 3 && ks + 1 < KB) {       // step ks + 1 may stay in flight
                    if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();         // step ks has landed for every wave; the slot refilled below is read out
            }
            const bool more = ks + NBUF - 1 < KB && !GVQA_H2_DBG(2);
            const unsigned char* sb = smem + rb * STAGE;
            if (ks == 0 || !GVQA_H2_DBG(4)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    af[i][p] = __builtin_bit_cast(h2_f16x8, *reinterpret_cast<const uint4*>(sb + a_off + i * 2048 + p * 1024));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    bfr[j][p] = __builtin_bit_cast(h2_f16x8, *reinterpret_cast<const uint4*>(sb + b_off + j * 2048 + p * 1024));
            }
            // smallest cross terms first; B fragment first: transposed accumulators (a lane owns 4 consecutive columns of a row)
#define GVQA_H2_GROUP(pa_, pb_, q_)                                                                                \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)            \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bfr[j][pb_], af[i][pa_], acc[i][j], 0, 0, 0);    \
            if (more) issue_pair(wb, q_);
            GVQA_H2_GROUP(1, 0, 0) GVQA_H2_GROUP(0, 1, 1) GVQA_H2_GROUP(0, 0, 2)
#undef GVQA_H2_GROUP
            rb = rb + 1 == NBUF ? 0 : rb + 1;
            wb = wb + 1 == NBUF ? 0 : wb + 1;
        }

        // ---- epilogue: two halves (MFMA tile j of every wave = hw channels of every head)
        __syncthreads();                              // main loop done in every wave: the ring is free
        GVQA_H2_STAMP(1);
        if (GVQA_H2_DBG(1)) {                         // (measurement: main loop only, accumulators kept live)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) h2_keep_live(acc[i][j]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GVQA_H2_STAMP(6);
            if (s + s_step < count) {
                item_of(s + s_step, grp, cb);
                ns = fh.group_ptr[grp]; cnt = fh.group_ptr[grp + 1] - ns;
                e0 = fh.rowptr[ns]; ne = fh.rowptr[ns + cnt] - e0;
            }
            continue;
        }
        if (!REGION_EARLY) dma_region();
        // the next item's metadata starts its trip now
        int grp_n = 0, cb_n = 0, ns_n = 0, cnt_n = 0, e0_n = 0, ne_n = 0;
        if (s + s_step < count) {
            item_of(s + s_step, grp_n, cb_n);
            ns_n = fh.group_ptr[grp_n]; cnt_n = fh.group_ptr[grp_n + 1] - ns_n;
            e0_n = fh.rowptr[ns_n]; ne_n = min(fh.rowptr[ns_n + cnt_n] - e0_n, fh.e_cap);
        }
        const int* rp_l = reinterpret_cast<const int*>(smem + REGION);
        const int* src_l = rp_l + src_off;
        const float* al_l = reinterpret_cast<const float*>(rp_l + al_off);
        const float* cst_l = reinterpret_cast<const float*>(rp_l + cst_off);
        const float4* xs4 = reinterpret_cast<const float4*>(xs);
        [[maybe_unused]] const float* rinv_l = reinterpret_cast<const float*>(rp_l + rinv_off);
        [[maybe_unused]] float* gscl_l = reinterpret_cast<float*>(smem + REGION) + gscl_off;
        [[maybe_unused]] unsigned* gmax_l = reinterpret_cast<unsigned*>(smem + REGION) + gmax_off + pm_par * 128;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();                // the first half's image has been read out
            const int cl = half * HW + oct * 8;       // this thread's first channel within the column block
            const int c = cb * CW + cl;
            {   // accumulators -> row image xs[128][128]: 16-byte chunk ch of row r at slot ch ^ (r & 15) -- conflict-free for the
                // row-per-lane ds_write_b128; lane (m, hh) owns columns 8 q + 4 hh + 0..3 of row m of a 32 x 32 tile
                const int m = lane & 31, hh = lane >> 5;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = (wr * TM + i) * 32 + m;
                        const int chunk = wc * 8 + 2 * q + hh;
                        float4 t = make_float4(acc[i][half][4 * q], acc[i][half][4 * q + 1], acc[i][half][4 * q + 2], acc[i][half][4 * q + 3]);
                        t.x *= sab[i]; t.y *= sab[i]; t.z *= sab[i]; t.w *= sab[i];
                        *reinterpret_cast<float4*>(xs + r * 128 + ((chunk ^ (r & 15)) << 2)) = t;
                    }
            }
            if (!REGION_EARLY && half == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the late region DMAs of this wave have landed
            __syncthreads();
            if (half == 0) GVQA_H2_STAMP(2); else GVQA_H2_STAMP(4);
            if constexpr (CHAIN) {
                // Scale of the output rows, decided BEFORE they exist (they leave as fp16 pieces straight from this epilogue): one power
                // of two per graph from an upper bound of its output magnitudes --
                //   |out| <= max|BN scale| (M (max L1 norm of a weight row + 1) + max|instruction term| + max|bias|) + max|BN shift|,
                // M = the largest input magnitude in the graph (a convex combination of projected neighbour rows, plus the skip row).
                // The bound overshoots the true maximum by 2^5 .. 2^10; the two-piece split keeps 2^-22 relative accuracy down to 2^-27
                // of the scaled maximum and degrades gracefully below, so 2^15 of overshoot is still invisible at the 1e-4 bar.
                if (half == 0 && chain_out && !ch.gscale) {       // (first hop only: later hops get their scales from the coefficient kernel)
                    if (tid < ngl) {
                        const int g = gf + tid;
                        float M = 0.f;
                        if (ch.PMin) {
                            for (int q = 0; q < a.ncb; ++q) M = fmaxf(M, ch.PMin[(int64_t)q * ch.B + g]);
                        } else {                      // first hop: the rows were packed with exact scales, 2^14 / scale bounds a row
                            const int r0 = max(ch.graph_ptr[g] - ns, 0), r1 = min(ch.graph_ptr[g + 1] - ns, cnt);
                            for (int r = r0; r < r1; ++r) M = fmaxf(M, rinv_l[r]);
                            M *= 16384.f;
                        }
                        const float tm = ch.Tmax ? ch.Tmax[g] : 0.f;
                        const float bound = (ch.bc[1] * (M * (ch.bc[0] + 1.f) + tm + ch.bc[3]) + ch.bc[2]) * 1.001f;
                        gscl_l[tid] = pow2i(split2h_exponent(bound));
                    }
                    __syncthreads();
                    // inverse scales of the output slots (every column block of the group writes the same values)
                    if (tid < 128) ch.a_inv_next[grp * 128 + tid] = tid < cnt ? 1.0f / gscl_l[my_graph] : 1.f;
                }
            }
            // one output row segment: node `row` of the group, channels [c, c + 8)
            auto process = [&](int slot, int row, int gq) {
                const bool row_on = slot < cnt;
                const int i = row_on ? row : 0;
                const int node = ns + i;
                // the skip row segment and the graph's instruction term: on their way from here, consumed after the edge loop
                [[maybe_unused]] float gsc_pre = 1.f;
                if constexpr (CHAIN) { if (chain_out && ch.gscale) gsc_pre = ch.gscale[gq]; }
                float4 sq[2], pbq[2];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const bool ok = c + 4 * v < fh.C;
                    pbq[v] = (fh.graph_term && ok) ? *reinterpret_cast<const float4*>(fh.graph_term + (int64_t)gq * fh.t_ld + c + 4 * v)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (!CHAIN)
                        sq[v] = (fh.skip && ok) ? *reinterpret_cast<const float4*>(fh.skip + (int64_t)node * fh.skip_ld + c + 4 * v)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                // (CHAIN) the skip row segment comes out of the packed input operand: channels [c, c + 8) of slot i are lane
                // (i & 31) + 32 ((c >> 3) & 1) of k block c >> 4 -- 16 bytes per piece
                [[maybe_unused]] uint4 sk1 = make_uint4(0u, 0u, 0u, 0u), sk2 = sk1;
                if constexpr (CHAIN) {
                    if (c < fh.C) {
                        const uint16_t* pp = a.Apk + ((int64_t)((grp * 4 + (i >> 5)) * a.KB + (c >> 4)) * 2) * 512 + ((i & 31) + 32 * ((c >> 3) & 1)) * 8;
                        sk1 = *reinterpret_cast<const uint4*>(pp);
                        sk2 = *reinterpret_cast<const uint4*>(pp + 512);
                    }
                }
                const int lo = rp_l[i] - e0, hi = row_on ? rp_l[i + 1] - e0 : lo;
                // wave-uniform trip count (clamped slot, zero weight past the end of a row): the largest in-degree among the wave's rows
                const int maxdeg = wave_max_i32(hi - lo);
                // The edge loop is software-pipelined by hand (the waves of an epilogue are alone on their SIMDs as far as LDS latency
                // goes: the co-resident wave is the other workgroup's matrix-core loop): a trip = EB edges x HU heads; while trip t
                // is accumulated, the row reads of trip t + 1 are in flight and the indices / coefficients of trip t + 2 are fetched.
                constexpr int HCH = H > 4 ? H / 4 : 1;        // head chunks per edge
                constexpr int HU = H / HCH;                   // heads per trip
                const int trips = ((maxdeg + EB - 1) / EB) * HCH;
                float4 a4[2], b4[2];                  // two chains per quad
#pragma unroll
                for (int v = 0; v < 2; ++v) a4[v] = b4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                struct Trip { int se[EB]; float al[EB][HU]; };
                // source rows and coefficients of trip tr: unconditional reads at a clamped (always mapped) slot, zero weight past the row's end
                auto load_idx = [&](int tr, Trip& t) {
                    const int s0 = lo + (tr / HCH) * EB, hb = (tr % HCH) * HU;
#pragma unroll
                    for (int e = 0; e < EB; ++e) {
                        const bool on = s0 + e < hi;
                        const int idx = max(min(s0 + e, hi - 1), 0);
                        // (the slot is always a valid one of the group, so its source row is used as it is -- weight zero when the slot
                        //  is not this row's; selecting on `on` here makes hipcc branch around the read and drain the LDS queue behind it)
                        t.se[e] = min(max(src_l[idx] - ns, 0), 127);
                        if constexpr (HU % 4 == 0) {
#pragma unroll
                            for (int h4 = 0; h4 < HU / 4; ++h4) {
                                const float4 t4 = *reinterpret_cast<const float4*>(al_l + idx * H + hb + h4 * 4);
                                t.al[e][h4 * 4] = on ? t4.x : 0.f; t.al[e][h4 * 4 + 1] = on ? t4.y : 0.f;
                                t.al[e][h4 * 4 + 2] = on ? t4.z : 0.f; t.al[e][h4 * 4 + 3] = on ? t4.w : 0.f;
                            }
                        } else {
#pragma unroll
                            for (int h = 0; h < HU; ++h) {
                                const float t1 = al_l[idx * H + hb + h];
                                t.al[e][h] = on ? t1 : 0.f;
                            }
                        }
                    }
                };
                auto load_rows = [&](int tr, const Trip& t, float4 (&v)[EB][HU][2]) {
                    const int hb = (tr % HCH) * HU;
#pragma unroll
                    for (int e = 0; e < EB; ++e)
#pragma unroll
                        for (int h = 0; h < HU; ++h)
#pragma unroll
                            for (int w = 0; w < 2; ++w) v[e][h][w] = xs4[t.se[e] * 32 + (((hb + h) * HC4 + oct * 2 + w) ^ (t.se[e] & 15))];
                };
                auto fma_rows = [&](const Trip& t, const float4 (&v)[EB][HU][2]) {
#pragma unroll
                    for (int e = 0; e < EB; ++e)
#pragma unroll
                        for (int h = 0; h < HU; ++h)
#pragma unroll
                            for (int w = 0; w < 2; ++w) {
                                float4& q = ((e * HU + h) & 1) ? b4[w] : a4[w];
                                q.x += t.al[e][h] * v[e][h][w].x; q.y += t.al[e][h] * v[e][h][w].y;
                                q.z += t.al[e][h] * v[e][h][w].z; q.w += t.al[e][h] * v[e][h][w].w;
                            }
                };
                if constexpr (!PIPE) {                // eight waves: two epilogue waves per SIMD cover each other's latency
                    for (int tr = 0; tr < trips; ++tr) {
                        Trip tA;
                        float4 vA[EB][HU][2];
                        load_idx(tr, tA);
                        load_rows(tr, tA, vA);
                        fma_rows(tA, vA);
                    }
                } else {
                    Trip tA, tB, tC, tD;
                    float4 vA[EB][HU][2], vB[EB][HU][2];
                    load_idx(0, tA);
                    load_idx(1, tB);
                    load_rows(0, tA, vA);
                    for (int tr = 0; tr < trips; tr += 2) {       // (an odd last trip drags one all-zero-weight trip along)
                        load_rows(tr + 1, tB, vB);
                        load_idx(tr + 2, tC);
                        fma_rows(tA, vA);
                        load_rows(tr + 2, tC, vA);
                        load_idx(tr + 3, tD);
                        fma_rows(tB, vB);
                        tA = tC;
                        tB = tD;
                    }
                }
                [[maybe_unused]] float4 rr[2];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const float4 bi = *reinterpret_cast<const float4*>(cst_l + cl + 4 * w);
                    // nodes without in-edges get no instruction term (empty softmax)
                    const float4 pb = hi > lo ? pbq[w] : make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 r = make_float4((a4[w].x + b4[w].x) * inv_h + pb.x, (a4[w].y + b4[w].y) * inv_h + pb.y,
                                           (a4[w].z + b4[w].z) * inv_h + pb.z, (a4[w].w + b4[w].w) * inv_h + pb.w);
                    r.x += bi.x; r.y += bi.y; r.z += bi.z; r.w += bi.w;
                    if constexpr (CHAIN) {            // (p1 + p2) / scale: the value the projection itself saw, 2^-22 from the fp32 row
                        const h2_f16x8 h1 = __builtin_bit_cast(h2_f16x8, sk1), h2 = __builtin_bit_cast(h2_f16x8, sk2);
                        const float rs = rinv_l[i];
                        sq[w] = make_float4(((float)h1[4 * w] + (float)h2[4 * w]) * rs, ((float)h1[4 * w + 1] + (float)h2[4 * w + 1]) * rs,
                                            ((float)h1[4 * w + 2] + (float)h2[4 * w + 2]) * rs, ((float)h1[4 * w + 3] + (float)h2[4 * w + 3]) * rs);
                    }
                    r.x += sq[w].x; r.y += sq[w].y; r.z += sq[w].z; r.w += sq[w].w;
                    if (relu) {                       // torch's eval BatchNorm: y = x (w invstd) + (b - mean w invstd), then ReLU
                        const float4 sc = *reinterpret_cast<const float4*>(cst_l + CPAD + cl + 4 * w);
                        const float4 sh = *reinterpret_cast<const float4*>(cst_l + 2 * CPAD + cl + 4 * w);
                        r.x = fmaxf(r.x * sc.x + sh.x, 0.f); r.y = fmaxf(r.y * sc.y + sh.y, 0.f);
                        r.z = fmaxf(r.z * sc.z + sh.z, 0.f); r.w = fmaxf(r.w * sc.w + sh.w, 0.f);
                    }
                    const bool live = row_on && c + 4 * w < fh.C;
                    if (chain_out) rr[w] = live ? r : make_float4(0.f, 0.f, 0.f, 0.f);
                    else if (live) *reinterpret_cast<float4*>(fh.out + (int64_t)node * fh.out_ld + c + 4 * w) = r;
                }
                if constexpr (CHAIN) {
                    if (chain_out && (c >> 4) < a.KB) {       // (channels >= C inside the last k block: the operand's zero padding)
                        // the segment leaves as its 16 bytes of the next hop's two operand fragments: slot (rows past the group's end
                        // write their zeros at their own slot), k block c >> 4, lane (slot & 31) + 32 ((c >> 3) & 1)
                        const int orow_i = row_on ? i : slot;
                        const float sc = row_on ? (ch.gscale ? gsc_pre : gscl_l[gq - gf]) : 1.f;
                        const float vv[8] = {rr[0].x, rr[0].y, rr[0].z, rr[0].w, rr[1].x, rr[1].y, rr[1].z, rr[1].w};
                        h2_f16x8 p0, p1;
                        float m = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xsc = vv[e] * sc;
                            const _Float16 hi16 = (_Float16)xsc;
                            p0[e] = hi16;
                            p1[e] = (_Float16)(xsc - (float)hi16);
                            m = fmaxf(m, fabsf(vv[e]));
                        }
                        uint16_t* dstp = ch.Pnext + ((int64_t)((grp * 4 + (orow_i >> 5)) * a.KB + (c >> 4)) * 2) * 512 + ((orow_i & 31) + 32 * ((c >> 3) & 1)) * 8;
                        *reinterpret_cast<uint4*>(dstp) = __builtin_bit_cast(uint4, p0);
                        *reinterpret_cast<uint4*>(dstp + 512) = __builtin_bit_cast(uint4, p1);
                        // the graph's largest output magnitude (bit patterns of non-negative floats order like integers)
                        if (row_on) atomicMax(&gmax_l[gq - gf], __float_as_uint(m));
                    }
                }
            };
            // (whole waves enter `process`: its trip count is a wave-wide maximum; rows past the group's end are masked inside)
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                const int slot = orow + k * RPP;
                if (__builtin_amdgcn_readfirstlane(slot - (lane / LPR)) < cnt) process(slot, ord[k], gid[k]);
                else if (CHAIN && chain_out && (c >> 4) < a.KB) {       // a wave of padding slots: zero pieces
                    uint16_t* dstp = ch.Pnext + ((int64_t)((grp * 4 + (slot >> 5)) * a.KB + (c >> 4)) * 2) * 512 + ((slot & 31) + 32 * ((c >> 3) & 1)) * 8;
                    *reinterpret_cast<uint4*>(dstp) = make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(dstp + 512) = make_uint4(0u, 0u, 0u, 0u);
                }
            }
            if (half == 0) GVQA_H2_STAMP(3); else GVQA_H2_STAMP(5);
        }
        if constexpr (CHAIN) {                        // (the item's per-graph maxima leave after the barrier that opens the next item)
            pm_cb = cb; pm_gf = gf; pm_n = chain_out ? ngl : 0;
        }
        GVQA_H2_STAMP(6);
        grp = grp_n; cb = cb_n; ns = ns_n; cnt = cnt_n; e0 = e0_n; ne = ne_n;
    }
    if constexpr (CHAIN) {
        __syncthreads();
        if (tid < pm_n) ch.PMout[(int64_t)pm_cb * ch.B + pm_gf + tid] = __uint_as_float(reinterpret_cast<const unsigned*>(smem + REGION)[gmax_off + pm_par * 128 + tid]);
    }
}

// Epilogue constants of a hop, per output channel: [0] bias, [1] BatchNorm scale w / sqrt(var + eps), [2] shift b - mean * scale
// (1 and 0 without BatchNorm, 0 without bias; channels >= C likewise) -> out[3][ld], ld = channels padded to whole column blocks.
// Parameter-only: lives in the weight cache.
__global__ __launch_bounds__(256) void k_hop2_consts(int C, int ld, const float* __restrict__ bias, const float* __restrict__ bn_w,
                                                     const float* __restrict__ bn_b, const float* __restrict__ bn_m,
                                                     const float* __restrict__ bn_v, float eps, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ld) return;
    float bi = 0.f, sc = 1.f, sh = 0.f;
    if (c < C) {
        if (bias) bi = bias[c];
        if (bn_w) {
            sc = bn_w[c] * (1.0f / sqrtf(bn_v[c] + eps));
            sh = bn_b[c] - bn_m[c] * sc;
        }
    }
    out[c] = bi; out[ld + c] = sc; out[2 * ld + c] = sh;
}

int hop2_consts_ld(int H, int C) { const int cw = 256 / H; return (int)cdiv(C, cw) * cw; }

int launch_hop2_consts(int H, int C, const float* bias, const float* bn_w, const float* bn_b, const float* bn_m, const float* bn_v,
                       float eps, float* out, hipStream_t stream) {
    GVQA_REQUIRE(out && C > 0 && (H == 1 || H == 2 || H == 4 || H == 8), GVQA_E_INVALID, "hop2_consts: bad argument");
    GVQA_REQUIRE(!bn_w || (bn_b && bn_m && bn_v), GVQA_E_INVALID, "hop2_consts: BatchNorm needs weight, bias, running_mean and running_var");
    const int ld = hop2_consts_ld(H, C);
    hipLaunchKernelGGL(k_hop2_consts, dim3((unsigned)cdiv(ld, 256)), dim3(256), 0, stream, C, ld, bias, bn_w, bn_b, bn_m, bn_v, eps, out);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// (CHAIN) the four magnitudes behind the output bound of a hop (see the kernel): largest L1 norm of a node-column weight row,
// largest |BN scale|, |BN shift|, |bias| -> out[4].  One block; parameter-only, runs when the weight cache is prepared.
__global__ __launch_bounds__(1024) void k_hop2_bound_consts(int rows, int Dn, const float* __restrict__ W, int64_t ldw, int C,
                                                           const float* __restrict__ epc, int epc_ld, float* __restrict__ out) {
    __shared__ float red[4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = wave; r < rows; r += 16) {           // a wave per weight row, lanes over the node columns
        float l1 = 0.f;
        for (int k = lane; k < Dn; k += 64) l1 += fabsf(W[(int64_t)r * ldw + k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l1 += __shfl_xor(l1, o, 64);
        m[0] = fmaxf(m[0], l1);
    }
    for (int c = tid; c < C; c += 1024) {
        m[3] = fmaxf(m[3], fabsf(epc[c]));
        m[1] = fmaxf(m[1], fabsf(epc[epc_ld + c]));
        m[2] = fmaxf(m[2], fabsf(epc[2 * epc_ld + c]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = m[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        if (lane == 0) red[q][wave] = v;
    }
    __syncthreads();
    if (tid < 4) {
        float v = 0.f;
        for (int w = 0; w < 16; ++w) v = fmaxf(v, red[tid][w]);
        out[tid] = v;
    }
}

// (CHAIN) out[r] = max_c |T[r, c]|, c < C: the largest instruction-term magnitude per (hop, graph) row.  A wave per row.
__global__ __launch_bounds__(256) void k_rows_absmax(int64_t rows, int C, const float* __restrict__ T, int64_t ld, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float m = 0.f;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, fabsf(T[r * ld + c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) out[r] = m;
}

int launch_hop2_bound_consts(int H, int C, int Dn, const float* W, int64_t ldw, const float* epc, float* out, hipStream_t stream) {
    GVQA_REQUIRE(W && epc && out, GVQA_E_INVALID, "hop2_bound_consts: null argument");
    hipLaunchKernelGGL(k_hop2_bound_consts, dim3(1), dim3(1024), 0, stream, H * C, Dn, W, ldw, C, epc, hop2_consts_ld(H, C), out);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

int launch_rows_absmax(int64_t rows, int C, const float* T, int64_t ld, float* out, hipStream_t stream) {
    if (rows <= 0) return GVQA_OK;
    GVQA_REQUIRE(T && out && C > 0 && ld >= C, GVQA_E_INVALID, "rows_absmax: bad argument");
    hipLaunchKernelGGL(k_rows_absmax, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, rows, C, T, ld, out);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// edges of one row group the kernel can hold in its 16 KiB region (beside rowptr, the padding, the column block's constants and --
// chained form -- the four 128-word row / graph arrays)
size_t hop2_lds_edge_capacity(int H, bool chain) {
    return (size_t)(4096 - 192 - 128 - 3 * std::max(256 / H, 64) - (chain ? 512 : 0)) / (size_t)(H + 1);
}

static int hop2_cus() { return device_cu_count(); }

int launch_hop2(int64_t K, const void* Apk, const void* Bpk, const FusedHopArgs& f, const float* epc, const Hop2ChainDesc* cd, hipStream_t stream) {
    const bool chain = cd != nullptr;
    GVQA_REQUIRE(Apk && Bpk && epc && (f.out || (chain && cd->Pnext)) && f.group_ptr && f.rowptr && f.csr_src && f.alpha_csr && f.node_graph, GVQA_E_INVALID,
                 "hop2: null operand");
    GVQA_REQUIRE(f.H * f.cw == 256 && f.C % 4 == 0 && (f.H == 1 || f.H == 2 || f.H == 4 || f.H == 8), GVQA_E_UNSUPPORTED,
                 "hop2: needs H in {1,2,4,8} and C %% 4 == 0");
    GVQA_REQUIRE((size_t)f.e_cap <= hop2_lds_edge_capacity(f.H, chain), GVQA_E_UNSUPPORTED, "hop2: row group has too many edges for LDS");
    GVQA_REQUIRE(!chain || (K == f.C && cd->bc && cd->graph_ptr && (!cd->Pnext || cd->PMout)), GVQA_E_INVALID,
                 "hop2: chained hop needs node_dim == out_channels and its side arrays");
    if (f.num_groups == 0) return GVQA_OK;
    Hop2Args a;
    memset(&a, 0, sizeof(a));
    a.f = f;
    a.KB = (int)cdiv(K, 16);
    a.ncb = (int)cdiv(f.C, f.cw);
    a.Apk = static_cast<const uint16_t*>(Apk);
    a.Bpk = static_cast<const uint16_t*>(Bpk);
    a.a_inv = reinterpret_cast<const float*>(static_cast<const char*>(Apk) + (size_t)f.num_groups * 4 * a.KB * 2048);
    a.b_inv = reinterpret_cast<const float*>(static_cast<const char*>(Bpk) + (size_t)a.ncb * 8 * a.KB * 2048);
    // Shipped form: three ring stages, four waves, no start offset (profiles/r03_hop2_variants.txt: NBUF 2 +5 %; eight waves equal
    // without chaining and +9 % with it; a start offset between a CU's two workgroups only adds its own length).  The other
    // instantiations exist in the measurement build, selected by environment variables read once per process.
#ifdef GVQA_PROBES
    static const int stag = []() { const char* v = getenv("GVQA_HOP2_STAGGER"); return v ? atoi(v) : 0; }();
    static const int nbuf = []() { const char* v = getenv("GVQA_HOP2_NBUF"); return v ? atoi(v) : 3; }();
    static const int waves = []() { const char* v = getenv("GVQA_HOP2_WAVES"); return v ? atoi(v) : 4; }();
#else
    constexpr int stag = 0, nbuf = 3, waves = 4;
#endif
    a.stagger = stag;
    a.epc = epc;
    a.epc_ld = hop2_consts_ld(f.H, f.C);
    if (chain) {
        a.ch.Pnext = static_cast<uint16_t*>(cd->Pnext);
        a.ch.a_inv_next = cd->Pnext ? reinterpret_cast<float*>(static_cast<char*>(cd->Pnext) + (size_t)f.num_groups * 4 * a.KB * 2048) : nullptr;
        a.ch.PMout = cd->PMout; a.ch.PMin = cd->PMin; a.ch.Tmax = cd->Tmax; a.ch.bc = cd->bc; a.ch.gscale = cd->gscale;
        a.ch.graph_ptr = cd->graph_ptr; a.ch.B = cd->B; a.ch.N = cd->N;
    }
#ifdef GVQA_PROBES
    a.probe = (g_hop2_calls++ % g_hop2_sel_every) == g_hop2_sel_which ? g_hop2_probe : nullptr;
    a.dbg = g_hop2_dbg;
    static const int imap = []() { const char* v = getenv("GVQA_HOP2_MAP"); return v ? atoi(v) : 0; }();
    a.item_map = imap;
#endif
    const int64_t items = (int64_t)f.num_groups * a.ncb;
    int wgs = 2 * hop2_cus();
    wgs = (int)std::min<int64_t>(wgs, cdiv(items, 8) * 8);         // (a multiple of 8: one slice of the list per XCD)
    if (wgs < 8) wgs = 8;
#ifdef GVQA_PROBES
    if (g_hop2_dbg & 16) wgs = std::min(wgs, hop2_cus());
#endif
    const dim3 grid((unsigned)wgs);
#ifdef GVQA_PROBES
#define GVQA_H2_LAUNCH_W(H_, CH_, NW_)                                                                               \
    do {                                                                                                             \
        if (nbuf == 3) hipLaunchKernelGGL((k_hop2<H_, 3, CH_, NW_>), grid, dim3(64 * NW_), 0, stream, a);            \
        else hipLaunchKernelGGL((k_hop2<H_, 2, CH_, NW_>), grid, dim3(64 * NW_), 0, stream, a);                      \
    } while (0)
#define GVQA_H2_LAUNCH(H_, CH_) do { if (waves == 8) GVQA_H2_LAUNCH_W(H_, CH_, 8); else GVQA_H2_LAUNCH_W(H_, CH_, 4); } while (0)
#else
#define GVQA_H2_LAUNCH_W(H_, CH_, NW_) hipLaunchKernelGGL((k_hop2<H_, 3, CH_, 4>), grid, dim3(256), 0, stream, a)
#define GVQA_H2_LAUNCH(H_, CH_) GVQA_H2_LAUNCH_W(H_, CH_, 4)
    (void)nbuf; (void)waves;
#endif
    if (chain) {
        switch (f.H) {
            case 1: GVQA_H2_LAUNCH(1, true); break;
            case 2: GVQA_H2_LAUNCH(2, true); break;
            case 4: GVQA_H2_LAUNCH(4, true); break;
            default: GVQA_H2_LAUNCH_W(8, true, 4); break;
        }
    } else {
        switch (f.H) {
            case 1: GVQA_H2_LAUNCH(1, false); break;
            case 2: GVQA_H2_LAUNCH(2, false); break;
            case 4: GVQA_H2_LAUNCH(4, false); break;
            default: GVQA_H2_LAUNCH_W(8, false, 4); break;      // (H = 8: 256 row segments per half, four waves)
        }
    }
#undef GVQA_H2_LAUNCH
#undef GVQA_H2_LAUNCH_W
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // namespace gvqa

#ifdef GVQA_PROBES
// measurement build only (python -m graphvqa_amd.build --probes): device buffer for the kernel's phase stamps
extern "C" GVQA_API int gvqa_probe_hop2_buffer(void* p) { gvqa::g_hop2_probe = static_cast<unsigned long long*>(p); return 0; }
extern "C" GVQA_API int gvqa_probe_hop2_debug(int bits) { gvqa::g_hop2_dbg = bits; return 0; }
extern "C" GVQA_API int gvqa_probe_hop2_select(int every, int which) { gvqa::g_hop2_sel_every = every > 0 ? every : 1; gvqa::g_hop2_sel_which = which; gvqa::g_hop2_calls = 0; return 0; }
#endif

// resident workgroups of the hop kernel per CU as the runtime sees them (2 expected) -- tests / diagnostics
extern "C" int gvqa_hop2_blocks_per_cu(int32_t H) {
    int n = 0;
    hipError_t e;
    switch (H) {
        case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gvqa::k_hop2<1, 3, false, 4>, 256, 0); break;
        case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gvqa::k_hop2<2, 3, false, 4>, 256, 0); break;
        case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gvqa::k_hop2<4, 3, false, 4>, 256, 0); break;
        case 8: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gvqa::k_hop2<8, 3, false, 4>, 256, 0); break;
        default: return GVQA_E_INVALID;
    }
    return e == hipSuccess ? n : GVQA_E_HIP;
}
