// Internal helpers shared by the HIP translation units of libgvqa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gvqa.h"

namespace gvqa {

// CUs of the CURRENT device, looked up once per device ordinal (the hop-kernel rules size their grids / thresholds by it; a process-wide
// static would hand device 0's count to a caller working on another device -- ADVICE r04).  Lock-free: a racing first call stores the same value.
inline int device_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cached[dev];
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
    return n;
}


void set_error(const char* fmt, ...);
int get_option(int option);          // gvqa_set_option / environment (capi.hip)

#define GVQA_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::gvqa::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                              __FILE__, __LINE__);                                        \
            return GVQA_E_HIP;                                                            \
        }                                                                                 \
    } while (0)

#define GVQA_REQUIRE(cond, code, ...)          \
    do {                                       \
        if (!(cond)) {                         \
            ::gvqa::set_error(__VA_ARGS__);    \
            return (code);                     \
        }                                      \
    } while (0)

// launch-error check without synchronising
#define GVQA_LAUNCH_CHECK() GVQA_HIP_CHECK(hipGetLastError())

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over a caller-provided workspace; 256-byte aligned slices.
struct Carver {
    char* base;
    size_t cap, off;
    Carver(void* p, size_t n) : base(static_cast<char*>(p)), cap(n), off(0) {}
    template <typename T>
    T* take(size_t count) {
        size_t bytes = align_up(count * sizeof(T), 256);
        T* r = reinterpret_cast<T*>(base + off);
        off += bytes;
        return r;
    }
    bool ok() const { return off <= cap; }
};

// Side stream (capi.hip): one non-blocking stream and a pair of events per host thread and device.  fork = the side stream waits for
// everything already enqueued on the caller's stream, join = the caller's stream waits for the side stream.  For work that is
// LATENCY-bound and independent of what the caller's stream runs meanwhile (LCGN's per-question command chain beside the node products);
// HBM-bound kernels beside each other gained nothing (gat.hip, rounds 1 / 4 / 6: GVQA_OVERLAP stays off there).
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    int device = -1;
    bool ok = false;
};
SideStream* side_stream_get();
int side_fork(SideStream* ss, hipStream_t main);
int side_join(SideStream* ss, hipStream_t main);

// Stage timing (gvqa_prof_*): RAII bracket recording two events on the stream when enabled.
struct StageTimer {
    int stage;
    hipStream_t stream;
    void* slot;
    StageTimer(int stage, hipStream_t s);
    ~StageTimer();
};

// bf16 storage helpers (node tensors of the LCGN bf16-node-feature mode): fp32 compute, bf16 in HBM
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {      // round to nearest even
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
template <bool H16>
__device__ __forceinline__ float load_elem(const void* p, int64_t i) {
    if (H16) return bf16_to_f32(static_cast<const uint16_t*>(p)[i]);
    return static_cast<const float*>(p)[i];
}

// Epilogue of the dense projection: v = acc + bias[n]; v += addend[m,n]; v *= mul[m,n]; relu.
struct LinearEpilogue {
    const float* bias;      // [N] or NULL
    const float* addend;    // [M, ld_add] or NULL (may alias C: accumulate in place); bf16 when C is bf16
    int64_t ld_add;
    const float* mul;       // [M, ld_mul] or NULL; bf16 when C is bf16
    int64_t ld_mul;
    int relu;               // activation: 0 none, 1 ReLU, 2 ELU(alpha = 1)
    // batched launches of the split GEMM (gridDim.z > 1; split-K chunks of the TN product): strides per batch of the packed
    // operands (16-bit elements), of C (floats) and of the operands' inverse-scale arrays (floats)
    int64_t zs_a, zs_b, zs_c, zs_ia, zs_ib;
    // split GEMM, two-piece operands: the finished rows' dot product with a vector, in 16 partial sums per row (slot = the wave
    // column that owns those output columns; unused slots are not written: clear the array first).  rowdot_out [M, 16];
    // sum over a row's 16 slots = (finished row) . rowdot_w[0:N].  C may be NULL then (the rows themselves are not stored).
    const float* rowdot_w;
    float* rowdot_out;
    // split GEMM, two-piece operands, N <= 512: the finished rows written as the NEXT product's packed A operand (the layout and
    // scales k_split2h_pack would give them; a buffer of split_packed_bytes(2, M, N) bytes) -- a chain of products without a pack
    // pass in between.  C may be NULL (or receives the fp32 rows as well).  pk_mul [*, pk_mul_ld] / pk_mul_idx [M]: the packed value
    // is the finished value times row pk_mul_idx[r] of pk_mul.  (pk_inv / pk_KB / pk_RT are filled in by the launcher.)
    uint16_t* pk_out;
    const float* pk_mul;
    const int32_t* pk_mul_idx;
    int64_t pk_mul_ld;
    float* pk_inv;
    int pk_KB, pk_RT;
    // split GEMM, two-piece operands: the A operand as TWO K segments with their own row scales -- k blocks [0, a2_kb0) from the packed
    // A operand itself (an image of a2_kb0 k blocks), k blocks [a2_kb0, KB) from `a2` (a packed image of the same M rows, a2_KB = KB -
    // a2_kb0 k blocks, inverse row scales a2_inv).  At the switch the accumulators are multiplied by the rows' ratio of the two scales
    // (powers of two: exact).  What it is for: [u | v] . W^T where u and v were packed by different producers (epilogues of earlier
    // products), without a pack pass over the concatenation (LCGN: [prod | x_ctx], [x_ctx | msg], lcgn.py:313,316).
    const uint16_t* a2;
    const float* a2_inv;
    int a2_kb0, a2_KB;
};

// GEMM entry used by the orchestration code (defined in gemm.hip).
// C[M,N] = A[M,K] . B[N,K]^T (+bias) (relu);  batched over blockIdx.z with element strides.
int launch_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                  int64_t ldb, const float* bias, int relu, float* C, int64_t ldc, int batch,
                  int64_t strideA, int64_t strideB, int64_t strideC, hipStream_t stream);
int launch_linear_ex(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                     int64_t ldb, LinearEpilogue ep, float* C, int64_t ldc, int batch, int64_t strideA,
                     int64_t strideB, int64_t strideC, hipStream_t stream);

int launch_linear_t(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                    int64_t ldb, LinearEpilogue ep, float* C, int64_t ldc, int batch, int64_t strideA,
                    int64_t strideB, int64_t strideC, int dtype_flags, hipStream_t stream);
// bf16 matrix-core projection of a bf16 node tensor against weights packed into P bf16 pieces (gemm_bf16.hip)
int launch_pack_weight_bf16(int64_t rows, int K, int Kp, int P, const float* W, int64_t ldw, void* out, hipStream_t stream);
bool linear_bf16_supported(int64_t K, int64_t lda, const void* A, const void* Wpk);
int launch_linear_bf16(int64_t M, int64_t N, int64_t K, int P, const void* A, int64_t lda, const void* Wpk, LinearEpilogue ep,
                       void* C, int64_t ldc, bool c16, hipStream_t stream);
// Fused hop (split3.hip): projection GEMM whose epilogue does the GAT aggregation of whole row groups out of LDS --
// xp never reaches HBM.  Filled by the gat_seq driver.
struct FusedHopArgs {
    const int32_t* group_ptr;   // [G+1] first node of row group r (<= 128 nodes each, graph-aligned)
    int num_groups;
    const int32_t* row_order;   // NULL or [N]: slot s of group r is aggregated as local row row_order[group_ptr[r] + s]
    const int32_t* rowptr;      // CSR by destination
    const int32_t* csr_src;
    const float* alpha_csr;     // [E, H] attention coefficients in CSR slot order (k_gat_alpha_general)
    const int32_t* node_graph;
    const float* graph_term;    // NULL or [B, t_ld]: columns [0, C) head-mean instruction term
    int64_t t_ld;
    const float* bias;          // NULL or [C]
    const float* bn_w;          // all four NULL: no BatchNorm / ReLU
    const float* bn_b;
    const float* bn_m;
    const float* bn_v;
    const float* skip;          // NULL or [N, skip_ld]
    int64_t skip_ld;
    float* out;                 // [N, out_ld]
    int64_t out_ld;
    int H, C, cw;               // cw = 256 / H channels of every head per column block
    int e_cap;                  // LDS capacity in edges per row group
    float bn_eps;
    // chained hops on the 8-wave kernel (k_linear_split3<..., EPI = 2, ..., CHN = 1>; all NULL / 0 otherwise): the skip rows are read
    // back from the packed INPUT operand, the output leaves as the NEXT hop's packed operand (hop2.hip does the same)
    const uint16_t* ch_apk;     // the packed input rows (= the A operand), for the skip connection
    const float* ch_a_inv;      // [128 G] their inverse scales
    uint16_t* ch_pnext;         // NULL (last hop: fp32 rows to `out`) or the next hop's packed operand
    float* ch_a_inv_next;       // [128 G] its inverse scales (written here on the first hop, by the coefficient kernel afterwards)
    const float* ch_gscale;     // NULL (first hop: scales from a bound, decided in the kernel) or [B] per-graph output scales
    float* ch_pmout;            // [ncb][B] per-graph output maxima by column block
    const float* ch_tmax;       // NULL or [B] largest |instruction term| per graph
    const float* ch_bc;         // [4] bound constants of this hop (launch_hop2_bound_consts)
    const int32_t* ch_graph_ptr;
    int ch_B, ch_KB;
    // ... with the attention coefficients computed INSIDE the hop kernel (CHN = 2; ic_a_edge != NULL): no coefficient kernel, no alpha_csr.
    // The node logits of this hop arrive as `ic_parts_in` partial sets [part][N][2 H] (hop 0: one set, from the pack pass; later hops: one
    // per column block of the previous hop, whose epilogue left ic_lp_out = its rows . Vn_next over its own channels) and are summed
    // per row group in the kernel; the edge halves are gathered through csr_eid; leaky-relu + segment softmax per (node, head) in LDS.
    const int32_t* ic_csr_eid;  // [E] COO edge id of CSR slot s
    const float* ic_a_edge;     // edge halves of this hop's logits: ic_a_edge[eid * ic_a_edge_stride + h]
    int64_t ic_a_edge_stride;
    const float* ic_lp_in;      // [ic_parts_in][N][2 H] partial node logits of this hop
    int ic_parts_in;
    int64_t ic_lp_stride;       // floats between two partial sets (N * 2 H)
    float* ic_lp_out;           // NULL (last hop) or [ncb][N][2 H]: the NEXT hop's partial node logits, this column block's set
    const float* ic_vn_next;    // [2 H][C] folded attention vectors of the next hop (gat_skip.py:134-135 folded into the weights)
    const float* ic_pmin;       // NULL (first hop) or [ic_parts_in][B]: per-graph maxima of this hop's input rows by column block
    float* ic_alpha_out;        // NULL or [E, H] (COO order): the attention weights, written by column block 0
    float ic_slope;
    int pair_blocks;            // (set by launch_hop_fused_split) row blocks that take TWO row groups; the blocks behind them take one each (half tiles)
    int xcd_cols;               // column blocks per XCD of the workgroup -> tile map (1: plain launch order)
    int debug;                  // measurement aid (GVQA_FUSED_DEBUG bit mask): 1 no row image, 2 no aggregation, 4 no store, 8 no epilogue at all
};
// fp32-accurate projection from split operands on the 16-bit matrix cores (split3.hip); np = pieces per value:
// 3 = three bf16 pieces (exact split, six products), 2 = two scaled fp16 pieces (2^-22 split, three products)
size_t split_packed_rows_bytes(int np, int64_t row_tiles, int64_t K);
// Vn / J / a_node: NULL / 0 / NULL, or the folded attention vectors [J = 2 H, K]: a_node[N, J] = X . Vn^T is produced on the way
bool split_pack_groups_logits_supported(int np, int J, int64_t K);
// Vn_packed (two-piece form only): NULL or the packed image of Vn (launch_split_pack(2, J, K, Vn, ...)): logits on the matrix cores
int launch_split_pack_groups(int np, int num_groups, const int32_t* group_ptr, int64_t K, const float* X, int64_t ld, void* packed,
                             const float* Vn, int J, float* a_node, hipStream_t stream, const void* Vn_packed = nullptr);
int launch_split_pack_heads(int np, int H, int C, int cw, int64_t K, const float* W, int64_t ldw, void* packed, hipStream_t stream);
// cd != NULL (two-piece operands, H = 4): chained hop, as launch_hop2's
struct Hop2ChainDesc;
int launch_hop_fused_split(int np, int64_t K, const void* Apk, const void* Bpk, const FusedHopArgs& f, hipStream_t stream,
                           const Hop2ChainDesc* cd = nullptr);
size_t hop_fused_chain_lds_edge_capacity(int H);
size_t hop_fused_ic_lds_edge_capacity(int H);
// hop2.hip: the same hop as a persistent kernel, two 4-wave workgroups per CU (two-piece operands, half-interleaved weights)
int launch_split_pack_heads2(int H, int C, int cw, int64_t K, const float* W, int64_t ldw, void* packed, hipStream_t stream);
// cd != NULL: chained hop -- skip rows come out of the packed input; with cd->Pnext the output leaves as the next hop's packed
// operand (+ per-graph maxima, + the slots' inverse scales) instead of fp32 rows (hop2.hip)
struct Hop2ChainDesc {
    void* Pnext;              // packed output buffer (split_packed_rows_bytes(2, 4 G, C) bytes) or NULL: fp32 rows to FusedHopArgs::out
    float* PMout;             // [ncb][B]
    const float* PMin;        // [ncb][B] or NULL (first hop)
    const float* gscale;      // NULL or [B]: output scales per graph decided before the launch (launch_alpha_packed), a_inv_next written there
    const float* Tmax;        // [B] or NULL
    const float* bc;          // [4] launch_hop2_bound_consts of this hop
    const int32_t* graph_ptr; // [B + 1] device
    int B, N;
};
int launch_hop2(int64_t K, const void* Apk, const void* Bpk, const FusedHopArgs& f, const float* epc, const Hop2ChainDesc* cd, hipStream_t stream);
int launch_hop2_bound_consts(int H, int C, int Dn, const float* W, int64_t ldw, const float* epc, float* out, hipStream_t stream);
int launch_rows_absmax(int64_t rows, int C, const float* T, int64_t ld, float* out, hipStream_t stream);
// per-channel epilogue constants of a hop ([3][hop2_consts_ld]: bias | BatchNorm scale | shift) -- parameter-only, weight cache
int hop2_consts_ld(int H, int C);
int launch_hop2_consts(int H, int C, const float* bias, const float* bn_w, const float* bn_b, const float* bn_m, const float* bn_v,
                       float eps, float* out, hipStream_t stream);
size_t hop2_lds_edge_capacity(int H, bool chain);
size_t hop_fused_lds_edge_capacity(int H);

// hopagg.hip: the hop "aggregate first" (heads concatenated along K; H = 4): rows live chunk-major in HBM between hops
struct HopAggArgs {
    const int32_t *group_ptr, *rowptr, *csr_src, *node_graph;
    const float* alpha_csr;     // (unused since the coefficients are computed in the hop kernel's prologue; kept for layout stability of callers' memset + fill)
    const float* X4in;          // [G][NQ][128][4] input rows (also the skip rows)
    const uint16_t* Wk;         // [NCT][NQ][2][64][8] packed weights
    const float* binv;          // [32 NCT] inverse scales of the output columns
    const float* epc;           // [3][epc_ld] bias | BatchNorm scale | shift per output channel (k_hop2_consts)
    int epc_ld;
    const float* graph_term;    // NULL or [B, t_ld]: columns [0, C) per-graph instruction term (head mean)
    int64_t t_ld;
    const float* gmax_in;       // [B] largest |x| per graph of the input rows
    float* gmax_out;            // NULL or [B]: the same of the output rows
    float* X4out;               // NULL or [G][C / 4][128][4]: output rows, chunk-major (the next hop's input)
    float* out;                 // NULL or [N, out_ld]: output rows, row-major
    int64_t out_ld;
    int C, NQ, NCT, relu;
    const float* Vn_next;       // with a_node_out: [2 H][Dn] folded attention vectors of the NEXT hop
    float* a_node_out;          // NULL or [N, 2 H]: the next hop's node logits of the rows this launch produces
    // coefficients computed in the launch's prologue (instead of alpha_csr from a coefficient kernel):
    const float* a_node_in;     // NULL or [N, 2 H]: node logits of the input rows (the previous launch's a_node_out)
    const float* a_edge;        // [E, a_edge_stride] edge halves of the logits, COO order, this hop's H columns
    int64_t a_edge_stride;
    const int32_t* csr_eid;     // [E] CSR slot -> COO edge id
    float* alpha_out;           // NULL or [E, H]: the attention weights, COO order
    float slope;
    // column parts (k_hopagg4<..., CP > 1>: a row group's output columns split over CP workgroups, per-hop launches only -- small batches /
    // strong-scaling shards, where one workgroup per row group leaves most CUs idle): the next hop's node logits and the per-graph maxima
    // leave as one set per part (a_node_out / gmax_out + part x stride) and are combined by the reader (sum / max over the parts_in sets)
    int parts_in;               // sets of a_node_in / gmax_in to combine (1: the layout pass or a CP = 1 launch produced them)
    int64_t an_part_stride, gm_part_stride;      // floats between two sets (N x 2 H, B)
    int dbg;                    // measurement build only (GVQA_HOPAGG_DEBUG): 1 no producer in the loop, 2 no weight DMA, 4 no MFMAs, 8 no fragment reads after step 0, 16 no waits / barriers, 32 no epilogue
    // packed row groups (gvqa_graph::pk_*): group_ptr / rowptr / csr_src / csr_eid / node_graph above are then the PACKED arrays, and
    const int32_t* row_map;     // NULL or [N]: row of `out` (and of the caller's x) of packed node n
    const int32_t* graph_old;   // NULL or [B]: row of graph_term of packed graph index j
};
// The K hops of gat_seq as ONE launch of the aggregate-first kernel (k_hopagg4<..., SEQ>): per-hop operands.  The HopAggArgs beside it
// carry the batch (CSR, row groups), hop 0's node logits (a_node_in, from the layout pass), the per-graph maxima of the input rows
// (gmax_in), the final rows (out) and the shapes; Wk / binv / epc / graph_term / relu / X4in / X4out of HopAggArgs are not read.
// (per-hop operands as base + hop x stride: the weight cache and the workspace lay the hops out at constant strides, and a table
//  indexed by the hop would be copied from the kernel arguments to scratch)
constexpr int HA_MAXHOPS = 8;
struct HopAggSeq {
    const uint16_t* Wk;         // packed weights of hop 0; hop i at + i w_hop_bytes; their inverse column scales binv_off_bytes behind
    int64_t w_hop_bytes, binv_off_bytes;
    const float* epc;           // bias | BatchNorm scale | shift of hop 0; hop i at + i epc_hop floats
    int64_t epc_hop;
    const float* graph_term;    // NULL or [K][B, t_ld]: per-graph instruction term [0, C) and logit offsets [C, C + H); hop i at + i t_hop floats
    int64_t t_hop;
    const float* Vn;            // [K][2 H][Dn] folded attention vectors (hop i + 1's are read by hop i's coefficient phase)
    const int32_t* csr_eid;     // [E] CSR slot -> COO edge id
    const float* a_edge;        // [E, a_edge_stride] edge halves of the logits, COO order; hop i's H columns start at column i H
    int64_t a_edge_stride;
    float *X4a, *X4b;           // chunk-major row buffers: hop i reads (i odd ? X4b : X4a) and writes the other
    float slope;
    int K;
    unsigned relu_mask;         // bit i: hop i ends in BatchNorm + ReLU
    unsigned long long* stamps; // measurement build only: [groups][K][8] 100 MHz timestamps of wave 0 at the phase boundaries of every hop
};
bool hopagg_supported(int H, int C, int Dn, int max_row_group_edges);
int launch_hopagg_seq(int H, const HopAggArgs& a, const HopAggSeq& hs, int num_groups, hipStream_t stream);
size_t hopagg_packed_w_bytes(int C, int Dn, int H);
int launch_hopagg_pack_w(int H, int C, int Dn, const float* W, int64_t ldw, void* packed, hipStream_t stream);
int launch_rows_to_x4(const gvqa_graph* g, int D, const float* X, int64_t ld, float* X4, float* gmax, hipStream_t stream,
                      const float* Vn = nullptr, float* a_node = nullptr, bool packed = false);      // Vn: [8][D] folded vectors of hop 0 -> a_node [N, 8]; packed: the handle's packed row groups
int launch_hopagg(int H, const HopAggArgs& a, int num_groups, hipStream_t stream, int col_parts = 1);

size_t split_packed_bytes(int np, int64_t rows, int64_t K);
int launch_split_pack(int np, int64_t rows, int64_t K, const float* X, int64_t ld, void* packed, hipStream_t stream, float* absmax = nullptr,
                      const float* Vn = nullptr, int J = 0, float* a_node = nullptr);
int launch_split2h_pack_gather(int64_t rows, int64_t K, const float* Y, int64_t ld, const float* a, const int64_t* ia, int64_t lda,
                               const float* b, const int64_t* ib, int64_t ldb, const float* bias, void* packed, hipStream_t stream,
                               const int64_t* tok = nullptr, int V = 0, const uint8_t* neg = nullptr);   // tok: Y rows are +-Y[tok[r]] (a table)
int launch_split2h_pack_rowmul(int64_t rows, int64_t K, const float* X, int64_t ld, const float* R, const int32_t* idx, int64_t ldr, void* packed,
                               hipStream_t stream);
// C = X^T Y from row-major fp32 operands, split-K chunks of KC rows into S partial results (tn_direct.hip)
int launch_linear_tn_direct(int64_t R, int64_t M, int64_t N, const float* X, int64_t ldx, const float* Y, int64_t ldy, const float* xmax, int nxmax,
                            const float* ymax, int nymax, int KC, int S, float* C, int64_t ldc, int64_t zs_c, hipStream_t stream);
bool linear_tn_direct_applies(int KC, int64_t ldx, int64_t ldy);
// C [R, N] (+)= A [R, K] x packed B^T with A read as row-major fp32 (one scale from amax) (tn_direct.hip); K % 16 == 0
bool linear_nn_direct_applies(int64_t R, int64_t N, int64_t K, int64_t lda);
int launch_linear_nn_direct(int64_t R, int64_t N, int64_t K, const float* A, int64_t lda, const float* amax, int namax, const void* Bpk, int KBb, int TB,
                            const float* b_inv, float* C, int64_t ldc, int accumulate, hipStream_t stream, const float* lr_g = nullptr,
                            const float* lr_v = nullptr, int J = 0, const float* addend = nullptr, int64_t ld_add = 0);
// plain two-piece product from packed operands in the direct kernels' step layout (tn_direct.hip); measurement switch GVQA_PK_DIRECT
int launch_linear_pk_direct(int64_t M, int64_t N, int KB, const void* Apk, const float* a_inv, const void* Bpk, const float* b_inv, const LinearEpilogue& ep,
                            float* C, int64_t ldc, hipStream_t stream);
bool linear_split3_supported(int64_t N, const LinearEpilogue& ep, const float* C, int64_t ldc);
int launch_linear_split(int np, int64_t M, int64_t N, int64_t K, const void* Apk, const void* Bpk, LinearEpilogue ep, float* C,
                        int64_t ldc, hipStream_t stream, int batch = 1,
                        const float* a_inv_batched = nullptr, const float* b_inv_batched = nullptr);
// fp32 Linear with caller scratch: two-piece kernels when applicable, f32-input MFMA otherwise (split3.hip)
size_t linear_auto_scratch_bytes(int64_t M, int64_t N, int64_t K);
int launch_linear_auto(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* W, int64_t ldw, LinearEpilogue ep,
                       float* C, int64_t ldc, void* scratch, size_t scratch_bytes, hipStream_t stream);
const char* gemm_backend_name();
// gine_mlp.hip: GINEConv's Lin -> ReLU -> Lin as one kernel (C <= 320; the hidden rows stay in registers)
size_t gine_mlp_packed_bytes(int C, int Dn);
bool gine_mlp_supported(int64_t N, int C, int Dn, const float* z, int64_t ldz, const float* out, int64_t ldo, int64_t ldp);
int launch_gine_mlp(int64_t N, int C, int Dn, const float* z, int64_t ldz, const float* zmax, const float* W1, int64_t ld1, const float* b1,
                    const float* W2, int64_t ld2, const float* b2, const float* P1, const float* P2, int64_t ldp, const int32_t* node_graph,
                    const int32_t* rowptr, float eps, float* out, int64_t ldo, void* packed, hipStream_t stream);

}  // namespace gvqa
