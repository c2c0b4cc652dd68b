// LCGN variant of the execution module (gfx950): `lcgn_seq.forward` with its `gat_lcgn` layer.
//
// Reference being replaced: baseline_and_test_models/lcgn.py:303-323 (lcgn_seq.forward),
// :292-300 (extract_textual_command), :120-199 + :202-238 (gat_lcgn.forward / message).
//
// Restructuring (same math, fp32, re-associated sums):
//   * x_joint = [x_loc || x_ctx || proj_x_ctx * proj_x_loc] (lcgn.py:313) is never materialised:
//     the three projections lin_l / lin_r / cal_x are ONE stacked weight [3O, 3O] and the product
//     x_joint . W^T is computed as three K-segments accumulated in place; the x_loc segment does
//     not change across iterations and is computed once.
//   * cal_x is applied per NODE (N x 3O x O) instead of per edge on x_j (E x 3O x O, lcgn.py:230):
//     the same row-wise linear map, gathered afterwards.
//   * F.one_hot(batch).matmul(.) ([N, B] dense matmuls, lcgn.py:150-153) is a gather by graph id.
//   * messages cal_x(x_j) * cal_cmd_j * alpha summed over in-edges (lcgn.py:231-238): cal_cmd_j is the
//     source's graph = the destination's graph, so it factors out of the sum as a per-graph channel
//     scale; the aggregation itself is the GAT message-passing kernel with H heads = 1, fed
//     with the dot-product logits as its per-edge logit term (leaky-relu + softmax happen inside).
#include "common.h"

namespace gvqa {

int launch_gat_mp_public(const gvqa_graph* g, const gvqa_gat_mp_desc* d, void* ws, size_t ws_bytes, hipStream_t stream);

// cmd[t, b, :] = sum_l softmax_l( (q_cmd[b, t] * lstm[l, b]) . w + bias ) * lstm[l, b, :]     (lcgn.py:292-300)
// for every iteration t at once (the commands do not depend on the node state).  q_cmd is [B, T*O]
// (iteration t at columns [t*O, (t+1)*O)), cmd is [T*B, O].  One block per (graph b, iteration t);
// dynamic LDS: L floats.
__global__ __launch_bounds__(256) void k_lcgn_command(int L, int B, int O, const float* __restrict__ q_cmd,
                                                      const float* __restrict__ lstm, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ cmd) {
    extern __shared__ float att[];
    const int b = blockIdx.x, t = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    q_cmd += ((int64_t)b * gridDim.y + t) * O - (int64_t)b * O;
    cmd += (int64_t)t * B * O;
    for (int l = wave; l < L; l += 4) {
        const float* row = lstm + ((int64_t)l * B + b) * O;
        float s = 0.f;
        for (int d = lane; d < O; d += 64) s += (q_cmd[(int64_t)b * O + d] * row[d]) * w[d];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) att[l] = s + bias[0];
    }
    __syncthreads();
    float m = -INFINITY;
    for (int l = 0; l < L; ++l) m = fmaxf(m, att[l]);
    float den = 0.f;
    for (int l = 0; l < L; ++l) den += expf(att[l] - m);
    for (int d = threadIdx.x; d < O; d += 256) {
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc += (expf(att[l] - m) / den) * lstm[((int64_t)l * B + b) * O + d];
        cmd[(int64_t)b * O + d] = acc;
    }
}

// logit[eid] = sum_c x_l[src, c] * (proj_cmd[g(dst), c] * x_r[dst, c])      (lcgn.py:154,207), H = 1.
// One wave per destination node; its y = proj_cmd * x_r row stays in registers.
constexpr int LCGN_MAXC_PER_LANE = 16;      // C <= 1024
template <bool T16>      // T16: x_l / x_r rows stored as bf16
__global__ __launch_bounds__(256) void k_lcgn_edge_logit(int N, int C, const void* __restrict__ xl, int64_t xl_ld,
                                                         const void* __restrict__ xr, int64_t xr_ld,
                                                         const float* __restrict__ proj_cmd, int64_t pc_ld,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ csr_src,
                                                         const int32_t* __restrict__ csr_eid,
                                                         const int32_t* __restrict__ node_graph, float* __restrict__ logit) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const int g = node_graph[i];
    float y[LCGN_MAXC_PER_LANE];
#pragma unroll
    for (int k = 0; k < LCGN_MAXC_PER_LANE; ++k) {
        const int c = lane + k * 64;
        y[k] = c < C ? proj_cmd[(int64_t)g * pc_ld + c] * load_elem<T16>(xr, (int64_t)i * xr_ld + c) : 0.f;
    }
    for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
        const int64_t row = (int64_t)csr_src[s] * xl_ld;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < LCGN_MAXC_PER_LANE; ++k) {
            const int c = lane + k * 64;
            if (c < C) acc += load_elem<T16>(xl, row + c) * y[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) logit[csr_eid[s]] = acc;
    }
}

// ---- vectorised variants (C % 8 == 0, 16-byte aligned rows): a lane owns 8 consecutive channels ------
template <bool H16>
__device__ __forceinline__ void load8(const void* base, int64_t elem, float (&v)[8]) {
    if (H16) {
        const uint4 r = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(base) + elem);
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xFFFF0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xFFFF0000u);
        v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xFFFF0000u);
        v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xFFFF0000u);
    } else {
        const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem);
        const float4 b = *reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// k_lcgn_edge_logit with 16/32-byte row segments per lane and two edges in flight.  KC = ceil(C / 512).
template <bool T16, int KC>
__global__ __launch_bounds__(256) void k_lcgn_edge_logit_v(int N, int C, const void* __restrict__ xl, int64_t xl_ld,
                                                           const void* __restrict__ xr, int64_t xr_ld,
                                                           const float* __restrict__ proj_cmd, int64_t pc_ld,
                                                           const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ csr_src,
                                                           const int32_t* __restrict__ csr_eid,
                                                           const int32_t* __restrict__ node_graph, float* __restrict__ logit) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (i >= N) return;
    const int g = node_graph[i];
    float y[KC][8];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = (lane + k * 64) * 8;
        if (c < C) {
            float a[8], b[8];
            load8<false>(proj_cmd, (int64_t)g * pc_ld + c, a);
            load8<T16>(xr, (int64_t)i * xr_ld + c, b);
#pragma unroll
            for (int q = 0; q < 8; ++q) y[k][q] = a[q] * b[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) y[k][q] = 0.f;
        }
    }
    auto dot_row = [&](int src) -> float {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = (lane + k * 64) * 8;
            if (c < C) {
                float v[8];
                load8<T16>(xl, (int64_t)src * xl_ld + c, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += v[q] * y[k][q];
            }
        }
        return acc;
    };
    const int lo = rowptr[i], hi = rowptr[i + 1];
    int s = lo;
    for (; s + 1 < hi; s += 2) {
        float a0 = dot_row(csr_src[s]), a1 = dot_row(csr_src[s + 1]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
        if (lane == 0) { logit[csr_eid[s]] = a0; logit[csr_eid[s + 1]] = a1; }
    }
    if (s < hi) {
        const float a0 = wave_sum(dot_row(csr_src[s]));
        if (lane == 0) logit[csr_eid[s]] = a0;
    }
}

// k_lcgn_aggregate_bf16 with the softmax weights computed once per edge (lanes over edges) and 16-byte
// row segments per lane in the weighted gather.
template <int KC>
__global__ __launch_bounds__(256) void k_lcgn_aggregate_bf16_v(int N, int C, const uint16_t* __restrict__ xval, int64_t xv_ld,
                                                               const float* __restrict__ logit, const float* __restrict__ cal_cmd,
                                                               int64_t cc_ld, const float* __restrict__ bias, float slope,
                                                               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_src,
                                                               const int32_t* __restrict__ csr_eid,
                                                               const int32_t* __restrict__ node_graph, uint16_t* __restrict__ msg,
                                                               int64_t msg_ld) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (i >= N) return;
    const int lo = rowptr[i], hi = rowptr[i + 1], g = node_graph[i];
    auto edge_logit = [&](int s) { float v = logit[csr_eid[s]]; return v > 0.f ? v : v * slope; };
    float m = -INFINITY;
    for (int s = lo + lane; s < hi; s += 64) m = fmaxf(m, edge_logit(s));
    m = wave_max(m);
    float den = 0.f;
    for (int s = lo + lane; s < hi; s += 64) den += expf(edge_logit(s) - m);
    den = wave_sum(den) + 1e-16f;
    float acc[KC][8];
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[k][q] = 0.f;
    for (int base = lo; base < hi; base += 64) {
        const int s = base + lane, cnt = min(64, hi - base);
        const float alpha = s < hi ? expf(edge_logit(s) - m) / den : 0.f;
        const int src = s < hi ? csr_src[s] : 0;
        for (int j = 0; j < cnt; ++j) {
            const float a = __shfl(alpha, j, 64);
            const int sj = __shfl(src, j, 64);
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int c = (lane + k * 64) * 8;
                if (c < C) {
                    float v[8];
                    load8<true>(xval, (int64_t)sj * xv_ld + c, v);
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[k][q] += a * v[q];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = (lane + k * 64) * 8;
        if (c < C) {
            float cc[8];
            load8<false>(cal_cmd, (int64_t)g * cc_ld + c, cc);
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float r0 = acc[k][2 * q] * cc[2 * q], r1 = acc[k][2 * q + 1] * cc[2 * q + 1];
                if (bias) { r0 += bias[c + 2 * q]; r1 += bias[c + 2 * q + 1]; }
                w[q] = (unsigned)f32_to_bf16(r0) | ((unsigned)f32_to_bf16(r1) << 16);
            }
            *reinterpret_cast<uint4*>(msg + (int64_t)i * msg_ld + c) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// out[r, c] = in[r, c] (c < cols), 0 (cols <= c < cols_out): a [rows, cols] fp32 block into a wider row
// (fp32 or bf16 storage), optionally zero-padded
template <bool OUT16>
__global__ __launch_bounds__(256) void k_place_rows(int64_t rows, int cols, int cols_out, const float* __restrict__ in,
                                                    void* __restrict__ out, int64_t out_ld) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols_out) return;
    const int64_t r = i / cols_out;
    const int c = (int)(i - r * cols_out);
    const float v = c < cols ? in[r * cols + c] : 0.f;
    if (OUT16) static_cast<uint16_t*>(out)[r * out_ld + c] = f32_to_bf16(v);
    else static_cast<float*>(out)[r * out_ld + c] = v;
}

// bf16-node-feature mode: msg[i] = (sum_{e -> i} alpha_e * x_val[src(e)]) * cal_cmd[g(i)] + bias with
// alpha = softmax over the in-edges of leaky_relu(logit) (lcgn.py:209-238, H = 1), x_val rows and msg
// stored as bf16, everything accumulated in fp32.  One wave per destination node; the (short) edge
// list is scanned three times by every lane (max, sum, weighted gather).
__global__ __launch_bounds__(256) void k_lcgn_aggregate_bf16(int N, int C, const uint16_t* __restrict__ xval, int64_t xv_ld,
                                                             const float* __restrict__ logit, const float* __restrict__ cal_cmd,
                                                             int64_t cc_ld, const float* __restrict__ bias, float slope,
                                                             const int32_t* __restrict__ rowptr, const int32_t* __restrict__ csr_src,
                                                             const int32_t* __restrict__ csr_eid,
                                                             const int32_t* __restrict__ node_graph, uint16_t* __restrict__ msg,
                                                             int64_t msg_ld) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const int lo = rowptr[i], hi = rowptr[i + 1], g = node_graph[i];
    float m = -INFINITY;
    for (int s = lo; s < hi; ++s) { float v = logit[csr_eid[s]]; v = v > 0.f ? v : v * slope; m = fmaxf(m, v); }
    float den = 0.f;
    for (int s = lo; s < hi; ++s) { float v = logit[csr_eid[s]]; v = v > 0.f ? v : v * slope; den += expf(v - m); }
    den += 1e-16f;
    for (int c = lane; c < C; c += 64) {
        float acc = 0.f;
        for (int s = lo; s < hi; ++s) {
            float v = logit[csr_eid[s]];
            v = v > 0.f ? v : v * slope;
            acc += (expf(v - m) / den) * bf16_to_f32(xval[(int64_t)csr_src[s] * xv_ld + c]);
        }
        float r = acc * cal_cmd[(int64_t)g * cc_ld + c];
        if (bias) r += bias[c];
        msg[(int64_t)i * msg_ld + c] = f32_to_bf16(r);
    }
}

// Call-invariant weight forms (stacked fp32 matrices + the bf16 pieces of the node-GEMM weights): built by
// lcgn_pack into a caller-held blob (gvqa_lcgn_pack_weights) or, per call, into the workspace.
struct LcgnPack {
    size_t Wx, Wj, Wpc, Wq, bq, Wpk, W2h, total;
    size_t w2h[8];          // byte offsets of the eight node-GEMM weights' two-piece images inside W2h (fp32 mode)
};
// the eight node-GEMM weights in one table: rows, K (order: proj_x_loc, Wx, Wj, proj_x_ctx, output, fin[:, :O], fin[:, O:], init)
static void lcgn_node_weight_shapes(const gvqa_lcgn_dims* d, int64_t (&rows)[8], int64_t (&K)[8]) {
    const int64_t O = d->out_channels, Cin = d->in_channels;
    const int64_t r[8] = {O, 3 * O, 3 * O, O, O, O, O, O}, k[8] = {O, O, 2 * O, O, 2 * O, O, O, Cin};
    for (int i = 0; i < 8; ++i) { rows[i] = r[i]; K[i] = k[i]; }
}
static LcgnPack lcgn_pack_layout(const gvqa_lcgn_dims* d) {
    LcgnPack L; size_t off = 0;
    auto take = [&](size_t n) { size_t r = off; off += align_up(n * sizeof(float), 256); return r; };
    const size_t O = d->out_channels, T = d->num_iters, K8 = align_up(d->in_channels, 8);
    L.Wx = take(3 * O * O); L.Wj = take(3 * O * 2 * O); L.Wpc = take(2 * O * O); L.Wq = take(T * O * O); L.bq = take(T * O);
    // node-GEMM weights as bf16 pieces: (15 O^2 + O K8) elements x <= 2 pieces x 2 bytes
    L.Wpk = take(d->node_bf16 ? 15 * O * O + O * K8 : 0);
    // fp32 mode: two-piece fp16 images of the same weights (split2h projection arithmetic, split3.hip)
    int64_t rows[8], Kk[8];
    lcgn_node_weight_shapes(d, rows, Kk);
    size_t w2 = 0;
    for (int i = 0; i < 8; ++i) { L.w2h[i] = w2; w2 += d->node_bf16 ? 0 : align_up(split_packed_bytes(2, rows[i], Kk[i]), 256); }
    L.W2h = take(w2 / sizeof(float));
    L.total = off;
    return L;
}
struct PackedWeights {      // bf16 piece matrices, [rows, PW * K] each
    uint16_t *pxl, *Wx, *Wj, *pxc, *out, *f1, *f2, *init;
};
static PackedWeights packed_weights(char* blob, const LcgnPack& L, const gvqa_lcgn_dims* d) {
    uint16_t* pk = reinterpret_cast<uint16_t*>(blob + L.Wpk);
    const size_t OO = (size_t)d->out_channels * d->out_channels * (d->node_bf16 == 2 ? 1 : 2);
    return {pk, pk + OO, pk + 4 * OO, pk + 10 * OO, pk + 11 * OO, pk + 13 * OO, pk + 14 * OO, pk + 15 * OO};
}
static int lcgn_pack(const gvqa_lcgn_dims* d, const gvqa_lcgn_params* p, char* blob, hipStream_t stream) {
    const int O = d->out_channels, T = d->num_iters, Cin = d->in_channels;
    const LcgnPack PL = lcgn_pack_layout(d);
    auto PB = [&](size_t off) { return reinterpret_cast<float*>(blob + off); };
    GVQA_REQUIRE(p->lin_l_weight && p->lin_r_weight && p->cal_x_weight && p->proj_cmd_weight && p->cal_cmd_weight &&
                 p->proj_x_loc_weight && p->proj_x_ctx_weight && p->output_weight && p->fin_weight && p->init_weight,
                 GVQA_E_INVALID, "lcgn_seq: null weight");
    // stacked weights over the rows [lin_l; lin_r; cal_x] (each [O, 3O], input = [x_loc | x_ctx | prod]):
    //   Wx = columns of x_loc ([3O, O]);  Wj = columns of [prod | x_ctx] in the order of the XC rows ([3O, 2O]);
    //   Wpc = [proj_cmd; cal_cmd] ([2O, O]);  Wq = [qInput2[0]; ...; qInput2[T-1]] ([T*O, O]), bq likewise
    const float* wsrc[3] = {p->lin_l_weight, p->lin_r_weight, p->cal_x_weight};
    const size_t fo = (size_t)O * sizeof(float);
    for (int m = 0; m < 3; ++m) {
        GVQA_HIP_CHECK(hipMemcpy2DAsync(PB(PL.Wx) + (size_t)m * O * O, fo, wsrc[m], 3 * fo, fo, O, hipMemcpyDeviceToDevice, stream));
        GVQA_HIP_CHECK(hipMemcpy2DAsync(PB(PL.Wj) + (size_t)m * O * 2 * O, 2 * fo, wsrc[m] + 2 * O, 3 * fo, fo, O,
                                        hipMemcpyDeviceToDevice, stream));
        GVQA_HIP_CHECK(hipMemcpy2DAsync(PB(PL.Wj) + (size_t)m * O * 2 * O + O, 2 * fo, wsrc[m] + O, 3 * fo, fo, O,
                                        hipMemcpyDeviceToDevice, stream));
    }
    GVQA_HIP_CHECK(hipMemcpyAsync(PB(PL.Wpc), p->proj_cmd_weight, (size_t)O * O * 4, hipMemcpyDeviceToDevice, stream));
    GVQA_HIP_CHECK(hipMemcpyAsync(PB(PL.Wpc) + (size_t)O * O, p->cal_cmd_weight, (size_t)O * O * 4, hipMemcpyDeviceToDevice, stream));
    for (int t = 0; t < T; ++t) {
        GVQA_REQUIRE(p->qinput2_weight[t] && p->qinput2_bias[t], GVQA_E_INVALID, "lcgn_seq: qinput2[%d] is null", t);
        GVQA_HIP_CHECK(hipMemcpyAsync(PB(PL.Wq) + (size_t)t * O * O, p->qinput2_weight[t], (size_t)O * O * 4, hipMemcpyDeviceToDevice, stream));
        GVQA_HIP_CHECK(hipMemcpyAsync(PB(PL.bq) + (size_t)t * O, p->qinput2_bias[t], fo, hipMemcpyDeviceToDevice, stream));
    }
    if (!d->node_bf16) {     // two-piece images for the fp32 mode's node GEMMs
        const float* src[8] = {p->proj_x_loc_weight, PB(PL.Wx), PB(PL.Wj), p->proj_x_ctx_weight, p->output_weight, p->fin_weight,
                               p->fin_weight + O, p->init_weight};
        const int64_t ldw[8] = {O, O, 2 * O, O, 2 * O, 2 * O, 2 * O, Cin};
        int64_t rows[8], Kk[8];
        lcgn_node_weight_shapes(d, rows, Kk);
        for (int i = 0; i < 8; ++i) {
            int rc = launch_split_pack(2, rows[i], Kk[i], src[i], ldw[i], blob + PL.W2h + PL.w2h[i], stream);
            if (rc) return rc;
        }
    }
    if (d->node_bf16 && O % 8 == 0) {
        const int PW = d->node_bf16 == 2 ? 1 : 2, K8 = (int)align_up(Cin, 8);
        const PackedWeights w = packed_weights(blob, PL, d);
        struct { const float* src; int64_t ldw; int rows, K, Kp; uint16_t* out; } pw[8] = {
            {p->proj_x_loc_weight, O, O, O, O, w.pxl}, {PB(PL.Wx), O, 3 * O, O, O, w.Wx}, {PB(PL.Wj), 2 * O, 3 * O, 2 * O, 2 * O, w.Wj},
            {p->proj_x_ctx_weight, O, O, O, O, w.pxc}, {p->output_weight, 2 * O, O, 2 * O, 2 * O, w.out},
            {p->fin_weight, 2 * O, O, O, O, w.f1}, {p->fin_weight + O, 2 * O, O, O, O, w.f2},
            {p->init_weight, Cin, O, Cin, K8, w.init}};       // K zero-padded to a multiple of 8
        for (auto& e : pw) {
            int rc = launch_pack_weight_bf16(e.rows, e.K, e.Kp, PW, e.src, e.ldw, e.out, stream);
            if (rc) return rc;
        }
    }
    return GVQA_OK;
}

struct LcgnLayout {
    size_t x_loc, proj_x_loc, q_emb, q_cmd, cmd, pc, XC0, XC1, XL, J, logit, x16, pack, alpha, apk, pk5, pk_one, total;
};
static LcgnLayout lcgn_layout(int64_t N, int64_t E, int64_t B, const gvqa_lcgn_dims* d) {
    LcgnLayout L; size_t off = 0;
    auto take = [&](size_t n) { size_t r = off; off += align_up(n * sizeof(float), 256); return r; };
    const size_t O = d->out_channels;
    const size_t T = d->num_iters;
    L.x_loc = take(N * O); L.proj_x_loc = take(N * O); L.q_emb = take(B * O); L.q_cmd = take(B * T * O);
    L.cmd = take(T * B * O); L.pc = take(T * B * 2 * O);
    // per-node iteration state, two copies (the output layer reads one and writes the other):
    // row = [ prod | x_ctx | msg ], so that [prod | x_ctx] and [x_ctx | msg] are contiguous K = 2O operands
    L.XC0 = take(N * 3 * O); L.XC1 = take(N * 3 * O);
    L.XL = take(N * 3 * O); L.J = take(N * 3 * O); L.logit = take(E);
    L.x16 = take(d->node_bf16 ? (N * align_up(d->in_channels, 8) + 1) / 2 : 0);     // x as bf16, K padded to 8
    L.pack = take(lcgn_pack_layout(d).total / sizeof(float));                        // used when params->packed is NULL
    L.alpha = take(E);
    // fp32 mode: the two-piece image of a node GEMM's A operand (largest: K = 2 O or in_channels)
    L.apk = take(d->node_bf16 ? 0 : split_packed_bytes(2, N, std::max<int64_t>(2 * (int64_t)d->out_channels, d->in_channels)) / sizeof(float));
    // fp32 mode, chained products: x_loc | x_ctx (two buffers) | prod | msg as packed two-piece images [N, O] (written by the producing
    // products' epilogues; msg by one pack pass)
    L.pk_one = d->node_bf16 ? 0 : align_up(split_packed_bytes(2, N, d->out_channels), 256);
    L.pk5 = take(5 * L.pk_one / sizeof(float));
    L.total = off;
    return L;
}

}  // namespace gvqa

extern "C" {
using namespace gvqa;

size_t gvqa_lcgn_pack_bytes(const gvqa_lcgn_dims* d) {
    if (!d || d->num_iters < 1 || d->num_iters > 8 || d->out_channels < 1 || d->in_channels < 1) return 0;
    return lcgn_pack_layout(d).total;
}

int gvqa_lcgn_pack_weights(const gvqa_lcgn_dims* d, const gvqa_lcgn_params* p, void* packed, size_t packed_bytes, void* stream) {
    GVQA_REQUIRE(d && p && packed, GVQA_E_INVALID, "lcgn_pack_weights: null argument");
    GVQA_REQUIRE(d->out_channels > 0 && d->in_channels > 0 && d->num_iters >= 1 && d->num_iters <= 8 && d->node_bf16 >= 0 &&
                 d->node_bf16 <= 2, GVQA_E_INVALID, "lcgn_pack_weights: bad dims");
    GVQA_REQUIRE(packed_bytes >= lcgn_pack_layout(d).total, GVQA_E_WORKSPACE, "lcgn_pack_weights: buffer %zu < required %zu",
                 packed_bytes, lcgn_pack_layout(d).total);
    return lcgn_pack(d, p, static_cast<char*>(packed), static_cast<hipStream_t>(stream));
}

size_t gvqa_lcgn_seq_workspace_bytes(const gvqa_graph* g, const gvqa_lcgn_dims* d) {
    if (!g || !d || d->num_iters < 1 || d->num_iters > 8 || d->out_channels < 1 || d->in_channels < 1) return 0;
    return lcgn_layout(g->num_nodes, g->num_edges, g->num_graphs, d).total;
}

int gvqa_lcgn_seq_forward(const gvqa_graph* g, const gvqa_lcgn_dims* d, const gvqa_lcgn_params* p, const float* x,
                          const float* q_encoding, const float* lstm_outputs, const float* x_ctx_init, float* out,
                          void* ws, size_t ws_bytes, void* stream_) {
    GVQA_REQUIRE(g && d && p, GVQA_E_INVALID, "lcgn_seq: null argument");
    const int O = d->out_channels, Cin = d->in_channels, Q = d->question_dim, T = d->num_iters, Lq = d->seq_len;
    GVQA_REQUIRE(O > 0 && Cin > 0 && Q > 0 && Lq > 0 && T >= 1 && T <= 8, GVQA_E_INVALID, "lcgn_seq: bad dims");
    GVQA_REQUIRE(d->heads == 1, GVQA_E_UNSUPPORTED, "lcgn_seq: gat_heads != 1 is not implemented (reference default 1)");
    GVQA_REQUIRE(O <= 64 * LCGN_MAXC_PER_LANE, GVQA_E_UNSUPPORTED, "lcgn_seq: out_channels > %d", 64 * LCGN_MAXC_PER_LANE);
    GVQA_REQUIRE(g->finalized && g->intra_graph, GVQA_E_UNSUPPORTED, "lcgn_seq: needs a finalized intra-graph batch");
    const int64_t N = g->num_nodes, E = g->num_edges, B = g->num_graphs;
    LcgnLayout L = lcgn_layout(N, E, B, d);
    GVQA_REQUIRE(ws && ws_bytes >= L.total, GVQA_E_WORKSPACE, "lcgn_seq: workspace %zu < required %zu", ws_bytes, L.total);
    if (N == 0) return GVQA_OK;
    GVQA_REQUIRE(x && q_encoding && lstm_outputs && x_ctx_init && out, GVQA_E_INVALID, "lcgn_seq: null tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* base = static_cast<char*>(ws);
    auto P = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    int rc;
    // node tensors (x_loc, proj_x_loc, x_ctx, prod, XL, J, msg) are fp32, or bf16 in the bf16-node-feature
    // mode (BASELINE config 5): same buffers, half the bytes.  In that mode the node GEMMs run on the bf16
    // matrix cores against the weights split into two bf16 pieces (node_bf16 = 1: weights keep 16
    // significant bits) or rounded to one (node_bf16 = 2), fp32 accumulation (gemm_bf16.hip); everything
    // else (logits, softmax, aggregation, per-graph command path) is fp32 arithmetic.  Channel counts that
    // are not multiples of 8 fall back to the f32-MFMA kernel reading the bf16 storage.
    const bool nb = d->node_bf16 != 0;
    GVQA_REQUIRE(d->node_bf16 >= 0 && d->node_bf16 <= 2, GVQA_E_INVALID, "lcgn_seq: node_bf16 must be 0, 1 or 2");
    const int FA = nb ? 1 : 0, FC = nb ? 2 : 0;      // dtype flags: A is a node tensor / C is a node tensor
    const int PW = d->node_bf16 == 2 ? 1 : 2;        // bf16 pieces per weight
    const bool mf16 = nb && O % 8 == 0;
    const LcgnPack PL = lcgn_pack_layout(d);
    GVQA_REQUIRE(!p->packed || p->packed_bytes >= PL.total, GVQA_E_INVALID, "lcgn_seq: packed weights %zu < required %zu bytes",
                 (size_t)p->packed_bytes, PL.total);
    char* const blob = p->packed ? static_cast<char*>(const_cast<void*>(p->packed)) : base + L.pack;
    const PackedWeights pkw = packed_weights(blob, PL, d);
    auto PB = [&](size_t off) { return reinterpret_cast<float*>(blob + off); };
    // node GEMM: A is a node tensor (bf16 in nb mode); c_node: C (and addend / mul) are node tensors too
    // fp32 mode: node GEMMs on the split2h arithmetic (GVQA_OPT_PROJECTION != f32): A packed per product, weights from the blob
    const bool split_nodes = !nb && get_option(GVQA_OPT_PROJECTION) != GVQA_PROJECTION_F32 && O % 4 == 0;
    auto NODE = [&](int64_t M_, int64_t N_, int64_t K_, const float* A_, int64_t lda_, const float* W_, int64_t ldw_,
                    const uint16_t* Wpk_, LinearEpilogue ep_, float* C_, int64_t ldc_, bool c_node, int widx) -> int {
        // (stages: the node products under "proj", their operand pack passes under "pack" -- VERDICT r04 #5: the forward was one "other")
        if (mf16 && linear_bf16_supported(K_, lda_, A_, Wpk_)) {
            StageTimer tp(GVQA_STAGE_PROJ, stream);
            return launch_linear_bf16(M_, N_, K_, PW, A_, lda_, Wpk_, ep_, C_, ldc_, c_node, stream);
        }
        if (split_nodes && linear_split3_supported(N_, ep_, C_, ldc_) &&
            2.0 * (double)M_ * (double)N_ * (double)K_ >= 1e6 * (double)get_option(GVQA_OPT_SPLIT3_MIN_MFLOP)) {
            {
                StageTimer tk(GVQA_STAGE_PACK, stream);
                int rc2 = launch_split_pack(2, M_, K_, A_, lda_, base + L.apk, stream);
                if (rc2) return rc2;
            }
            StageTimer tp(GVQA_STAGE_PROJ, stream);
            return launch_linear_split(2, M_, N_, K_, base + L.apk, blob + PL.W2h + PL.w2h[widx], ep_, C_, ldc_, stream);
        }
        StageTimer tp(GVQA_STAGE_PROJ, stream);
        return launch_linear_t(M_, N_, K_, A_, lda_, W_, ldw_, ep_, C_, ldc_, 1, 0, 0, 0, FA | (c_node ? FC : 0), stream);
    };
#define NODE_LIN(...) do { rc = NODE(__VA_ARGS__); if (rc) return rc; } while (0)
#define LINT(M_, N_, K_, A_, lda_, W_, ldw_, ep_, C_, ldc_, fl_)                                                     \
    do { rc = launch_linear_t(M_, N_, K_, A_, lda_, W_, ldw_, ep_, C_, ldc_, 1, 0, 0, 0, fl_, stream); if (rc) return rc; } while (0)
#define LIN(M_, N_, K_, A_, lda_, W_, ldw_, bias_, relu_, C_, ldc_)                                                  \
    do { LinearEpilogue e_{bias_, nullptr, 0, nullptr, 0, relu_}; LINT(M_, N_, K_, A_, lda_, W_, ldw_, e_, C_, ldc_, 0); } while (0)
    // element offset into a node tensor (bf16 pointers travel as float*: offsets are halved)
    auto NP = [&](float* base_, int64_t elems) -> float* {
        return nb ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(base_) + elems) : base_ + elems;
    };

    if (!p->packed) { StageTimer tw(GVQA_STAGE_FOLD, stream); rc = lcgn_pack(d, p, blob, stream); if (rc) return rc; }
    // ---- fp32 mode, CHAINED node products (round 5): x_loc, x_ctx and prod exist only as packed two-piece operands, written by the
    // epilogue of the product that makes them (k_linear_split3<..., EPI = 3>); the two K = 2 O products take [prod | x_ctx] and
    // [x_ctx | msg] as two K segments with their own row scales (LinearEpilogue::a2).  Pack passes per forward: x, the initial x_ctx and
    // msg once per iteration (2 + T) instead of 5 + 3 T; no fp32 x_loc / x_ctx / prod rows at all.
    const bool chained = split_nodes && O % 32 == 0 && O <= 512 && Cin % 4 == 0 && N <= 65535ll * 128 &&
                         2.0 * (double)N * (double)O * (double)O >= 1e6 * (double)get_option(GVQA_OPT_SPLIT3_MIN_MFLOP);
    static const bool overlap_off = []() { const char* v = getenv("GVQA_LCGN_OVERLAP"); return v && v[0] == '0'; }();      // (A/B switch)
    SideStream* ss = (overlap_off || T < 1) ? nullptr : side_stream_get();
    if (chained) {
        char* pkb = base + L.pk5;
        char *xloc_pk = pkb, *xc_pk[2] = {pkb + L.pk_one, pkb + 2 * L.pk_one}, *prod_pk = pkb + 3 * L.pk_one, *msg_pk = pkb + 4 * L.pk_one;
        const int KBO = O / 16;
        const size_t inv_off = (size_t)cdiv(N, 32) * KBO * 2048;                 // inverse row scales behind an [N, O] image
        auto W2 = [&](int widx) { return blob + PL.W2h + PL.w2h[widx]; };
        auto PACK = [&](const float* X_, int64_t K_, int64_t ld_, char* dst_) -> int {
            StageTimer tk(GVQA_STAGE_PACK, stream);
            return launch_split_pack(2, N, K_, X_, ld_, dst_, stream);
        };
        // one product: A (packed, K1 columns) [+ second segment], weights image widx, epilogue, fp32 rows to C_ and / or packed rows to pk_
        auto PROD = [&](int64_t N_, int64_t K1_, const char* A_, const char* A2_, int widx, LinearEpilogue ep_, float* C_, int64_t ldc_, char* pk_) -> int {
            StageTimer tp(GVQA_STAGE_PROJ, stream);
            int64_t K_ = K1_;
            if (A2_) {
                ep_.a2 = reinterpret_cast<const uint16_t*>(A2_);
                ep_.a2_inv = reinterpret_cast<const float*>(A2_ + inv_off);
                ep_.a2_kb0 = (int)(K1_ / 16); ep_.a2_KB = KBO;
                K_ = K1_ + O;
            }
            ep_.pk_out = reinterpret_cast<uint16_t*>(pk_);
            return launch_linear_split(2, N, N_, K_, A_, W2(widx), ep_, C_, ldc_, stream);
        };
#define CH(call_) do { rc = (call_); if (rc) return rc; } while (0)
        CH(PACK(x, Cin, Cin, base + L.apk));                                                                    // x_loc = init(x)            :305
        CH(PROD(O, Cin, base + L.apk, nullptr, 7, LinearEpilogue{p->init_bias, nullptr, 0, nullptr, 0, 0}, nullptr, O, xloc_pk));
        {   // the per-question command chain (all T iterations): "graph_term"                                   :307,292-300,148-149
            // (four small dependent launches on [B, O]-sized operands -- 0.14 ms of pure latency at config 5 -- that nothing needs before the
            //  first iteration's edge logits: on the side stream, beside the node products below; joined in front of those logits)
            if (ss) CH(side_fork(ss, stream));
            hipStream_t stream_main = stream;
            (void)stream_main;
            hipStream_t stream = ss ? ss->stream : stream_main;           // (the LIN macro launches on `stream`)
            StageTimer tq(GVQA_STAGE_GRAPH_TERM, stream);
            LIN(B, O, Q, q_encoding, Q, p->qinput1_weight, Q, p->qinput1_bias, 1, P(L.q_emb), O);
            LIN(B, T * O, O, P(L.q_emb), O, PB(PL.Wq), O, PB(PL.bq), 0, P(L.q_cmd), (int64_t)T * O);
            hipLaunchKernelGGL(k_lcgn_command, dim3((unsigned)B, (unsigned)T), dim3(256), (size_t)Lq * sizeof(float), stream, Lq, (int)B,
                               O, P(L.q_cmd), lstm_outputs, p->cmd_logit_weight, p->cmd_logit_bias, P(L.cmd));
            GVQA_LAUNCH_CHECK();
            LIN((int64_t)T * B, 2 * O, O, P(L.cmd), O, PB(PL.Wpc), O, nullptr, 0, P(L.pc), 2 * O);
        }
        CH(PROD(O, O, xloc_pk, nullptr, 0, LinearEpilogue{p->proj_x_loc_bias, nullptr, 0, nullptr, 0, 0}, P(L.proj_x_loc), O, nullptr));      // :308
        CH(PROD(3 * O, O, xloc_pk, nullptr, 1, LinearEpilogue{nullptr, nullptr, 0, nullptr, 0, 0}, P(L.XL), 3 * O, nullptr));                  // x_loc segment of lin_l / lin_r / cal_x
        CH(PACK(x_ctx_init, O, O, xc_pk[0]));                                                                                                // :306
        float* msg = P(L.XC0);                                                   // [N, O] dense (the [prod | x_ctx | msg] rows of the unchained form are not needed)
        for (int t = 0; t < T; ++t) {
            const int cur = t & 1;
            const float* pc = P(L.pc) + (size_t)t * B * 2 * O;                  // [proj_cmd(cmd_t) | cal_cmd(cmd_t)] per graph
            // prod = proj_x_ctx(x_ctx) * proj_x_loc: packed only                                                 :312-313
            CH(PROD(O, O, xc_pk[cur], nullptr, 3, LinearEpilogue{p->proj_x_ctx_bias, nullptr, 0, P(L.proj_x_loc), O, 0}, nullptr, O, prod_pk));
            // J = XL + [prod | x_ctx] . Wj^T: two K segments                                                     :144-145,230
            CH(PROD(3 * O, O, prod_pk, xc_pk[cur], 2, LinearEpilogue{nullptr, P(L.XL), 3 * O, nullptr, 0, 0}, P(L.J), 3 * O, nullptr));
            if (ss && t == 0) CH(side_join(ss, stream));                          // the command chain's [proj_cmd | cal_cmd] rows
            {
                StageTimer tl(GVQA_STAGE_EDGE_LOGIT, stream);
                const dim3 ngrid((unsigned)cdiv(N, 4));
#define EDGE_LOGIT(KERNEL_)                                                                                              \
                hipLaunchKernelGGL(KERNEL_, ngrid, dim3(256), 0, stream, (int)N, O, P(L.J), (int64_t)3 * O, P(L.J) + O, (int64_t)3 * O, pc,   \
                                   (int64_t)2 * O, g->rowptr, g->csr_src, g->csr_eid, g->node_graph, P(L.logit))
                EDGE_LOGIT((k_lcgn_edge_logit_v<false, 1>));                      // (O % 32 == 0, O <= 512)
#undef EDGE_LOGIT
                GVQA_LAUNCH_CHECK();
            }
            gvqa_gat_mp_desc m;
            memset(&m, 0, sizeof(m));
            m.C = O; m.H = 1; m.negative_slope = d->negative_slope; m.bn_eps = 1e-5f;
            m.xp = P(L.J) + 2 * O; m.xp_ld = 3 * O;
            m.a_edge = P(L.logit); m.a_edge_stride = 1;
            m.graph_scale = pc + O; m.graph_scale_ld = 2 * O;
            m.bias = p->bias; m.out = msg; m.out_ld = O;
            CH(launch_gat_mp_public(g, &m, P(L.alpha), (size_t)E * sizeof(float), stream));                                                  // :209-238,166-168
            CH(PACK(msg, O, O, msg_pk));
            // x_ctx = output_layer([x_ctx | msg]): two K segments, packed only                                   :316-319
            CH(PROD(O, O, xc_pk[cur], msg_pk, 4, LinearEpilogue{p->output_bias, nullptr, 0, nullptr, 0, 0}, nullptr, O, xc_pk[cur ^ 1]));
        }
        // out = fin_layer([x_loc | x_ctx])                                                                        :321-322
        CH(PROD(O, O, xloc_pk, nullptr, 5, LinearEpilogue{p->fin_bias, nullptr, 0, nullptr, 0, 0}, out, O, nullptr));
        CH(PROD(O, O, xc_pk[T & 1], nullptr, 6, LinearEpilogue{nullptr, out, O, nullptr, 0, 0}, out, O, nullptr));
#undef CH
        return GVQA_OK;
    }
    {   // x_loc = init(x)                                                                       lcgn.py:305
        LinearEpilogue e{p->init_bias, nullptr, 0, nullptr, 0, 0};
        if (mf16) {     // x is a node tensor too: bf16 copy with K zero-padded to a multiple of 8
            const int K8 = (int)align_up(Cin, 8);
            StageTimer tp(GVQA_STAGE_PROJ, stream);
            hipLaunchKernelGGL(k_place_rows<true>, dim3((unsigned)cdiv(N * K8, 256)), dim3(256), 0, stream, N, Cin, K8, x,
                               static_cast<void*>(base + L.x16), (int64_t)K8);
            GVQA_LAUNCH_CHECK();
            rc = launch_linear_bf16(N, O, K8, PW, base + L.x16, K8, pkw.init, e, P(L.x_loc), O, true, stream);
            if (rc) return rc;
        } else {
            if (split_nodes) NODE_LIN(N, O, Cin, x, Cin, p->init_weight, Cin, pkw.init, e, P(L.x_loc), O, true, 7);
            else LINT(N, O, Cin, x, Cin, p->init_weight, Cin, e, P(L.x_loc), O, FC);      // (x itself is fp32 in every mode)
        }
    }
    {   // the per-question command chain (all T iterations): "graph_term" -- on the side stream, as in the chained form above
    if (ss) { rc = side_fork(ss, stream); if (rc) return rc; }
    hipStream_t stream_main = stream;
    (void)stream_main;
    hipStream_t stream = ss ? ss->stream : stream_main;
    StageTimer tq(GVQA_STAGE_GRAPH_TERM, stream);
    LIN(B, O, Q, q_encoding, Q, p->qinput1_weight, Q, p->qinput1_bias, 1, P(L.q_emb), O);           // :307
    // textual commands (:292-300) and their projections (:148-149) for all T iterations: per-graph, fp32,
    // independent of the node state -> three launches instead of 3T
    LIN(B, T * O, O, P(L.q_emb), O, PB(PL.Wq), O, PB(PL.bq), 0, P(L.q_cmd), (int64_t)T * O);
    hipLaunchKernelGGL(k_lcgn_command, dim3((unsigned)B, (unsigned)T), dim3(256), (size_t)Lq * sizeof(float), stream, Lq, (int)B,
                       O, P(L.q_cmd), lstm_outputs, p->cmd_logit_weight, p->cmd_logit_bias, P(L.cmd));
    GVQA_LAUNCH_CHECK();
    LIN((int64_t)T * B, 2 * O, O, P(L.cmd), O, PB(PL.Wpc), O, nullptr, 0, P(L.pc), 2 * O);
    }
    {   // proj_x_loc                                                                            :308
        LinearEpilogue e{p->proj_x_loc_bias, nullptr, 0, nullptr, 0, 0};
        NODE_LIN(N, O, O, P(L.x_loc), O, p->proj_x_loc_weight, O, pkw.pxl, e, P(L.proj_x_loc), O, true, 0);
    }
    {   // x_loc segment of lin_l / lin_r / cal_x                                                :144-145,230
        LinearEpilogue e{nullptr, nullptr, 0, nullptr, 0, 0};
        NODE_LIN(N, 3 * O, O, P(L.x_loc), O, PB(PL.Wx), O, pkw.Wx, e, P(L.XL), 3 * O, true, 1);
    }
    // x_ctx (:306) into the x_ctx columns of XC0
    {
    StageTimer to(GVQA_STAGE_OTHER, stream);
    if (nb) hipLaunchKernelGGL(k_place_rows<true>, dim3((unsigned)cdiv(N * O, 256)), dim3(256), 0, stream, N, O, O, x_ctx_init,
                               static_cast<void*>(NP(P(L.XC0), O)), (int64_t)3 * O);
    else hipLaunchKernelGGL(k_place_rows<false>, dim3((unsigned)cdiv(N * O, 256)), dim3(256), 0, stream, N, O, O, x_ctx_init,
                            static_cast<void*>(P(L.XC0) + O), (int64_t)3 * O);
    GVQA_LAUNCH_CHECK();
    }
    const int64_t ldx = 3 * O;
    const bool vec8 = O % 8 == 0;      // 16-byte row segments (all node tensors / per-graph rows have O-multiple offsets)
    for (int t = 0; t < T; ++t) {
        float* XC = (t & 1) ? P(L.XC1) : P(L.XC0);
        float* XCn = (t & 1) ? P(L.XC0) : P(L.XC1);
        float *prod = XC, *x_ctx = NP(XC, O), *msg = NP(XC, 2 * O), *x_ctx_next = NP(XCn, O);
        const float* pc = P(L.pc) + (size_t)t * B * 2 * O;      // [proj_cmd(cmd_t) | cal_cmd(cmd_t)] per graph
        // prod = proj_x_ctx(x_ctx) * proj_x_loc                                                     // :312-313
        LinearEpilogue ep_mul{p->proj_x_ctx_bias, nullptr, 0, P(L.proj_x_loc), O, 0};
        NODE_LIN(N, O, O, x_ctx, ldx, p->proj_x_ctx_weight, O, pkw.pxc, ep_mul, prod, ldx, true, 3);
        // J = x_joint . [lin_l; lin_r; cal_x]^T = XL + [prod | x_ctx] . Wj^T   (one K = 2O product)  // :144-145,230
        LinearEpilogue ep_add{nullptr, P(L.XL), 3 * O, nullptr, 0, 0};
        NODE_LIN(N, 3 * O, 2 * O, prod, ldx, PB(PL.Wj), 2 * O, pkw.Wj, ep_add, P(L.J), 3 * O, true, 2);
        // dot-product attention logits per edge                                                     // :154,207
        const dim3 ngrid((unsigned)cdiv(N, 4));
        if (ss && t == 0) { rc = side_join(ss, stream); if (rc) return rc; }      // the command chain's [proj_cmd | cal_cmd] rows
        {
        StageTimer tl(GVQA_STAGE_EDGE_LOGIT, stream);
#define EDGE_LOGIT(KERNEL_, XL_, XR_)                                                                                    \
        hipLaunchKernelGGL(KERNEL_, ngrid, dim3(256), 0, stream, (int)N, O, XL_, (int64_t)3 * O, XR_, (int64_t)3 * O, pc,  \
                           (int64_t)2 * O, g->rowptr, g->csr_src, g->csr_eid, g->node_graph, P(L.logit))
        if (nb) {
            if (vec8 && O <= 512) EDGE_LOGIT((k_lcgn_edge_logit_v<true, 1>), P(L.J), NP(P(L.J), O));
            else if (vec8) EDGE_LOGIT((k_lcgn_edge_logit_v<true, 2>), P(L.J), NP(P(L.J), O));
            else EDGE_LOGIT(k_lcgn_edge_logit<true>, P(L.J), NP(P(L.J), O));
        } else {
            if (vec8 && O <= 512) EDGE_LOGIT((k_lcgn_edge_logit_v<false, 1>), P(L.J), P(L.J) + O);
            else if (vec8) EDGE_LOGIT((k_lcgn_edge_logit_v<false, 2>), P(L.J), P(L.J) + O);
            else EDGE_LOGIT(k_lcgn_edge_logit<false>, P(L.J), P(L.J) + O);
        }
#undef EDGE_LOGIT
        GVQA_LAUNCH_CHECK();
        }
        // leaky-relu, softmax over in-edges, alpha-weighted sum of cal_x(x_joint)[src], x cal_cmd[g], + bias  // :209-238,166-168
        if (nb) {
            StageTimer tm(GVQA_STAGE_MP, stream);
#define AGGREGATE(KERNEL_)                                                                                              \
            hipLaunchKernelGGL(KERNEL_, ngrid, dim3(256), 0, stream, (int)N, O,                                          \
                               reinterpret_cast<const uint16_t*>(NP(P(L.J), 2 * O)), (int64_t)3 * O, P(L.logit), pc + O,  \
                               (int64_t)2 * O, p->bias, d->negative_slope, g->rowptr, g->csr_src, g->csr_eid, g->node_graph, \
                               reinterpret_cast<uint16_t*>(msg), ldx)
            if (vec8 && O <= 512) AGGREGATE(k_lcgn_aggregate_bf16_v<1>);
            else if (vec8) AGGREGATE(k_lcgn_aggregate_bf16_v<2>);
            else AGGREGATE(k_lcgn_aggregate_bf16);
#undef AGGREGATE
            GVQA_LAUNCH_CHECK();
        } else {
            gvqa_gat_mp_desc m;
            memset(&m, 0, sizeof(m));
            m.C = O; m.H = 1; m.negative_slope = d->negative_slope; m.bn_eps = 1e-5f;
            m.xp = P(L.J) + 2 * O; m.xp_ld = 3 * O;
            m.a_edge = P(L.logit); m.a_edge_stride = 1;
            m.graph_scale = pc + O; m.graph_scale_ld = 2 * O;
            m.bias = p->bias; m.out = msg; m.out_ld = ldx;
            rc = launch_gat_mp_public(g, &m, P(L.alpha), (size_t)E * sizeof(float), stream);
            if (rc) return rc;
        }
        // x_ctx = output_layer([x_ctx || msg])   (one K = 2O product)                                 // :316-319
        LinearEpilogue ep_o{p->output_bias, nullptr, 0, nullptr, 0, 0};
        NODE_LIN(N, O, 2 * O, x_ctx, ldx, p->output_weight, 2 * O, pkw.out, ep_o, x_ctx_next, ldx, true, 4);
    }
    float* x_ctx_fin = NP((T & 1) ? P(L.XC1) : P(L.XC0), O);
    // out = fin_layer([x_loc || x_ctx]) (fp32 result)                                                // :321-322
    LinearEpilogue ep_f1{p->fin_bias, nullptr, 0, nullptr, 0, 0};
    NODE_LIN(N, O, O, P(L.x_loc), O, p->fin_weight, 2 * O, pkw.f1, ep_f1, out, O, false, 5);
    LinearEpilogue ep_fin{nullptr, out, O, nullptr, 0, 0};
    NODE_LIN(N, O, O, x_ctx_fin, ldx, p->fin_weight + O, 2 * O, pkw.f2, ep_fin, out, O, false, 6);
#undef LIN
#undef NODE_LIN
#undef LINT
    return GVQA_OK;
}

}  // extern "C"
