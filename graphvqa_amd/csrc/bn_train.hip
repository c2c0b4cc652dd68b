// BatchNorm1d with batch statistics + ReLU, forward and backward (gat_skip.py:273-275 under model.train(); the
// post-ops of the differentiable path).  x, y, dy, dx are fp32 [N, C] contiguous.  All HBM-bound
// column reductions: a block owns 256 columns x BN_ROWS rows (coalesced 1 KiB row segments), partial sums go
// through a [row blocks, C] workspace, so the result is deterministic.
//   forward : mean, then the variance of the centred values (biased, like torch), y = relu(xhat * w + b)
//   backward: g = dy * [xhat * w + b > 0];  db = sum g;  dw = sum g * xhat;
//             dx = w * invstd * (g - db / N - xhat * dw / N)
#include <algorithm>

#include "common.h"

namespace gvqa {
namespace {

constexpr int BN_ROWS = 64;           // rows per block of the column reductions (N / 64 row blocks x C / 256 column blocks)

// Both statistics from ONE pass over x: a thread keeps its column's (<= 64) rows of the block in registers, forms the block's sum, then the block's
// sum of squared deviations from the BLOCK mean out of the registers; k_bn_col_finish_m2 combines the blocks exactly (Chan et al.:
// M2 = sum_b [M2_b + n_b (mean_b - mean)^2]) -- the accuracy of the two-pass form (deviations from a mean, never x^2), one read of x instead of two.
// partial[b, 0, c] = sum_b, partial[b, 1, c] = M2_b
__global__ __launch_bounds__(256) void k_bn_col_stats_fused(int64_t N, int C, const float* __restrict__ x, float* __restrict__ partial) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS;
    const int n = (int)min<int64_t>(BN_ROWS, N - r0);
    float v[BN_ROWS];
#pragma unroll
    for (int r = 0; r < BN_ROWS; ++r) v[r] = r < n ? x[(r0 + r) * C + c] : 0.f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int r = 0; r < BN_ROWS; r += 4) { a0 += v[r]; a1 += v[r + 1]; a2 += v[r + 2]; a3 += v[r + 3]; }
    const float sum = (a0 + a1) + (a2 + a3), mb = sum / (float)n;
    a0 = a1 = a2 = a3 = 0.f;
#pragma unroll
    for (int r = 0; r < BN_ROWS; r += 4) {
        const float d0 = v[r] - mb, d1 = v[r + 1] - mb, d2 = v[r + 2] - mb, d3 = v[r + 3] - mb;
        a0 += r < n ? d0 * d0 : 0.f; a1 += r + 1 < n ? d1 * d1 : 0.f; a2 += r + 2 < n ? d2 * d2 : 0.f; a3 += r + 3 < n ? d3 * d3 : 0.f;
    }
    partial[((int64_t)blockIdx.x * 2 + 0) * C + c] = sum;
    partial[((int64_t)blockIdx.x * 2 + 1) * C + c] = (a0 + a1) + (a2 + a3);
}

// var[c] = (1 / N) sum_b [ M2_b + n_b (sum_b / n_b - mean[c])^2 ]   (same thread geometry and summation order as k_bn_col_finish)
__global__ __launch_bounds__(1024) void k_bn_col_finish_m2(int nblocks, int C, int64_t N, const float* __restrict__ partial, const float* __restrict__ mean,
                                                           float* __restrict__ var) {
    __shared__ float red[64][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float a0 = 0.f;
    if (c < C) {
        const float m = mean[c];
        for (int b = ry; b < nblocks; b += 64) {
            const float nb = (float)min<int64_t>(BN_ROWS, N - (int64_t)b * BN_ROWS);
            const float d = partial[((int64_t)b * 2 + 0) * C + c] / nb - m;
            a0 += partial[((int64_t)b * 2 + 1) * C + c] + nb * d * d;
        }
    }
    red[ry][cx] = a0;
    __syncthreads();
    if (ry == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) t += red[k][cx];
        var[c] = t / (float)N;
    }
}

// out[c] = scale * sum_b partial[b * blk_stride + c]: a block owns 16 columns, its 64 thread rows sum interleaved
// row blocks (16-deep load chains at 1024 row blocks), then the 64 sums are added in index order
__global__ __launch_bounds__(1024) void k_bn_col_finish(int nblocks, int C, const float* __restrict__ partial, int64_t blk_stride,
                                                        float scale, float* __restrict__ out) {
    __shared__ float red[64][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        int b = ry;
        for (; b + 64 < nblocks; b += 128) {
            a0 += partial[(int64_t)b * blk_stride + c];
            a1 += partial[(int64_t)(b + 64) * blk_stride + c];
        }
        if (b < nblocks) a0 += partial[(int64_t)b * blk_stride + c];
    }
    red[ry][cx] = a0 + a1;
    __syncthreads();
    if (ry == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) t += red[k][cx];
        out[c] = t * scale;
    }
}

// Feature-dropout decisions made in the kernels (Philox4x32-10, the counter-based generator torch's own CUDA dropout uses): the quad of 4
// consecutive channels at element index 4 q gets counter (q, offset), key = seed ^ a library constant; an element is kept when its 32-bit draw is below
// `thresh` = (1 - p) 2^32.  Forward and both backward passes regenerate the same decisions: no [N, C] mask exists in HBM.
struct KeepRng { unsigned long long seed, offset; unsigned thresh; int on; };
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ uchar4 rng_keep4(const KeepRng& g, int64_t quad) {
    const uint4 d = philox4x32_10(make_uint4((unsigned)quad, (unsigned)((unsigned long long)quad >> 32), (unsigned)g.offset, (unsigned)(g.offset >> 32)),
                                  make_uint2((unsigned)g.seed ^ 0x48495031u, (unsigned)(g.seed >> 32) ^ 0x67767161u));
    // (the key is the caller's seed XOR a library constant: with the seed itself -- torch's own key -- and counters laid out (quad, offset) against torch's
    //  (offset / 4, thread), a draw here could coincide with one of torch's dropout / randn kernels under the same torch.manual_seed; a different key is a
    //  different, independent stream whatever the counters are -- ADVICE r05)
    return make_uchar4(d.x < g.thresh, d.y < g.thresh, d.z < g.thresh, d.w < g.thresh);
}
__global__ __launch_bounds__(256) void k_rng_scale_mask(int64_t quads, KeepRng g, float kscale, float* __restrict__ out) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
        const uchar4 k = rng_keep4(g, q);
        *reinterpret_cast<float4*>(out + 4 * q) = make_float4(k.x ? kscale : 0.f, k.y ? kscale : 0.f, k.z ? kscale : 0.f, k.w ? kscale : 0.f);
    }
}
__global__ __launch_bounds__(256) void k_rng_keep_mask(int64_t quads, KeepRng g, uint8_t* __restrict__ out) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256)
        *reinterpret_cast<uchar4*>(out + 4 * q) = rng_keep4(g, q);
}

__global__ __launch_bounds__(256) void k_bn_relu_apply(int64_t total, int C, const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const float* __restrict__ w,
                                                       const float* __restrict__ b, float eps, float* __restrict__ y,
                                                       const uint8_t* __restrict__ keep, float kscale) {
    // keep != NULL: feature dropout applied on the way out (gat_skip.py:276): y *= keep[i] ? kscale : 0
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float invstd = 1.0f / sqrtf(var[c] + eps);
        const float v = fmaxf((x[i] - mean[c]) * invstd * w[c] + b[c], 0.f);
        y[i] = keep ? (keep[i] ? v * kscale : 0.f) : v;
    }
}

// the same with 16-byte accesses (C % 4 == 0): a thread owns 4 consecutive channels and walks rows
__global__ __launch_bounds__(256) void k_bn_relu_apply_v4(int64_t N, int C, int TPR, const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ var, const float* __restrict__ w,
                                                          const float* __restrict__ b, float eps, float* __restrict__ y,
                                                          const uint8_t* __restrict__ keep, float kscale, KeepRng rng) {
    const int tcol = threadIdx.x & (TPR - 1), trow = threadIdx.x / TPR, RPB = 256 / TPR;
    for (int c4 = tcol; c4 < (C >> 2); c4 += TPR) {
        const int c = c4 * 4;
        const float4 m = *reinterpret_cast<const float4*>(mean + c), v = *reinterpret_cast<const float4*>(var + c);
        const float4 wc = *reinterpret_cast<const float4*>(w + c), bc = *reinterpret_cast<const float4*>(b + c);
        const float4 is = make_float4(1.0f / sqrtf(v.x + eps), 1.0f / sqrtf(v.y + eps), 1.0f / sqrtf(v.z + eps), 1.0f / sqrtf(v.w + eps));
        for (int64_t r = (int64_t)blockIdx.x * RPB + trow; r < N; r += (int64_t)gridDim.x * RPB) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * C + c);
            // (x - mean) * invstd * w + b, in the scalar kernel's operation order
            float4 o = make_float4(fmaxf((xv.x - m.x) * is.x * wc.x + bc.x, 0.f), fmaxf((xv.y - m.y) * is.y * wc.y + bc.y, 0.f),
                                   fmaxf((xv.z - m.z) * is.z * wc.z + bc.z, 0.f), fmaxf((xv.w - m.w) * is.w * wc.w + bc.w, 0.f));
            if (keep || rng.on) {
                const uchar4 k4 = rng.on ? rng_keep4(rng, (r * C + c) >> 2) : *reinterpret_cast<const uchar4*>(keep + r * C + c);
                o.x = k4.x ? o.x * kscale : 0.f; o.y = k4.y ? o.y * kscale : 0.f; o.z = k4.z ? o.z * kscale : 0.f; o.w = k4.w ? o.w * kscale : 0.f;
            }
            *reinterpret_cast<float4*>(y + r * C + c) = o;
        }
    }
}

// partial[blk, 0, c] = sum g, partial[blk, 1, c] = sum g * xhat over the block's rows
__global__ __launch_bounds__(256) void k_bn_relu_bwd_reduce(int64_t N, int C, const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ mean, const float* __restrict__ var,
                                                            const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                            float* __restrict__ partial, const uint8_t* __restrict__ keep, float kscale) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS, r1 = min(N, r0 + BN_ROWS);
    const float m = mean[c], invstd = 1.0f / sqrtf(var[c] + eps), wc = w[c], bc = b[c];
    float s0 = 0.f, s1 = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const float xh = (x[r * C + c] - m) * invstd;
        float g = xh * wc + bc > 0.f ? dy[r * C + c] : 0.f;
        if (keep) g = keep[r * C + c] ? g * kscale : 0.f;
        s0 += g;
        s1 += g * xh;
    }
    partial[((int64_t)blockIdx.x * 2 + 0) * C + c] = s0;
    partial[((int64_t)blockIdx.x * 2 + 1) * C + c] = s1;
}

__global__ __launch_bounds__(256) void k_bn_relu_bwd_apply(int64_t total, int C, float inv_n, const float* __restrict__ x,
                                                           const float* __restrict__ dy, const float* __restrict__ mean,
                                                           const float* __restrict__ var, const float* __restrict__ w,
                                                           const float* __restrict__ b, float eps, const float* __restrict__ dbias,
                                                           const float* __restrict__ dweight, float* __restrict__ dx,
                                                           const uint8_t* __restrict__ keep, float kscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float invstd = 1.0f / sqrtf(var[c] + eps);
        const float xh = (x[i] - mean[c]) * invstd;
        float g = xh * w[c] + b[c] > 0.f ? dy[i] : 0.f;
        if (keep) g = keep[i] ? g * kscale : 0.f;
        dx[i] = w[c] * invstd * (g - dbias[c] * inv_n - xh * dweight[c] * inv_n);
    }
}

// The same two passes with 16-byte accesses (C % 4 == 0): a thread owns 4 consecutive channels -- their constants live in
// registers -- and walks rows; TPR threads per row (a power of two >= C / 4, at most 256), 256 / TPR rows per workgroup trip.
__global__ __launch_bounds__(256) void k_bn_relu_bwd_reduce_v4(int64_t N, int C, int TPR, const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ mean, const float* __restrict__ var,
                                                               const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                               float* __restrict__ partial, const uint8_t* __restrict__ keep, float kscale, KeepRng rng) {
    __shared__ float4 red[2][256];
    const int tcol = threadIdx.x & (TPR - 1), trow = threadIdx.x / TPR, RPB = 256 / TPR;
    const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS, r1 = min(N, r0 + BN_ROWS);
    for (int c4b = 0; c4b < (C >> 2); c4b += TPR) {       // (uniform trip count: the barriers below are reached by every thread)
        const bool on = c4b + tcol < (C >> 2);
        const int c = on ? (c4b + tcol) * 4 : 0;
        const float4 m = *reinterpret_cast<const float4*>(mean + c), v = *reinterpret_cast<const float4*>(var + c);
        const float4 wc = *reinterpret_cast<const float4*>(w + c), bc = *reinterpret_cast<const float4*>(b + c);
        const float4 is = make_float4(1.0f / sqrtf(v.x + eps), 1.0f / sqrtf(v.y + eps), 1.0f / sqrtf(v.z + eps), 1.0f / sqrtf(v.w + eps));
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        for (int64_t r = r0 + trow; on && r < r1; r += RPB) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * C + c);
            float4 gv = *reinterpret_cast<const float4*>(dy + r * C + c);
            if (keep || rng.on) {
                const uchar4 k4 = rng.on ? rng_keep4(rng, (r * C + c) >> 2) : *reinterpret_cast<const uchar4*>(keep + r * C + c);
                gv.x = k4.x ? gv.x * kscale : 0.f; gv.y = k4.y ? gv.y * kscale : 0.f; gv.z = k4.z ? gv.z * kscale : 0.f; gv.w = k4.w ? gv.w * kscale : 0.f;
            }
            const float4 xh = make_float4((xv.x - m.x) * is.x, (xv.y - m.y) * is.y, (xv.z - m.z) * is.z, (xv.w - m.w) * is.w);
            const float4 g = make_float4(xh.x * wc.x + bc.x > 0.f ? gv.x : 0.f, xh.y * wc.y + bc.y > 0.f ? gv.y : 0.f,
                                         xh.z * wc.z + bc.z > 0.f ? gv.z : 0.f, xh.w * wc.w + bc.w > 0.f ? gv.w : 0.f);
            s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
            s1.x += g.x * xh.x; s1.y += g.y * xh.y; s1.z += g.z * xh.z; s1.w += g.w * xh.w;
        }
        // the RPB thread rows of this column group, added in index order
        red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
        __syncthreads();
        if (trow == 0 && on) {
            for (int q = 1; q < RPB; ++q) {
                const float4 a0 = red[0][q * TPR + tcol], a1 = red[1][q * TPR + tcol];
                s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
                s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
            }
            *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 0) * C + c) = s0;
            *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 1) * C + c) = s1;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_bn_relu_bwd_apply_v4(int64_t N, int C, int TPR, float inv_n, const float* __restrict__ x,
                                                              const float* __restrict__ dy, const float* __restrict__ mean,
                                                              const float* __restrict__ var, const float* __restrict__ w,
                                                              const float* __restrict__ b, float eps, const float* __restrict__ dbias,
                                                              const float* __restrict__ dweight, float* __restrict__ dx,
                                                              const uint8_t* __restrict__ keep, float kscale, KeepRng rng) {
    const int tcol = threadIdx.x & (TPR - 1), trow = threadIdx.x / TPR, RPB = 256 / TPR;
    for (int c4 = tcol; c4 < (C >> 2); c4 += TPR) {
        const int c = c4 * 4;
        const float4 m = *reinterpret_cast<const float4*>(mean + c), v = *reinterpret_cast<const float4*>(var + c);
        const float4 wc = *reinterpret_cast<const float4*>(w + c), bc = *reinterpret_cast<const float4*>(b + c);
        const float4 db = *reinterpret_cast<const float4*>(dbias + c), dw = *reinterpret_cast<const float4*>(dweight + c);
        const float4 is = make_float4(1.0f / sqrtf(v.x + eps), 1.0f / sqrtf(v.y + eps), 1.0f / sqrtf(v.z + eps), 1.0f / sqrtf(v.w + eps));
        auto one = [&](float xv, float gv, float mm, float ii, float ww, float bb, float dbb, float dww) {
            const float xh = (xv - mm) * ii;
            const float g = xh * ww + bb > 0.f ? gv : 0.f;
            return ww * ii * (g - dbb * inv_n - xh * dww * inv_n);
        };
        for (int64_t r = (int64_t)blockIdx.x * RPB + trow; r < N; r += (int64_t)gridDim.x * RPB) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * C + c);
            float4 gv = *reinterpret_cast<const float4*>(dy + r * C + c);
            if (keep || rng.on) {
                const uchar4 k4 = rng.on ? rng_keep4(rng, (r * C + c) >> 2) : *reinterpret_cast<const uchar4*>(keep + r * C + c);
                gv.x = k4.x ? gv.x * kscale : 0.f; gv.y = k4.y ? gv.y * kscale : 0.f; gv.z = k4.z ? gv.z * kscale : 0.f; gv.w = k4.w ? gv.w * kscale : 0.f;
            }
            *reinterpret_cast<float4*>(dx + r * C + c) =
                make_float4(one(xv.x, gv.x, m.x, is.x, wc.x, bc.x, db.x, dw.x), one(xv.y, gv.y, m.y, is.y, wc.y, bc.y, db.y, dw.y),
                            one(xv.z, gv.z, m.z, is.z, wc.z, bc.z, db.z, dw.z), one(xv.w, gv.w, m.w, is.w, wc.w, bc.w, db.w, dw.w));
        }
    }
}

}  // namespace
}  // namespace gvqa

extern "C" size_t gvqa_bn_train_workspace_bytes(int64_t N, int32_t C) {
    if (N < 0 || C <= 0) return 0;
    return (size_t)gvqa::cdiv(N, gvqa::BN_ROWS) * 2 * C * sizeof(float) + 256;
}

extern "C" int gvqa_bn_relu_train_forward(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, float eps,
                                          float* y, float* save_mean, float* save_var, void* ws, size_t ws_bytes, void* stream_) {
    return gvqa_bn_relu_dropout_train_forward(N, C, x, weight, bias, eps, nullptr, 1.0f, y, save_mean, save_var, ws, ws_bytes, stream_);
}

namespace gvqa {
namespace {
KeepRng keep_rng_of(uint64_t seed, uint64_t offset, float p) {
    KeepRng g;
    g.seed = seed; g.offset = offset; g.on = 1;
    const double keep = 1.0 - (double)p;
    g.thresh = keep >= 1.0 ? 0xFFFFFFFFu : (unsigned)(keep * 4294967296.0);
    return g;
}
const KeepRng kNoRng{0, 0, 0, 0};
}  // namespace
}  // namespace gvqa

static int bn_relu_dropout_train_forward_impl(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, float eps,
                                              const uint8_t* keep, float keep_scale, gvqa::KeepRng rng, float* y, float* save_mean, float* save_var,
                                              void* ws, size_t ws_bytes, void* stream_);
static int bn_relu_dropout_train_backward_impl(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, const float* save_mean,
                                               const float* save_var, float eps, const uint8_t* keep, float keep_scale, gvqa::KeepRng rng, const float* dy,
                                               float* dx, float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream_);

extern "C" int gvqa_bn_relu_dropout_train_forward(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, float eps,
                                                  const uint8_t* keep, float keep_scale, float* y, float* save_mean, float* save_var,
                                                  void* ws, size_t ws_bytes, void* stream_) {
    return bn_relu_dropout_train_forward_impl(N, C, x, weight, bias, eps, keep, keep_scale, gvqa::kNoRng, y, save_mean, save_var, ws, ws_bytes, stream_);
}

// the same with the keep decisions drawn in the kernels (p in [0, 1): y = relu(bn(x)) * (kept ? 1 / (1 - p) : 0)); C % 4 == 0, 16-byte aligned rows
extern "C" int gvqa_bn_relu_dropout_train_forward_rng(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, float eps,
                                                      uint64_t seed, uint64_t offset, float p, float* y, float* save_mean, float* save_var, void* ws,
                                                      size_t ws_bytes, void* stream_) {
    GVQA_REQUIRE(p >= 0.f && p < 1.f, GVQA_E_INVALID, "bn_relu_dropout_train_forward_rng: p in [0, 1)");
    GVQA_REQUIRE(C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, GVQA_E_UNSUPPORTED,
                 "bn_relu_dropout_train_forward_rng: C %% 4 == 0 and 16-byte aligned rows (explicit masks otherwise)");
    return bn_relu_dropout_train_forward_impl(N, C, x, weight, bias, eps, nullptr, 1.0f / (1.0f - p), gvqa::keep_rng_of(seed, offset, p), y, save_mean,
                                              save_var, ws, ws_bytes, stream_);
}

extern "C" int gvqa_bn_relu_dropout_train_backward_rng(int64_t N, int32_t C, const float* x, const float* weight, const float* bias,
                                                       const float* save_mean, const float* save_var, float eps, uint64_t seed, uint64_t offset, float p,
                                                       const float* dy, float* dx, float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream_) {
    GVQA_REQUIRE(p >= 0.f && p < 1.f, GVQA_E_INVALID, "bn_relu_dropout_train_backward_rng: p in [0, 1)");
    GVQA_REQUIRE(C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
                 GVQA_E_UNSUPPORTED, "bn_relu_dropout_train_backward_rng: C %% 4 == 0 and 16-byte aligned rows (explicit masks otherwise)");
    return bn_relu_dropout_train_backward_impl(N, C, x, weight, bias, save_mean, save_var, eps, nullptr, 1.0f / (1.0f - p),
                                               gvqa::keep_rng_of(seed, offset, p), dy, dx, dweight, dbias, ws, ws_bytes, stream_);
}

// the keep decisions of the two _rng entry points as a byte mask [N, C] (tests / reproducing a run's masks); C % 4 == 0
extern "C" int gvqa_dropout_keep_mask(int64_t N, int32_t C, uint64_t seed, uint64_t offset, float p, uint8_t* keep, void* stream_) {
    using namespace gvqa;
    GVQA_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && p >= 0.f && p < 1.f && (reinterpret_cast<uintptr_t>(keep) & 3) == 0, GVQA_E_INVALID,
                 "dropout_keep_mask: C %% 4 == 0, p in [0, 1), 4-byte aligned mask");
    if (N == 0) return GVQA_OK;
    GVQA_REQUIRE(keep, GVQA_E_INVALID, "dropout_keep_mask: null mask");
    const int64_t quads = N * C / 4;
    hipLaunchKernelGGL(k_rng_keep_mask, dim3((unsigned)std::min<int64_t>(cdiv(quads, 256), 8192)), dim3(256), 0, static_cast<hipStream_t>(stream_), quads,
                       keep_rng_of(seed, offset, p), keep);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

// mask[i] = kept ? 1 / (1 - p) : 0 for n floats (n % 4 == 0): the multiplicative mask of F.dropout on the attention coefficients
// (gat_skip.py:205; gvqa_gat_mp_desc.alpha_mask), one launch instead of torch's fill + bernoulli + div
extern "C" int gvqa_dropout_scale_mask(int64_t n, uint64_t seed, uint64_t offset, float p, float* mask, void* stream_) {
    using namespace gvqa;
    GVQA_REQUIRE(n >= 0 && n % 4 == 0 && p >= 0.f && p < 1.f && (reinterpret_cast<uintptr_t>(mask) & 15) == 0, GVQA_E_INVALID,
                 "dropout_scale_mask: n %% 4 == 0, p in [0, 1), 16-byte aligned mask");
    if (n == 0) return GVQA_OK;
    GVQA_REQUIRE(mask, GVQA_E_INVALID, "dropout_scale_mask: null mask");
    hipLaunchKernelGGL(k_rng_scale_mask, dim3((unsigned)std::min<int64_t>(cdiv(n / 4, 256), 8192)), dim3(256), 0, static_cast<hipStream_t>(stream_), n / 4,
                       keep_rng_of(seed, offset, p), 1.0f / (1.0f - p), mask);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

static int bn_relu_dropout_train_forward_impl(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, float eps,
                                              const uint8_t* keep, float keep_scale, gvqa::KeepRng rng, float* y, float* save_mean, float* save_var,
                                              void* ws, size_t ws_bytes, void* stream_) {
    using namespace gvqa;
    GVQA_REQUIRE(N >= 0 && C > 0, GVQA_E_INVALID, "bn_relu_train_forward: bad sizes");
    if (N == 0) return GVQA_OK;
    GVQA_REQUIRE(x && weight && bias && y && save_mean && save_var, GVQA_E_INVALID, "bn_relu_train_forward: null tensor");
    GVQA_REQUIRE(ws && ws_bytes >= gvqa_bn_train_workspace_bytes(N, C), GVQA_E_WORKSPACE, "bn_relu_train_forward: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* partial = static_cast<float*>(ws);
    const int nb = (int)cdiv(N, BN_ROWS);
    GVQA_REQUIRE(cdiv(N, BN_ROWS) < (1ll << 31) && C <= 65535 * 256, GVQA_E_UNSUPPORTED, "bn_relu_train: sizes out of range");
    const dim3 grid((unsigned)nb, (unsigned)cdiv(C, 256)), cgrid((unsigned)cdiv(C, 16));
    // batch mean and biased variance from one pass over x (block sums + block M2, combined exactly)
    hipLaunchKernelGGL(k_bn_col_stats_fused, grid, dim3(256), 0, stream, N, (int)C, x, partial);
    hipLaunchKernelGGL(k_bn_col_finish, cgrid, dim3(1024), 0, stream, nb, (int)C, partial, (int64_t)2 * C, 1.0f / (float)N, save_mean);
    hipLaunchKernelGGL(k_bn_col_finish_m2, cgrid, dim3(1024), 0, stream, nb, (int)C, N, partial, save_mean, save_var);
    const int64_t blocks = std::min<int64_t>(cdiv(N * C, 256), 4096);
    const bool v4 = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && (reinterpret_cast<uintptr_t>(keep) & 3) == 0;
    int TPR = 1;
    while (TPR < 256 && TPR < C / 4) TPR <<= 1;
    GVQA_REQUIRE(!rng.on || v4, GVQA_E_UNSUPPORTED, "bn_relu_dropout_train: in-kernel masks need the 16-byte form");
    if (v4) hipLaunchKernelGGL(k_bn_relu_apply_v4, dim3((unsigned)std::min<int64_t>(cdiv(N, 256 / TPR), 4096)), dim3(256), 0, stream, N, (int)C, TPR, x,
                               save_mean, save_var, weight, bias, eps, y, keep, keep_scale, rng);
    else hipLaunchKernelGGL(k_bn_relu_apply, dim3((unsigned)blocks), dim3(256), 0, stream, N * C, (int)C, x, save_mean, save_var, weight, bias,
                            eps, y, keep, keep_scale);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

extern "C" int gvqa_bn_relu_train_backward(int64_t N, int32_t C, const float* x, const float* weight, const float* bias,
                                           const float* save_mean, const float* save_var, float eps, const float* dy, float* dx,
                                           float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream_) {
    return gvqa_bn_relu_dropout_train_backward(N, C, x, weight, bias, save_mean, save_var, eps, nullptr, 1.0f, dy, dx, dweight, dbias, ws, ws_bytes,
                                               stream_);
}

extern "C" int gvqa_bn_relu_dropout_train_backward(int64_t N, int32_t C, const float* x, const float* weight, const float* bias,
                                                   const float* save_mean, const float* save_var, float eps, const uint8_t* keep,
                                                   float keep_scale, const float* dy, float* dx, float* dweight, float* dbias, void* ws,
                                                   size_t ws_bytes, void* stream_) {
    return bn_relu_dropout_train_backward_impl(N, C, x, weight, bias, save_mean, save_var, eps, keep, keep_scale, gvqa::kNoRng, dy, dx, dweight, dbias, ws,
                                               ws_bytes, stream_);
}

static int bn_relu_dropout_train_backward_impl(int64_t N, int32_t C, const float* x, const float* weight, const float* bias, const float* save_mean,
                                               const float* save_var, float eps, const uint8_t* keep, float keep_scale, gvqa::KeepRng rng, const float* dy,
                                               float* dx, float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream_) {
    using namespace gvqa;
    GVQA_REQUIRE(N >= 0 && C > 0, GVQA_E_INVALID, "bn_relu_train_backward: bad sizes");
    if (N == 0) return GVQA_OK;
    GVQA_REQUIRE(x && weight && bias && save_mean && save_var && dy && dx && dweight && dbias, GVQA_E_INVALID,
                 "bn_relu_train_backward: null tensor");
    GVQA_REQUIRE(ws && ws_bytes >= gvqa_bn_train_workspace_bytes(N, C), GVQA_E_WORKSPACE, "bn_relu_train_backward: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float* partial = static_cast<float*>(ws);
    const int nb = (int)cdiv(N, BN_ROWS);
    GVQA_REQUIRE(cdiv(N, BN_ROWS) < (1ll << 31) && C <= 65535 * 256, GVQA_E_UNSUPPORTED, "bn_relu_train: sizes out of range");
    const dim3 grid((unsigned)nb, (unsigned)cdiv(C, 256)), cgrid((unsigned)cdiv(C, 16));
    // 16-byte form when the rows allow it (same operations per element, same summation order within a row block's thread rows)
    const bool v4 = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(keep) & 3) == 0;
    int TPR = 1;
    while (TPR < 256 && TPR < C / 4) TPR <<= 1;
    GVQA_REQUIRE(!rng.on || v4, GVQA_E_UNSUPPORTED, "bn_relu_dropout_train: in-kernel masks need the 16-byte form");
    if (v4) hipLaunchKernelGGL(k_bn_relu_bwd_reduce_v4, dim3((unsigned)nb), dim3(256), 0, stream, N, (int)C, TPR, x, dy, save_mean, save_var, weight,
                               bias, eps, partial, keep, keep_scale, rng);
    else hipLaunchKernelGGL(k_bn_relu_bwd_reduce, grid, dim3(256), 0, stream, N, (int)C, x, dy, save_mean, save_var, weight, bias, eps, partial, keep,
                            keep_scale);
    hipLaunchKernelGGL(k_bn_col_finish, cgrid, dim3(1024), 0, stream, nb, (int)C, partial, (int64_t)2 * C, 1.0f, dbias);
    hipLaunchKernelGGL(k_bn_col_finish, cgrid, dim3(1024), 0, stream, nb, (int)C, partial + C, (int64_t)2 * C, 1.0f, dweight);
    const int64_t blocks = std::min<int64_t>(cdiv(N * C, 256), 4096);
    if (v4) hipLaunchKernelGGL(k_bn_relu_bwd_apply_v4, dim3((unsigned)std::min<int64_t>(cdiv(N, 256 / TPR), 4096)), dim3(256), 0, stream, N, (int)C,
                               TPR, 1.0f / (float)N, x, dy, save_mean, save_var, weight, bias, eps, dbias, dweight, dx, keep, keep_scale, rng);
    else hipLaunchKernelGGL(k_bn_relu_bwd_apply, dim3((unsigned)blocks), dim3(256), 0, stream, N * C, (int)C, 1.0f / (float)N, x, dy, save_mean,
                            save_var, weight, bias, eps, dbias, dweight, dx, keep, keep_scale);
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}
