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
#include "common.h"

namespace gvqa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;   // padded leading dimension (floats)

template <int BM, int BN, int WR, int WC>
__global__ __launch_bounds__(256) void k_linear_f32(int M, int N, int K, const float* __restrict__ A,
                                                    int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                    const float* __restrict__ bias, int relu,
                                                    float* __restrict__ C, int64_t ldc, int64_t strideA,
                                                    int64_t strideB, int64_t strideC, int vec_ok) {
    static_assert(WR * WC == 4, "4 waves per block");
    constexpr int WM = BM / WR, WN = BN / WC;     // wave tile
    constexpr int MT = WM / 32, NT = WN / 32;     // 32x32 MFMA tiles per wave
    constexpr int A_V4 = BM * BK / 4 / 256;       // float4 staged per thread
    constexpr int B_V4 = (BN * BK / 4 + 255) / 256;
    static_assert(BM * BK / 4 % 256 == 0, "A tile must divide evenly");

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

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            int idx = tid + i * 256;
            int r = idx >> 3, c4 = (idx & 7) * 4;
            int gr = m0 + r, gk = k0 + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < M) {
                const float* p = A + (int64_t)gr * lda + gk;
                if (vec_ok && gk + 3 < K) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (gk + 0 < K) v.x = p[0];
                    if (gk + 1 < K) v.y = p[1];
                    if (gk + 2 < K) v.z = p[2];
                    if (gk + 3 < K) v.w = p[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            int idx = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < BN * BK / 4) {
                int r = idx >> 3, c4 = (idx & 7) * 4;
                int gr = n0 + r, gk = k0 + c4;
                if (gr < N) {
                    const float* p = B + (int64_t)gr * ldb + gk;
                    if (vec_ok && gk + 3 < K) {
                        v = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (gk + 0 < K) v.x = p[0];
                        if (gk + 1 < K) v.y = p[1];
                        if (gk + 2 < K) v.z = p[2];
                        if (gk + 3 < K) v.w = p[3];
                    }
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            int idx = tid + i * 256;
            int r = idx >> 3, c4 = (idx & 7) * 4;
            *reinterpret_cast<float4*>(&As[buf][r * LDS_LD + c4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            int idx = tid + i * 256;
            if (idx < BN * BK / 4) {
                int r = idx >> 3, c4 = (idx & 7) * 4;
                *reinterpret_cast<float4*>(&Bs[buf][r * LDS_LD + c4]) = rb[i];
            }
        }
    };

    const int nkt = (K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
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
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }

    // Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gc = n0 + wc * WN + j * 32 + ccol;
        if (gc >= N) continue;
        const float bv = bias ? bias[gc] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int gr0 = m0 + wr * WM + i * 32 + crow0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = gr0 + (r & 3) + 8 * (r >> 2);
                if (gr < M) {
                    float v = acc[i][j][r] + bv;
                    if (relu) v = fmaxf(v, 0.f);
                    C[(int64_t)gr * ldc + gc] = v;
                }
            }
        }
    }
}

int launch_linear(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                  int64_t ldb, const float* bias, int relu, float* C, int64_t ldc, int batch,
                  int64_t strideA, int64_t strideB, int64_t strideC, hipStream_t stream) {
    GVQA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 1, GVQA_E_INVALID, "linear: negative size");
    GVQA_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), GVQA_E_INVALID, "linear: size overflow");
    if (M == 0 || N == 0) return GVQA_OK;
    GVQA_REQUIRE(cdiv(M, 128) <= 65535, GVQA_E_INVALID, "linear: M too large for one launch");
    GVQA_REQUIRE(A && B && C, GVQA_E_INVALID, "linear: null operand");
    GVQA_REQUIRE(lda >= K && ldb >= K && ldc >= N, GVQA_E_INVALID, "linear: leading dimension too small");
    // 16-byte vector loads need 16-byte aligned rows
    int vec_ok = (lda % 4 == 0) && (ldb % 4 == 0) && (strideA % 4 == 0) && (strideB % 4 == 0) &&
                 ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    if (N <= 32) {
        dim3 grid((unsigned)cdiv(N, 32), (unsigned)cdiv(M, 128), (unsigned)batch);
        hipLaunchKernelGGL((k_linear_f32<128, 32, 4, 1>), grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, A,
                           lda, B, ldb, bias, relu, C, ldc, strideA, strideB, strideC, vec_ok);
    } else if (N <= 64) {
        dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 128), (unsigned)batch);
        hipLaunchKernelGGL((k_linear_f32<128, 64, 2, 2>), grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, A,
                           lda, B, ldb, bias, relu, C, ldc, strideA, strideB, strideC, vec_ok);
    } else {
        dim3 grid((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 128), (unsigned)batch);
        hipLaunchKernelGGL((k_linear_f32<128, 128, 2, 2>), grid, dim3(256), 0, stream, (int)M, (int)N, (int)K, A,
                           lda, B, ldb, bias, relu, C, ldc, strideA, strideB, strideC, vec_ok);
    }
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

}  // namespace gvqa

extern "C" int gvqa_linear_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                               int64_t ldb, const float* bias, int relu, float* C, int64_t ldc, void* stream) {
    return gvqa::launch_linear(M, N, K, A, lda, B, ldb, bias, relu, C, ldc, 1, 0, 0, 0,
                               static_cast<hipStream_t>(stream));
}
