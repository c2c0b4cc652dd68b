// Pieces shared by the LDS-tiled MFMA GEMM kernels (gemm.hip, gemm_bf16.hip): the block-tile epilogue and
// the LDS-DMA issue helper.
#pragma once
#include <type_traits>

#include "common.h"

namespace gvqa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Epilogue of one BM x BN block tile held in MFMA accumulators (shared by the two kernels below).
// `smem` must be free (all operand reads retired by a barrier) and hold BM x (BN + 4) floats.
template <int BM, int BN, int WR, int WC, bool C16, typename TC, typename ACC>
__device__ __forceinline__ void tile_epilogue(ACC& acc, unsigned char* smem, int M, int N, int m0, int n0,
                                              const LinearEpilogue& ep, TC* C, int64_t ldc, int vec_ep) {
    constexpr int NTH = 64 * WR * WC, WM = BM / WR, WN = BN / WC, MT = WM / 32, NT = WN / 32, ST_LD = BN + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    // Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
    if (vec_ep) {
        // through LDS (the operand buffers are free after the last barrier): every thread then owns 8
        // consecutive columns of a row, so C, addend and mul move as 16-byte (bf16) / 2 x 16-byte (fp32)
        // accesses, 16 threads per 128-column row segment, instead of one element per lane
        float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(wr * WM + i * 32 + crow0 + (r & 3) + 8 * (r >> 2)) * ST_LD + wc * WN + j * 32 + ccol] = acc[i][j][r];
        __syncthreads();
        constexpr int CQ = BN / 8;                // 8-column chunks per tile row
#pragma unroll 2
        for (int idx = tid; idx < BM * CQ; idx += NTH) {
            const int row = idx / CQ, col = (idx % CQ) * 8;
            const int gr = m0 + row, gc = n0 + col;
            if (gr >= M || gc >= N) continue;     // N % 8 == 0: a chunk is entirely inside or outside
            float v[8];
            *reinterpret_cast<float4*>(&v[0]) = *reinterpret_cast<const float4*>(&stage[row * ST_LD + col]);
            *reinterpret_cast<float4*>(&v[4]) = *reinterpret_cast<const float4*>(&stage[row * ST_LD + col + 4]);
            if (ep.bias) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += ep.bias[gc + q];
            }
            auto load8c = [&](const float* base, int64_t elem, float (&o)[8]) {
                if constexpr (C16) {
                    const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + elem);
                    o[0] = __uint_as_float(raw.x << 16); o[1] = __uint_as_float(raw.x & 0xFFFF0000u);
                    o[2] = __uint_as_float(raw.y << 16); o[3] = __uint_as_float(raw.y & 0xFFFF0000u);
                    o[4] = __uint_as_float(raw.z << 16); o[5] = __uint_as_float(raw.z & 0xFFFF0000u);
                    o[6] = __uint_as_float(raw.w << 16); o[7] = __uint_as_float(raw.w & 0xFFFF0000u);
                } else {
                    *reinterpret_cast<float4*>(&o[0]) = *reinterpret_cast<const float4*>(base + elem);
                    *reinterpret_cast<float4*>(&o[4]) = *reinterpret_cast<const float4*>(base + elem + 4);
                }
            };
            if (ep.addend) {
                float a[8];
                load8c(ep.addend, (int64_t)gr * ep.ld_add + gc, a);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += a[q];
            }
            if (ep.mul) {
                float a[8];
                load8c(ep.mul, (int64_t)gr * ep.ld_mul + gc, a);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] *= a[q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (ep.relu == 1) v[q] = fmaxf(v[q], 0.f);
                else if (ep.relu == 2) v[q] = v[q] > 0.f ? v[q] : expf(v[q]) - 1.f;
            }
            if constexpr (C16) {
                uint4 o;
                o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
                o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
                *reinterpret_cast<uint4*>(C + (int64_t)gr * ldc + gc) = o;
            } else {
                *reinterpret_cast<float4*>(C + (int64_t)gr * ldc + gc) = *reinterpret_cast<const float4*>(&v[0]);
                *reinterpret_cast<float4*>(C + (int64_t)gr * ldc + gc + 4) = *reinterpret_cast<const float4*>(&v[4]);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int gc = n0 + wc * WN + j * 32 + ccol;
        if (gc >= N) continue;
        const float bv = ep.bias ? ep.bias[gc] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int gr0 = m0 + wr * WM + i * 32 + crow0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = gr0 + (r & 3) + 8 * (r >> 2);
                if (gr < M) {
                    float v = acc[i][j][r] + bv;
                    if (ep.addend) v += load_elem<C16>(ep.addend, (int64_t)gr * ep.ld_add + gc);
                    if (ep.mul) v *= load_elem<C16>(ep.mul, (int64_t)gr * ep.ld_mul + gc);
                    if (ep.relu == 1) v = fmaxf(v, 0.f);
                    else if (ep.relu == 2) v = v > 0.f ? v : expf(v) - 1.f;
                    if constexpr (C16) C[(int64_t)gr * ldc + gc] = f32_to_bf16(v);
                    else C[(int64_t)gr * ldc + gc] = v;
                }
            }
        }
    }
}

__device__ __forceinline__ int split2h_exponent(float rowmax) {
    if (!(rowmax > 0.f) || !(rowmax <= 3.0e38f)) return 0;           // zero rows; inf / nan rows are garbage-in / garbage-out
    int e;
    frexpf(rowmax, &e);                                               // rowmax = f 2^e, f in [0.5, 1)
    return max(-114, min(126, 14 - e));                               // 2^e and 2^-e are normal fp32 numbers
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }

typedef __attribute__((address_space(3))) unsigned char* lds_bytes_t;
__device__ __forceinline__ void lds_dma16_b(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// Three 16-byte DMAs 1 KiB apart in BOTH global memory and LDS from one M0 set-up: the instruction offset of
// global_load_lds is added to the global address and to the LDS address alike (scripts/microbench/dma_offset_probe.hip).
__device__ __forceinline__ void lds_dma16_x3(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// the same for two fragments (the two fp16 pieces of split2h)
__device__ __forceinline__ void lds_dma16_x2(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// ... and four (two K steps x two fp16 pieces of one operand tile: 4 KiB contiguous on both sides)
__device__ __forceinline__ void lds_dma16_x4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// The same groups with the source as SCALAR base + 32-bit lane offset (`global_load_lds_dwordx4 voff, s[base:base+1]`) and M0 declared clobbered instead
// of saved and restored: a packed operand's fragment is contiguous, so the lane part is the constant lane * 16 and the base advances in SGPRs -- no
// per-lane 64-bit pointer arithmetic and two SALU instructions less per group (round 6: +4 % on the 256 x 256 bf16 kernel, `profiles/r06_bf16_big_ablation.jsonl`)
template <int NX>
__device__ __forceinline__ void lds_dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    static_assert(NX >= 1 && NX <= 4, "one to four 16-byte DMAs 1 KiB apart");
    if constexpr (NX == 1)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
    else if constexpr (NX == 2)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                     : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
    else if constexpr (NX == 3)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048"
                     : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                     : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// 4 bytes per lane: LDS byte address `lds_dst` + lane * 4 (no alignment requirement beyond 4 bytes on either side)
__device__ __forceinline__ void lds_dma4_b(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

}  // namespace gvqa
