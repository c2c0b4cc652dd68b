// Micro-benchmark: what bounds the fp32 MFMA GEMM inner loop on gfx950?
// One block = 4 waves, each wave owns a 64 x 64 output tile (2 x 2 MFMA 32x32x2 tiles) and runs KSTEPS
// K steps of 64 MFMAs.  Modes strip the loop down:
//   mode 0: MFMAs only (operands constant registers)            -> pure matrix-core issue rate
//   mode 1: + ds_read_b128 fragment reads from a static LDS tile -> LDS feed
//   mode 2: + one barrier per K step
//   mode 3: + LDS-DMA of the next tile (global_load_lds) issued at the top of the step, counted vmcnt, 2 barriers
//   mode 4: mode 3 + the direct global store of the tile
//   mode 5: ONE barrier per step, the next tile's DMA interleaved into this tile's MFMA stream
// Every wave records shader-clock cycles (s_memtime) and the 100 MHz wall clock (s_memrealtime) around the
// loop: cycles per MFMA and the implied core clock come out directly.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_probe mfma_f32_probe.hip ; run: ./mfma_f32_probe [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned char* lds_bytes_t;
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ A, int64_t lda, int ksteps, float* __restrict__ sink,
                                               unsigned long long* __restrict__ stamps, float* __restrict__ Cout) {
    constexpr int TILE = 16384;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;
    for (int i = tid; i < 4 * TILE / 4; i += 256) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* pa[8];
    for (int j = 0; j < 8; ++j) {
        const int row = ((j & 3) * 4 + wave) * 8 + (lane >> 3);
        // A tile shared by the 16 blocks of a row of tiles, B tile by every 16th block (the reuse of a real GEMM grid)
        const int64_t tile_row = (j >> 2) ? (int64_t)(blockIdx.x & 15) * 128 : (int64_t)(16 + (blockIdx.x >> 4)) * 128;
        pa[j] = A + (tile_row + row) * lda + ((lane & 7) ^ ((row >> 1) & 7)) * 4;
    }
    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 1) & 7;
    unsigned xo[4];
    for (int kg = 0; kg < 4; ++kg) xo[kg] = (unsigned)(((kg * 2 + fh) ^ swz) * 16);
    const unsigned a_row = (unsigned)((wr * 64 + frow) * 128), b_row = (unsigned)((wc * 64 + frow) * 128);
    auto issue = [&](int buf) {
        const unsigned dst = lds_base + buf * 2 * TILE + wave * 1024;
#pragma unroll
        for (int j = 0; j < 8; ++j) { lds_dma16(pa[j], __builtin_amdgcn_readfirstlane(dst + j * 4096)); pa[j] += 32; }
    };
    float4 ca = make_float4(1.f + lane, 2.f, 3.f, 4.f), cb = make_float4(0.5f, 0.25f, 0.125f, 1.f);

    if (MODE >= 3) issue(0);
    auto issue2 = [&](int buf, int j0) {     // two of the eight DMAs of a tile
        const unsigned dst = lds_base + buf * 2 * TILE + wave * 1024;
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) { lds_dma16(pa[j], __builtin_amdgcn_readfirstlane(dst + j * 4096)); pa[j] += 32; }
    };
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int t = 0; t < ksteps; ++t) {
        const int cur = t & 1;
        if (MODE == 5) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE >= 3) {
            if (t + 1 < ksteps) { issue(cur ^ 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (MODE >= 2) __builtin_amdgcn_s_barrier();
        const unsigned char* at = smem + cur * 2 * TILE;
        const unsigned char* bt = at + TILE;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            float4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (MODE >= 1) {
                    af[i] = *reinterpret_cast<const float4*>(at + a_row + i * 4096 + xo[kg]);
                    bf[i] = *reinterpret_cast<const float4*>(bt + b_row + i * 4096 + xo[kg]);
                } else {
                    af[i] = ca; bf[i] = cb;
                    asm volatile("" : "+v"(af[i].x), "+v"(bf[i].x));      // keep the operands opaque
                }
            }
#define MFMA_K(c_)                                                                                      \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].c_, bf[j].c_, acc[i][j], 0, 0, 0);
            MFMA_K(x)
            if (MODE == 5 && t + 1 < ksteps) issue2(cur ^ 1, kg * 2);
            MFMA_K(y) MFMA_K(z) MFMA_K(w)
#undef MFMA_K
        }
        if (MODE >= 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (MODE >= 3 && MODE != 5) __builtin_amdgcn_s_barrier(); }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (MODE == 4) {      // the GEMM's direct epilogue: 64 stores of 128-byte row segments per lane pair
        float* C = Cout + (size_t)(blockIdx.x & 1023) * 128 * 128;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(wr * 64 + i * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2)) * 128 + wc * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
    }
    const unsigned long long c2 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[tid] = s;
    if (lane == 0) {
        stamps[(blockIdx.x * 4 + wave) * 2 + 0] = c1 - c0;
        stamps[(blockIdx.x * 4 + wave) * 2 + 1] = w1 - w0;
        if (MODE == 4 && blockIdx.x == 4000 && wave == 0) printf("epilogue cycles (block 4000): %llu\n", c2 - c1);
    }
}

template <int MODE>
static void run(int blocks, int ksteps, const float* A, int64_t lda, float* sink, unsigned long long* stamps_d, float* Cout) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, A, lda, ksteps, sink, stamps_d, Cout);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, A, lda, ksteps, sink, stamps_d, Cout);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st((size_t)blocks * 8);
    hipMemcpy(st.data(), stamps_d, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int w = 0; w < blocks * 4; ++w) {
        cyc.push_back((double)st[w * 2] / ((double)ksteps * 64));
        clk.push_back((double)st[w * 2] / ((double)st[w * 2 + 1] * 10.0));   // cycles per ns (wall clock 100 MHz) = GHz
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double flops = (double)blocks * 4 * ksteps * 64 * 4096.0;
    printf("mode %d blocks %5d ksteps %4d: %8.1f us  %6.1f TFLOP/s | shader cycles per MFMA per wave: median %6.1f (min %6.1f max %6.1f) | clock %.2f GHz\n",
           MODE, blocks, ksteps, ms * 1e3, flops / (ms * 1e-3) / 1e12, cyc[cyc.size() / 2], cyc.front(), cyc.back(), clk[clk.size() / 2]);
}

int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 2;
    const int ksteps = argc > 2 ? atoi(argv[2]) : 256;
    const int blocks = argc > 3 ? atoi(argv[3]) : 256 * per_cu;
    const int64_t lda = (int64_t)ksteps * 32 + 64;
    float *A, *sink; unsigned long long* stamps;
    const size_t rows = (size_t)(16 + blocks / 16 + 1) * 128;
    hipMalloc(&A, rows * lda * 4); hipMemset(A, 0, rows * lda * 4);
    hipMalloc(&sink, 1024 * 4); hipMalloc(&stamps, (size_t)blocks * 8 * 8);
    float* Cout; hipMalloc(&Cout, (size_t)1024 * 128 * 128 * 4);
    run<0>(blocks, ksteps, A, lda, sink, stamps, Cout);
    run<2>(blocks, ksteps, A, lda, sink, stamps, Cout);
    run<3>(blocks, ksteps, A, lda, sink, stamps, Cout);
    run<4>(blocks, ksteps, A, lda, sink, stamps, Cout);
    run<5>(blocks, ksteps, A, lda, sink, stamps, Cout);
    return 0;
}
