// Micro-benchmark: floor of the MP kernel's streaming structure on gfx950.
// grid = B blocks x 512 threads, per block T stages of an [n x cw] fp32 tile DMA'd HBM->LDS
// (global_load_lds_dwordx4), NBUF stage buffers, counted vmcnt + raw barrier, no compute.
//   mode 0: rows strided like xp[N, H*C] (row segments of cw*4 bytes at stride H*C*4)
//   mode 1: every stage tile contiguous (n*cw*4 bytes)
// build: hipcc --offload-arch=gfx950 -O3 -o dma_stream dma_stream.hip ; run: ./dma_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) char* lds_ptr_t;
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NBUF>
__global__ __launch_bounds__(512) void k_stream(const float* __restrict__ xp, float* __restrict__ out, int n, int H, int C, int cw,
                                                 int mode, int lds_pad_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, g = blockIdx.x;
    const int wave_unit0 = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int q4 = cw >> 2, units = n * q4;                      // float4 units per stage
    const int nch = C / cw, T = nch * H;
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned buf_bytes = ((units + 511) / 512) * 512 * 16;
    const int row0 = tid / q4, col0 = tid - row0 * q4, rstep = 512 / q4;
    auto prefetch = [&](int t) {
        const int cr = t / H, h = t - cr * H;
        unsigned dst = lds_base + (t % NBUF) * buf_bytes + wave_unit0 * 16u;
        if (mode == 0) {
            const float* base = xp + ((size_t)g * n * H + h) * C + cr * cw;
            int row = row0;
            for (int u0 = 0; u0 < units; u0 += 512) {
                const int r = row < n ? row : n - 1;
                lds_dma16(base + (size_t)r * H * C + col0 * 4, __builtin_amdgcn_readfirstlane(dst));
                dst += 512 * 16; row += rstep;
            }
        } else {
            const float* base = xp + ((size_t)g * T + t) * (size_t)units * 4;
            for (int u0 = 0; u0 < units; u0 += 512) {
                const int u = u0 + tid < units ? u0 + tid : units - 1;
                lds_dma16(base + (size_t)u * 4, __builtin_amdgcn_readfirstlane(dst));
                dst += 512 * 16;
            }
        }
    };
    const int per_stage = (units + 511) / 512;
    for (int t = 0; t < NBUF - 1 && t < T; ++t) prefetch(t);
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NBUF == 3) { if (t + 1 < T) { if (per_stage == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + NBUF - 1 < T) prefetch(t + NBUF - 1);
        acc += reinterpret_cast<const float*>(smem + (t % NBUF) * buf_bytes)[tid];   // touch the stage
    }
    if (acc == 12345.678f) out[g] = acc;
}

// The same bytes with ordinary loads into registers (UNR x 16 B per thread in flight), grid-stride over contiguous memory:
// what a non-DMA streaming kernel gets from the memory system.
template <int UNR>
__global__ __launch_bounds__(256) void k_read_regs(const float4* __restrict__ p, size_t n4, float* __restrict__ out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
// ... and LDS-DMA with NO barrier and no LDS reads: every wave keeps DEPTH 1 KiB DMAs in flight into its own LDS slice
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_dma(const float* __restrict__ p, size_t n4, float* __restrict__ out) {
    __shared__ __attribute__((aligned(1024))) char ring[4 * DEPTH * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)(lds_ptr_t)ring + wave * DEPTH * 1024;
    const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + wave;
    size_t chunk = w;                                  // 1 KiB chunks (64 float4), wave-strided
    const size_t nchunks = n4 / 64;
    int slot = 0;
    for (; chunk < nchunks; chunk += waves) {
        lds_dma16(p + chunk * 256 + lane * 4, __builtin_amdgcn_readfirstlane(base + slot * 1024));
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (reinterpret_cast<float*>(ring)[threadIdx.x] == 12345.678f) out[0] = 1.f;
}

int main(int argc, char** argv) {
    const int B = 2048, n = 32, H = 4, C = 512;
    const size_t elems = (size_t)B * n * H * C;
    float *xp, *out;
    hipMalloc(&xp, elems * 4); hipMalloc(&out, B * 4);
    hipMemset(xp, 0x3c, elems * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { int nbuf, cw, mode, extra_lds; };
    std::vector<Cfg> cfgs = {{2,128,0,11000},{2,128,1,11000},{3,128,0,11000},{3,128,1,11000},{2,256,0,11000},{2,256,1,11000},
                             {2,64,0,11000},{2,64,1,11000},{2,128,0,0},{2,128,1,0},{3,64,0,0},{3,64,1,0},{2,512,1,0}};
    for (auto c : cfgs) {
        const int q4 = c.cw / 4, units = n * q4;
        const size_t lds = (size_t)((units + 511) / 512) * 512 * 16 * c.nbuf + c.extra_lds;
        auto launch = [&]() {
            if (c.nbuf == 2) { hipFuncSetAttribute((const void*)&k_stream<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                               hipLaunchKernelGGL(k_stream<2>, dim3(B), dim3(512), lds, 0, xp, out, n, H, C, c.cw, c.mode, 0); }
            else { hipFuncSetAttribute((const void*)&k_stream<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                   hipLaunchKernelGGL(k_stream<3>, dim3(B), dim3(512), lds, 0, xp, out, n, H, C, c.cw, c.mode, 0); }
        };
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / 20 * 1e3, gbs = elems * 4.0 / (us * 1e-6) / 1e9;
        printf("nbuf=%d cw=%3d mode=%s lds=%6zu B  blocks/CU=%d : %7.1f us  %6.0f GB/s  (%s)\n", c.nbuf, c.cw, c.mode ? "contig " : "strided",
               lds, (int)(163840 / lds), us, gbs, hipGetErrorString(hipGetLastError()));
    }
    {
        const size_t n4 = elems / 4;
        auto timeit = [&](auto launch, const char* name) {
            for (int i = 0; i < 3; ++i) launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-44s %7.1f us  %6.0f GB/s (%s)\n", name, ms / 20 * 1e3, elems * 4.0 / (ms / 20 * 1e-3) / 1e9, hipGetErrorString(hipGetLastError()));
        };
        const float4* p4 = reinterpret_cast<const float4*>(xp);
        timeit([&]() { hipLaunchKernelGGL(k_read_regs<4>, dim3(256 * 8), dim3(256), 0, 0, p4, n4, out); }, "register loads, 4 x 16 B/thread, 8 blocks/CU");
        timeit([&]() { hipLaunchKernelGGL(k_read_regs<8>, dim3(256 * 8), dim3(256), 0, 0, p4, n4, out); }, "register loads, 8 x 16 B/thread, 8 blocks/CU");
        timeit([&]() { hipLaunchKernelGGL(k_read_regs<8>, dim3(256 * 4), dim3(256), 0, 0, p4, n4, out); }, "register loads, 8 x 16 B/thread, 4 blocks/CU");
        timeit([&]() { hipLaunchKernelGGL(k_read_dma<4>, dim3(256 * 8), dim3(256), 0, 0, xp, n4, out); }, "LDS-DMA, 4 KiB/wave in flight, 8 blocks/CU");
        timeit([&]() { hipLaunchKernelGGL(k_read_dma<8>, dim3(256 * 4), dim3(256), 0, 0, xp, n4, out); }, "LDS-DMA, 8 KiB/wave in flight, 4 blocks/CU");
        timeit([&]() { hipLaunchKernelGGL(k_read_dma<16>, dim3(256 * 2), dim3(256), 0, 0, xp, n4, out); }, "LDS-DMA, 16 KiB/wave in flight, 2 blocks/CU");
    }
    return 0;
}
