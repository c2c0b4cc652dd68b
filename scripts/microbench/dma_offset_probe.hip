// Does the instruction offset of global_load_lds_dwordx4 apply to the LDS address too?  (hipcc --offload-arch=gfx950 -O2)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) unsigned char* lds_t;
__global__ void probe(const float* src, float* out) {
    __shared__ __attribute__((aligned(1024))) float smem[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) smem[i] = -1.f;
    __syncthreads();
    unsigned base = (unsigned)(size_t)(lds_t)smem;
    const float* p = src + threadIdx.x * 4;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "s_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(p), "s"(base) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = smem[i];
}
int main() {
    float *src, *out, h[2048], hs[1024];
    for (int i = 0; i < 1024; ++i) hs[i] = (float)i;
    hipMalloc(&src, 4096); hipMalloc(&out, 8192);
    hipMemcpy(src, hs, 4096, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, out);
    hipMemcpy(h, out, 8192, hipMemcpyDeviceToHost);
    // expectation if the offset applies to BOTH: lds[0..255] = src[0..255], lds[256..511] = src[256..511], lds[512..767] = src[512..767]
    printf("lds[0]=%g lds[255]=%g lds[256]=%g lds[511]=%g lds[512]=%g lds[767]=%g lds[768]=%g\n", h[0], h[255], h[256], h[511], h[512], h[767], h[768]);
    int both = h[256] == 256.f && h[512] == 512.f && h[767] == 767.f;
    int global_only = h[0] == 512.f || h[0] == 256.f;
    printf("offset applies to: %s\n", both ? "BOTH global and LDS" : global_only ? "GLOBAL only (LDS overwritten in place)" : "unclear");
    return 0;
}
