// Which workgroups of a 512-workgroup launch (256 threads, 80 KiB LDS: two per CU) share a CU?  Every workgroup records its
// XCC id and HW_ID (cu / sh / se fields) and its start time; the host prints how the partner of workgroup w relates to w.
// hipcc --offload-arch=gfx950 -O2 wg_census.hip -o wg_census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void census(unsigned* out, int spin) {
    __shared__ unsigned char smem[80 * 1024];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);     // keep every workgroup resident for a while
    __syncthreads();
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x + 0] = hwid;
        out[4 * blockIdx.x + 1] = xcc;
        out[4 * blockIdx.x + 2] = (unsigned)t0;
        out[4 * blockIdx.x + 3] = smem[5];
    }
}
int main() {
    const int nwg = 512;
    unsigned* d; hipMalloc(&d, nwg * 16);
    std::vector<unsigned> h(nwg * 4);
    for (int rep = 0; rep < 2; ++rep) {
        census<<<nwg, 256>>>(d, 2000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, nwg * 16, hipMemcpyDeviceToHost);
        std::map<unsigned long long, std::vector<int>> cu;
        for (int w = 0; w < nwg; ++w) {
            const unsigned hw = h[4 * w], xcc = h[4 * w + 1] & 0xf;
            const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(w);
        }
        std::map<int, int> delta;
        int maxper = 0;
        for (auto& kv : cu) {
            if ((int)kv.second.size() > maxper) maxper = (int)kv.second.size();
            if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
        }
        printf("rep %d: distinct CUs %zu, max workgroups per CU %d; partner deltas:", rep, cu.size(), maxper);
        for (auto& kv : delta) printf(" %d x%d", kv.first, kv.second);
        printf("\n  first 24 workgroups (w: xcc se sh cu):");
        for (int w = 0; w < 24; ++w) {
            const unsigned hw = h[4 * w];
            printf(" %d:%u/%u/%u/%u", w, h[4 * w + 1] & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf);
        }
        printf("\n  xcc of w = 0..15:");
        for (int w = 0; w < 16; ++w) printf(" %u", h[4 * w + 1] & 0xf);
        printf("\n");
    }
    return 0;
}
