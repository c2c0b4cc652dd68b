// Error reporting, version string and in-library stage timing for libgvqa_hip.so.
#include <stdarg.h>

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"
#include "gemm_tile.h"

namespace gvqa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- run-time options (gvqa_set_option); initial values from the environment ---------------------
static std::atomic<int> g_opt[GVQA_NUM_OPTIONS];
static std::once_flag g_opt_once;
static void options_init() {
    auto env_int = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
    const char* proj = getenv("GVQA_PROJ");
    g_opt[GVQA_OPT_PROJECTION] = (proj && !strcmp(proj, "f32")) ? GVQA_PROJECTION_F32
                               : (proj && !strcmp(proj, "split3")) ? GVQA_PROJECTION_SPLIT3 : GVQA_PROJECTION_SPLIT2H;
    const char* be = getenv("GVQA_GEMM_BACKEND");
    g_opt[GVQA_OPT_VENDOR_GEMM] = (be && !strcmp(be, "rocblas")) ? 1 : 0;
    g_opt[GVQA_OPT_SPLIT3_MIN_MFLOP] = env_int("GVQA_SPLIT3_MIN_MFLOP", 1000);
    g_opt[GVQA_OPT_SPLIT3_VARIANT] = env_int("GVQA_SPLIT3_VARIANT", 0);
    g_opt[GVQA_OPT_HOP_FUSION] = env_int("GVQA_HOP_FUSION", 3);
    g_opt[GVQA_OPT_COEFF_KERNEL] = 0;
    g_opt[GVQA_OPT_MP_PARTS] = 0;
    g_opt[GVQA_OPT_HOP_COEFFS] = env_int("GVQA_HOP_COEFFS", 2);
    g_opt[GVQA_OPT_HOP_HALF_TILES] = env_int("GVQA_HOP_HALF_TILES", 1);
    g_opt[GVQA_OPT_TN_DIRECT] = env_int("GVQA_TN_DIRECT", 1);
    g_opt[GVQA_OPT_PACKED_GROUPS] = env_int("GVQA_PACKED_GROUPS", 1);
}
int get_option(int option) {
    std::call_once(g_opt_once, options_init);
    return (option >= 0 && option < GVQA_NUM_OPTIONS) ? g_opt[option].load(std::memory_order_relaxed) : 0;
}

// ---- side stream (see common.h) ------------------------------------------------------------
SideStream* side_stream_get() {
    static thread_local SideStream ss;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (!ss.ok || ss.device != dev) {
        ss.ok = hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&ss.fork_ev, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&ss.join_ev, hipEventDisableTiming) == hipSuccess;
        ss.device = dev;
    }
    return ss.ok ? &ss : nullptr;
}
int side_fork(SideStream* ss, hipStream_t main) {
    GVQA_HIP_CHECK(hipEventRecord(ss->fork_ev, main));
    GVQA_HIP_CHECK(hipStreamWaitEvent(ss->stream, ss->fork_ev, 0));
    return GVQA_OK;
}
int side_join(SideStream* ss, hipStream_t main) {
    GVQA_HIP_CHECK(hipEventRecord(ss->join_ev, ss->stream));
    GVQA_HIP_CHECK(hipStreamWaitEvent(main, ss->join_ev, 0));
    return GVQA_OK;
}

// ---- stage timing -------------------------------------------------------------------------
struct ProfSlot {
    int stage;
    hipEvent_t start, stop;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static unsigned g_prof_mask = ~0u;          // stages that record events (gvqa_prof_enable)
static std::vector<ProfSlot> g_prof_live;     // recorded, not yet collected
static std::vector<ProfSlot> g_prof_free;     // event pairs available for reuse

StageTimer::StageTimer(int st, hipStream_t s) : stage(st), stream(s), slot(nullptr) {
    if (!g_prof_on || !((g_prof_mask >> st) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfSlot ps;
    if (!g_prof_free.empty()) {
        ps = g_prof_free.back();
        g_prof_free.pop_back();
    } else {
        if (hipEventCreate(&ps.start) != hipSuccess) return;
        if (hipEventCreate(&ps.stop) != hipSuccess) return;
    }
    ps.stage = st;
    (void)hipEventRecord(ps.start, s);
    g_prof_live.push_back(ps);
    slot = reinterpret_cast<void*>(g_prof_live.size());   // 1-based index
}

StageTimer::~StageTimer() {
    if (!slot) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    size_t idx = reinterpret_cast<size_t>(slot) - 1;
    if (idx < g_prof_live.size()) (void)hipEventRecord(g_prof_live[idx].stop, stream);
}

// ---- streaming copies: the "measured device copy" denominator of SURVEY 8(d) (gvqa_stream_copy) -------------------------------
// variant 0 / 1: grid-stride float4 loads and (non-temporal) stores, UNR loads in flight per thread
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_stream_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (NT) __builtin_nontemporal_store(v4{v[u].x, v[u].y, v[u].z, v[u].w}, reinterpret_cast<v4*>(dst + i + u * stride));
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
// variant 2: HBM -> LDS by LDS-DMA (a 4-stage ring of 8 KiB stages per 512-thread block, counted waits), LDS -> registers -> HBM:
// the streaming structure of k_gat_mp_tiled without its arithmetic
__global__ __launch_bounds__(512) void k_stream_copy_dma(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 8192];
    const int tid = threadIdx.x;
    const unsigned lds_base = (unsigned)(size_t)(lds_bytes_t)smem;
    const unsigned wave_off = __builtin_amdgcn_readfirstlane((unsigned)(tid & ~63) * 16u);
    const size_t ntile = n4 / 512;                     // whole 8 KiB tiles; the tail is copied plainly below
    const size_t stride = gridDim.x;
    auto issue = [&](size_t tile, int slot) {
        const size_t tl = tile < ntile ? tile : ntile - 1;              // (clamped re-load past the end: uniform instruction counts)
        lds_dma16_b(src + tl * 512 + tid, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)slot * 8192u + wave_off));
    };
    size_t t0 = blockIdx.x;
    if (ntile > 0) {
        for (int k = 0; k < 3; ++k) issue(t0 + k * stride, k);
        int slot = 0;
        for (size_t t = t0; t < ntile; t += stride) {
            issue(t + 3 * stride, (slot + 3) & 3);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");          // this wave's DMA of tile t has landed (every lane reads only what its own wave loaded)
            const float4 v = *reinterpret_cast<const float4*>(smem + slot * 8192 + tid * 16);
            dst[t * 512 + tid] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            slot = (slot + 1) & 3;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (size_t i = ntile * 512 + (size_t)blockIdx.x * 512 + tid; i < n4; i += stride * 512) dst[i] = src[i];
}

}  // namespace gvqa

extern "C" {

int gvqa_stream_copy(void* dst, const void* src, size_t bytes, int variant, void* stream) {
    using namespace gvqa;
    GVQA_REQUIRE(dst && src && bytes % 16 == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0 && variant >= 0 && variant <= 2,
                 GVQA_E_INVALID, "gvqa_stream_copy: null / unaligned operand or unknown variant");
    if (bytes == 0) return GVQA_OK;
    const size_t n4 = bytes / 16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cus = device_cu_count();
    const float4* s4 = static_cast<const float4*>(src);
    float4* d4 = static_cast<float4*>(dst);
    if (variant == 2) {
        const unsigned grid = (unsigned)std::min<size_t>((size_t)cus * 4, std::max<size_t>(n4 / 512, 1));      // 4 blocks x 32 KiB of LDS per CU
        hipLaunchKernelGGL(k_stream_copy_dma, dim3(grid), dim3(512), 0, st, s4, d4, n4);
    } else {
        const unsigned grid = (unsigned)std::min<size_t>((size_t)cus * 8, std::max<size_t>(n4 / (256 * 4), 1));
        if (variant == 1) hipLaunchKernelGGL((k_stream_copy<4, true>), dim3(grid), dim3(256), 0, st, s4, d4, n4);
        else hipLaunchKernelGGL((k_stream_copy<4, false>), dim3(grid), dim3(256), 0, st, s4, d4, n4);
    }
    GVQA_LAUNCH_CHECK();
    return GVQA_OK;
}

const char* gvqa_last_error(void) { return gvqa::g_err; }

const char* gvqa_version(void) { return "gvqa-hip 0.1 gfx950"; }

int gvqa_set_option(int option, int value) {
    GVQA_REQUIRE(option >= 0 && option < GVQA_NUM_OPTIONS, GVQA_E_INVALID, "gvqa_set_option: unknown option %d", option);
    (void)gvqa::get_option(option);                   // environment defaults first
    gvqa::g_opt[option].store(value, std::memory_order_relaxed);
    return GVQA_OK;
}

int gvqa_get_option(int option) { return gvqa::get_option(option); }

int gvqa_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(gvqa::g_prof_mu);
    gvqa::g_prof_on = on != 0;
    gvqa::g_prof_mask = (on & 1) && (on >> 1) ? (unsigned)(on >> 1) : ~0u;
    return GVQA_OK;
}

int gvqa_prof_collect(double* ms_by_stage, int64_t* launches_by_stage) {
    using namespace gvqa;
    GVQA_REQUIRE(ms_by_stage && launches_by_stage, GVQA_E_INVALID, "gvqa_prof_collect: null output");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& ps : g_prof_live) {
        GVQA_HIP_CHECK(hipEventSynchronize(ps.stop));
        float ms = 0.f;
        GVQA_HIP_CHECK(hipEventElapsedTime(&ms, ps.start, ps.stop));
        if (ps.stage >= 0 && ps.stage < GVQA_NUM_STAGES) {
            ms_by_stage[ps.stage] += ms;
            launches_by_stage[ps.stage] += 1;
        }
        g_prof_free.push_back(ps);
    }
    g_prof_live.clear();
    return GVQA_OK;
}

}  // extern "C"
