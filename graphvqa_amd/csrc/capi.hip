// Error reporting, version string and in-library stage timing for libgvqa_hip.so.
#include <stdarg.h>

#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

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
}
int get_option(int option) {
    std::call_once(g_opt_once, options_init);
    return (option >= 0 && option < GVQA_NUM_OPTIONS) ? g_opt[option].load(std::memory_order_relaxed) : 0;
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

}  // namespace gvqa

extern "C" {

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
