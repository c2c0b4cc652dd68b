// Error reporting, version string and in-library stage timing for libgvqa_hip.so.
#include <stdarg.h>

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

// ---- stage timing -------------------------------------------------------------------------
struct ProfSlot {
    int stage;
    hipEvent_t start, stop;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfSlot> g_prof_live;     // recorded, not yet collected
static std::vector<ProfSlot> g_prof_free;     // event pairs available for reuse

StageTimer::StageTimer(int st, hipStream_t s) : stage(st), stream(s), slot(nullptr) {
    if (!g_prof_on) return;
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

int gvqa_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(gvqa::g_prof_mu);
    gvqa::g_prof_on = on != 0;
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
