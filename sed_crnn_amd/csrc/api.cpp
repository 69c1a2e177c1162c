// api.cpp — library-level entry points: version, thread-local error string, measurement timers.
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";

void sed_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sed_version(void) { return 100; }  /* 0.1.0 */
extern "C" const char* sed_last_error_string(void) { return g_err; }

// ── kernel timers ──
unsigned g_sed_prof_mask = 0;
namespace {
struct Rec { int tag; hipEvent_t a, b; double units; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
std::mutex g_prof_mu;                  // the timers are a measurement aid, but entries may be called from several threads
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
const char* kNames[SED_K_COUNT] = {"conv3x3_mfma_fwd", "conv3x3_small_fwd", "conv3x3_mfma_wgrad", "conv3x3_small_wgrad",
                                   "bn_relu_pool_drop_fwd", "bn_bwd_reduce", "bn_bwd_apply", "gemm_f32", "gru_seq_fwd",
                                   "gru_seq_bwd", "adam", "logmel", "conv3x3_mfma_dgrad_bnred"};
}  // namespace

void sed_prof_begin(int tag, hipStream_t s, double units) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    Rec r{tag, get_event(), get_event(), units};
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
}
void sed_prof_end(int tag, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = g_recs.size(); i-- > 0;)
        if (g_recs[i].tag == tag) { (void)hipEventRecord(g_recs[i].b, s); return; }
}

extern "C" int sed_prof_enable(unsigned mask) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    g_sed_prof_mask = mask;
    return 0;
}

extern "C" int sed_prof_read(int tag, double* total_ms, long* launches, double* total_units) {
    SED_REQUIRE(tag >= 0 && tag < SED_K_COUNT, "prof_read: bad tag %d", tag);
    double ms = 0, units = 0;
    long n = 0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_recs) {
        if (r.tag != tag) continue;
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) { sed_set_error("prof_read: %s", hipGetErrorString(e)); return (int)e; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        ms += t; units += r.units; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (total_units) *total_units = units;
    return 0;
}

extern "C" int sed_prof_tag_count(void) { return SED_K_COUNT; }
extern "C" const char* sed_prof_tag_name(int tag) { return (tag >= 0 && tag < SED_K_COUNT) ? kNames[tag] : "?"; }
