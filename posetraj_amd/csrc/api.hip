// Library-level pieces of the C ABI: error reporting, the zero page, hipEvent profiling hooks.
#include <stdarg.h>

#include <vector>

#include "pt_common.h"

namespace {
thread_local char g_err[512] = "";
constexpr int PT_MAX_DEVICES = 64;
const void* g_zero_page[PT_MAX_DEVICES] = {};       // one zero page per device (set by the process that owns the device)

struct ProfRec { hipEvent_t a, b; double flops; };
bool g_prof_on = false;
std::vector<ProfRec> g_open[PT_PROF_FAMILIES];       // recorded, not yet collected
std::vector<hipEvent_t> g_pool;
hipEvent_t g_cur_start[PT_PROF_FAMILIES];

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void pt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int pt_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return (d >= 0 && d < PT_MAX_DEVICES) ? d : 0;
}

const void* pt_zero_page() { return g_zero_page[pt_device()]; }

void pt_prof_begin(int family, hipStream_t s, double flops) {
    if (!g_prof_on) return;
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, s);
    g_cur_start[family] = e;
    g_open[family].push_back({e, nullptr, flops});
}

void pt_prof_end(int family, hipStream_t s) {
    if (!g_prof_on) return;
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, s);
    g_open[family].back().b = e;
}

extern "C" int pt_abi_version(void) { return PT_ABI_VERSION; }

extern "C" const char* pt_last_error(void) { return g_err; }

extern "C" int pt_set_zero_page(const void* dev_zeros_256B) {
    PT_CHECK(dev_zeros_256B && ((uintptr_t)dev_zeros_256B & 15) == 0, "pt_set_zero_page: need a 16-byte aligned device buffer");
    g_zero_page[pt_device()] = dev_zeros_256B;        // registered for the CURRENT device
    return 0;
}

extern "C" int pt_prof_enable(int32_t on) {
    g_prof_on = on != 0;
    return 0;
}

extern "C" int pt_prof_collect(int32_t family, int64_t* launches, double* ms, double* flops) {
    PT_CHECK(family >= 0 && family < PT_PROF_FAMILIES && launches && ms && flops, "pt_prof_collect: bad arguments");
    *launches = 0; *ms = 0.0; *flops = 0.0;
    for (ProfRec& r : g_open[family]) {
        if (!r.b) continue;
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { *launches += 1; *ms += t; *flops += r.flops; }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_open[family].clear();
    return 0;
}

// per-launch variant: fills ms[i], flops[i] for up to cap launches (in launch order) and clears the family's records
extern "C" int64_t pt_prof_collect_list(int32_t family, double* ms, double* flops, int64_t cap) {
    if (family < 0 || family >= PT_PROF_FAMILIES || !ms || !flops) return -1;
    int64_t n = 0;
    for (ProfRec& r : g_open[family]) {
        if (!r.b) continue;
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && n < cap) { ms[n] = t; flops[n] = r.flops; ++n; }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_open[family].clear();
    return n;
}
