#include "common.h"
#include <string.h>

namespace fresco {
static thread_local char g_last_error[256] = "";
void set_last_error(hipError_t e) {
    const char* s = hipGetErrorString(e);
    strncpy(g_last_error, s ? s : "unknown HIP error", sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}

// ---- opt-in profiler: a fixed pool of event pairs, filled in launch order -------------------------
struct ProfRec {
    hipEvent_t e0, e1;
    int tag, dims[4];
};
static ProfRec* g_prof = nullptr;
static int g_prof_cap = 0, g_prof_n = 0;
static bool g_prof_on = false;

bool prof_active() { return g_prof_on; }

ProfScope::ProfScope(int tag, int a, int b, int c, int d, hipStream_t s) : on(false), st(s) {
    if (!g_prof_on || g_prof_n >= g_prof_cap) return;
    ProfRec& r = g_prof[g_prof_n];
    r.tag = tag;
    r.dims[0] = a;
    r.dims[1] = b;
    r.dims[2] = c;
    r.dims[3] = d;
    (void)hipEventRecord(r.e0, st);
    on = true;
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(g_prof[g_prof_n].e1, st);
    ++g_prof_n;
}
}  // namespace fresco

extern "C" int fresco_prof_enable(int capacity) {
    using namespace fresco;
    if (capacity <= 0) return FRESCO_EINVAL;
    if (g_prof_cap < capacity) {
        ProfRec* n = new ProfRec[capacity];
        for (int i = 0; i < capacity; ++i) {
            if (i < g_prof_cap) {
                n[i] = g_prof[i];
            } else {
                (void)hipEventCreate(&n[i].e0);
                (void)hipEventCreate(&n[i].e1);
            }
        }
        delete[] g_prof;
        g_prof = n;
        g_prof_cap = capacity;
    }
    g_prof_n = 0;
    g_prof_on = true;
    return FRESCO_OK;
}

extern "C" int fresco_prof_disable(void) {
    fresco::g_prof_on = false;
    return FRESCO_OK;
}

extern "C" int fresco_prof_read(int max_records, int* tags, int* dims, float* ms) {
    using namespace fresco;
    int n = g_prof_n < max_records ? g_prof_n : max_records;
    for (int i = 0; i < n; ++i) {
        (void)hipEventSynchronize(g_prof[i].e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_prof[i].e0, g_prof[i].e1);
        if (tags) tags[i] = g_prof[i].tag;
        if (dims)
            for (int j = 0; j < 4; ++j) dims[i * 4 + j] = g_prof[i].dims[j];
        if (ms) ms[i] = t;
    }
    g_prof_n = 0;
    return n;
}

extern "C" const char* fresco_version(void) { return "fresco_hip 0.1.0 gfx950"; }
extern "C" const char* fresco_last_error(void) { return fresco::g_last_error; }
