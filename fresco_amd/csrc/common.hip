#include "common.h"
#include <string.h>

namespace fresco {
static thread_local char g_last_error[256] = "";
void set_last_error(hipError_t e) {
    const char* s = hipGetErrorString(e);
    strncpy(g_last_error, s ? s : "unknown HIP error", sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}
}  // namespace fresco

extern "C" const char* fresco_version(void) { return "fresco_hip 0.1.0 gfx950"; }
extern "C" const char* fresco_last_error(void) { return fresco::g_last_error; }
