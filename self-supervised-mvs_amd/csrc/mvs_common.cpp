// Error reporting + version for the C ABI (include/mvs_hip.h).  No exceptions cross the boundary:
// every entry point returns 0 or a negative code and leaves a message in a thread-local buffer.
#include <stdarg.h>
#include <stdio.h>
#include "mvs_rt.h"

static thread_local char g_err[512] = "";

void mvs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mvs_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mvs_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MVS_ERR_LAUNCH;
    }
    return MVS_OK;
}

extern "C" const char* mvs_last_error(void) { return g_err; }
extern "C" int mvs_version(void) { return 100; }  // 0.1.0
extern "C" int mvs_is_emulation(void) {
#if defined(MVS_CPU_EMUL)
    return 1;
#else
    return 0;
#endif
}
