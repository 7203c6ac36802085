// Error reporting shared by every translation unit of libbdetr.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void bdetr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bdetr_last_error(void) { return g_err; }
extern "C" int bdetr_abi_version(void) { return BDETR_ABI_VERSION; }

// A non-blocking HIP stream of the LOWEST priority the device offers: the host runs the weight-gradient
// GEMMs on it so that the workgroup dispatcher serves the critical path (the caller's stream) first and
// lets the side work fill what is left.  torch.cuda.Stream only exposes the normal and high classes.
extern "C" int bdetr_low_priority_stream_create(void** out) {
    BDETR_CHECK_ARG(out != nullptr, "bdetr_low_priority_stream_create: null out");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e == hipSuccess) {
        hipStream_t s = nullptr;
        e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
        if (e == hipSuccess) { *out = (void*)s; return 0; }
    }
    bdetr_set_error("bdetr_low_priority_stream_create: %s", hipGetErrorString(e));
    return (int)e;
}
extern "C" int bdetr_stream_priority_range(int* least, int* greatest) {
    hipError_t e = hipDeviceGetStreamPriorityRange(least, greatest);
    if (e != hipSuccess) { bdetr_set_error("bdetr_stream_priority_range: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}
