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
