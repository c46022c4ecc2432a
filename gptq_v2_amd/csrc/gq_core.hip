// Error reporting and version for libgptq_amd.so.
#include <stdarg.h>

#include "gq_common.h"

static thread_local char g_last_error[512] = "";

void gq_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" const char *gq_last_error(void) { return g_last_error; }

extern "C" int gq_version(void) { return GQ_ABI_VERSION; }
