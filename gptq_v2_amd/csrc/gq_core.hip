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

extern "C" int gq_device_info(int *compute_units, int *max_clock_khz)
{
    GQ_REQUIRE(compute_units && max_clock_khz, "gq_device_info: NULL pointer");
    int dev = 0;
    GQ_HIP(hipGetDevice(&dev));
    GQ_HIP(hipDeviceGetAttribute(compute_units, hipDeviceAttributeMultiprocessorCount, dev));
    GQ_HIP(hipDeviceGetAttribute(max_clock_khz, hipDeviceAttributeClockRate, dev));
    return GQ_OK;
}
