// Error reporting and library-level entry points of libl3d_hip.so.
#include "common.h"
#include <cstring>

namespace l3d {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

} // namespace l3d

extern "C" const char *l3d_last_error(void) { return l3d::g_err; }

extern "C" const char *l3d_version(void) { return "l3d-hip 0.1 (gfx950)"; }

extern "C" int l3d_device_count(int *n)
{
    L3D_REQUIRE(n, "l3d_device_count: null argument");
    *n = 0;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) {
        l3d::set_error("no HIP device available");
        return L3D_ERR_NO_DEVICE;
    }
    *n = c;
    return L3D_OK;
}
