// Error channel + device query of libdb1_hip.so.
#include "db1_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void db1_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int db1_version(void) { return 100; }
extern "C" const char* db1_last_error(void) { return g_err; }

extern "C" int db1_device_is_gfx950(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { db1_set_error("hipGetDevice failed"); return 0; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { db1_set_error("hipGetDeviceProperties failed"); return 0; }
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
