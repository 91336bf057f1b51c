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

// ---- thread-local A/B knobs (test header only)
static thread_local int g_knob_val[DB1_KNOB_COUNT];
static thread_local bool g_knob_set[DB1_KNOB_COUNT];
static const char* const g_knob_names[DB1_KNOB_COUNT] = {"gemm_tile", "gemm_splitk", "pp32_stages", "linear_decode_splitk", "w4",
                                                         "flash_fwd2", "flash_kv3", "conv_wgrad_ks", "geglu_epi", "gemm_halfwave", "w4n", "conv_patch", "tri_split"};
int db1_knob(int id, int dflt) { return (id >= 0 && id < DB1_KNOB_COUNT && g_knob_set[id]) ? g_knob_val[id] : dflt; }
extern "C" int db1_test_set_knob(const char* name, int value) {
    for (int i = 0; i < DB1_KNOB_COUNT; i++)
        if (name && strcmp(name, g_knob_names[i]) == 0) { g_knob_val[i] = value; g_knob_set[i] = true; return 0; }
    db1_set_error("db1_test_set_knob: unknown knob '%s'", name ? name : "(null)");
    return DB1_ERR_BAD_SHAPE;
}
extern "C" void db1_test_clear_knobs(void) { for (int i = 0; i < DB1_KNOB_COUNT; i++) g_knob_set[i] = false; }

extern "C" int db1_version(void) { return 100; }
// 1 when the library was compiled with -DDB1_EXPERIMENT (timing ablations that produce WRONG results by construction, tools/exp): the
// Python loader refuses such a build unless DB1_ALLOW_EXPERIMENT=1 is set by the experiment script itself
extern "C" int db1_is_experiment_build(void) {
#ifdef DB1_EXPERIMENT
    return 1;
#else
    return 0;
#endif
}
extern "C" const char* db1_last_error(void) { return g_err; }

extern "C" int db1_device_is_gfx950(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { db1_set_error("hipGetDevice failed"); return 0; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { db1_set_error("hipGetDeviceProperties failed"); return 0; }
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ---- trace marker (test header): tools/prof_table.py cuts a rocprofv3 kernel trace to the dispatches between two of these
__global__ void db1_marker_kernel(int tag) { (void)tag; }
extern "C" int db1_test_marker(int tag, void* stream) {
    db1_marker_kernel<<<1, 64, 0, (hipStream_t)stream>>>(tag);
    DB1_CHECK_LAUNCH("db1_marker");
    return DB1_OK;
}
