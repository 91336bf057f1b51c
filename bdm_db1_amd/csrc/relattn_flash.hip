// Relative-position flash attention (bf16, d_head = 128) -- placeholder entry points until the fused
// kernels land; db1_relattn_flash_supported() returns 0 so callers use the materialised path.
#include "db1_common.h"

extern "C" int db1_relattn_flash_supported(int B, int L, int H, int D, int dt) {
    (void)B; (void)L; (void)H; (void)D; (void)dt;
    return 0;
}
extern "C" int db1_relattn_flash_fwd(const void*, const void*, const void*, const void*, void*, float*, int, int, int, int, int, float, int, void*) {
    DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_flash_fwd: not built in this version");
}
extern "C" int db1_relattn_flash_bwd(const void*, const void*, const void*, const void*, const void*, const void*, const float*, float*, void*, void*,
                                     int, int, int, int, int, float, int, void*) {
    DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_flash_bwd: not built in this version");
}
