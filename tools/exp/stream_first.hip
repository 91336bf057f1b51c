// Time to the first rows: 256 workgroups x 8 waves; each wave requests 4 KB (its "W0" row), then 16 KB ("W1"), waits for the first 4 KB
// (vmcnt(16)), stamps, waits for all, stamps.  Cold buffers in turn.  EXTRA idle waves / LDS per workgroup as in db1_decode_chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int EXTRA, int LDSB> __global__ __launch_bounds__((8 + EXTRA) * 64, 1) void k(const u32x4* __restrict__ w0, const u32x4* __restrict__ w1, unsigned long long* ts, unsigned* out) {
    __shared__ char pad[LDSB > 0 ? LDSB : 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 8) { if (LDSB > 0 && threadIdx.x == 100000) pad[lane] = 1; return; }
    unsigned long long t0 = wall_clock64();
    const u32x4* a = w0 + ((size_t)blockIdx.x * 8 + wave) * 4 * 64 + lane;
    const u32x4* b = w1 + ((size_t)blockIdx.x * 8 + wave) * 16 * 64 + lane;
    u32x4 v0[4], v1[16];
#pragma unroll
    for (int i = 0; i < 4; i++) v0[i] = __builtin_nontemporal_load(a + i * 64);
#pragma unroll
    for (int i = 0; i < 16; i++) v1[i] = __builtin_nontemporal_load(b + i * 64);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    unsigned long long t2 = wall_clock64();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s += v0[i][0] ^ v0[i][3];
#pragma unroll
    for (int i = 0; i < 16; i++) s += v1[i][0] ^ v1[i][3];
    asm volatile("" :: "v"(s));
    unsigned long long t3 = wall_clock64();
    if (lane == 0 && wave == 0) { ts[blockIdx.x * 4] = t0; ts[blockIdx.x * 4 + 1] = t1; ts[blockIdx.x * 4 + 2] = t2; ts[blockIdx.x * 4 + 3] = t3; }
    if (s == 0x12345678u) out[0] = s;
}
template <int EXTRA, int LDSB> void run(const char* name, char* buf, size_t layer, unsigned long long* ts, unsigned* out) {
    std::vector<unsigned long long> h(1024);
    for (int l = 0; l < 6; l++) {
        const char* base = buf + (size_t)l * layer;
        hipDeviceSynchronize();
        k<EXTRA, LDSB><<<256, (8 + EXTRA) * 64>>>((const u32x4*)base, (const u32x4*)(base + (8u << 20)), ts, out);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), ts, 1024 * 8, hipMemcpyDeviceToHost);
        unsigned long long m = h[0]; for (int i = 0; i < 256; i++) m = std::min(m, h[i * 4]);
        double q[4][3];
        for (int kx = 0; kx < 4; kx++) { std::vector<double> v; for (int i = 0; i < 256; i++) v.push_back((h[i * 4 + kx] - m) / 100.0); std::sort(v.begin(), v.end()); q[kx][0] = v[0]; q[kx][1] = v[128]; q[kx][2] = v[255]; }
        if (l >= 2) printf("%-22s start %.2f/%.2f/%.2f  issued %.2f/%.2f/%.2f  first 4 KB %.2f/%.2f/%.2f  all 20 KB %.2f/%.2f/%.2f  (min/med/max us)\n", name, q[0][0], q[0][1], q[0][2], q[1][0], q[1][1], q[1][2], q[2][0], q[2][1], q[2][2], q[3][0], q[3][1], q[3][2]);
    }
}
int main() {
    const size_t layer = 96u << 20;
    char* buf; hipMalloc(&buf, layer * 6); hipMemset(buf, 1, layer * 6);
    unsigned long long* ts; hipMalloc(&ts, 1024 * 8); unsigned* out; hipMalloc(&out, 64);
    run<0, 0>("8 waves", buf, layer, ts, out);
    run<3, 0>("8 + 3 idle waves", buf, layer, ts, out);
    run<3, 98304>("8 + 3 idle, 96 KB LDS", buf, layer, ts, out);
    return 0;
}
