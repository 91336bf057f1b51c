// How long does one hand-off between 256 resident workgroups take on gfx950?  (floor of a persistent multi-stage kernel's stage boundary)
//   mode 0: write-through stores, vmcnt(0), agent-scope counter add, spin on the counter, L1-bypassing loads of the row (db1_decode_chain today)
//   mode 1: every 32-bit word carries its own 16-bit tag; consumers poll the row itself until every tag is the round's (no counter, no store wait)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define WG 256
#define ROW 2048
__device__ __forceinline__ unsigned long long ld64(const void* p) { return __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st64(void* p, unsigned long long v) { __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE> __global__ __launch_bounds__(64, 1) void sync_bench(unsigned* rows, unsigned* ctr, int rounds, unsigned* out, int compute_ns) {
    const int lane = threadIdx.x, bid = blockIdx.x;
    unsigned acc = 0;
    for (int r = 0; r < rounds; r++) {
        unsigned* row = rows + (r % 3) * ROW;
        const unsigned tag = (unsigned)(r + 1) & 0xffffu;
        if (lane < 4) {   // this workgroup's 8 values (words bid * 8 .. + 7): 4 lanes x 8 bytes
            const unsigned w0 = ((acc + lane * 2) << 16) | tag, w1 = ((acc + lane * 2 + 1) << 16) | tag;
            st64(row + bid * 8 + lane * 2, ((unsigned long long)w1 << 32) | w0);
        }
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_fetch_add(ctr + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < WG) __builtin_amdgcn_s_sleep(1);
            }
            unsigned s = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) { const unsigned long long v = ld64(row + (k * 64 + lane) * 2); s += (unsigned)v + (unsigned)(v >> 32); }
            acc += s >> 16;
        } else {
            unsigned s;
            int spins = 0;
            while (true) {
                bool ok = true;
                s = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const unsigned long long v = ld64(row + (k * 64 + lane) * 2);
                    ok = ok && ((unsigned)v & 0xffffu) == tag && ((unsigned)(v >> 32) & 0xffffu) == tag;
                    s += (unsigned)v + (unsigned)(v >> 32);
                }
                if (__all(ok) || ++spins > (1 << 20)) break;
            }
            acc += s >> 16;
        }
        acc &= 0xff;
    }
    if (lane == 0) out[bid] = acc;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 96;
    unsigned *rows, *ctr, *out;
    hipMalloc(&rows, 3 * ROW * 4); hipMalloc(&ctr, rounds * 4); hipMalloc(&out, WG * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int it = 0; it < 5; it++) {
            hipMemset(rows, 0, 3 * ROW * 4); hipMemset(ctr, 0, rounds * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) sync_bench<0><<<WG, 64>>>(rows, ctr, rounds, out, 0); else sync_bench<1><<<WG, 64>>>(rows, ctr, rounds, out, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned h[WG]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            bool same = true; for (int i = 1; i < WG; i++) same = same && h[i] == h[0];
            printf("mode %d: %d rounds %.1f us  -> %.2f us per hand-off  (all workgroups agree: %d, acc %u)\n", mode, rounds, ms * 1e3, ms * 1e3 / rounds, (int)same, h[0]);
        }
    }
    return 0;
}
