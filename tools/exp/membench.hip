// HBM bandwidth vs contiguity on gfx950: every wave reads (or writes) chunks of C contiguous bytes at scattered places of a big buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/membench.hip -o tools/exp/membench && tools/exp/membench
// Result (profiles/r02d_membench.txt) decides the layout of the tensors the attention backward streams through HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// chunk id -> scattered chunk slot (odd multiplier modulo a power of two is a bijection)
__device__ __forceinline__ uint64_t scatter(uint64_t i, uint64_t nchunks, int mode) {
    return mode == 0 ? i : (i * 2654435761ull + 12345ull) & (nchunks - 1);
}

template <int WRITE>
__global__ __launch_bounds__(512) void chunk_kernel(u32x4* buf, uint64_t nchunks, int chunk_kb, int mode, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t nwaves = (uint64_t)gridDim.x * 8, wid = (uint64_t)blockIdx.x * 8 + wave;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t c = wid; c < nchunks; c += nwaves) {
        u32x4* p = buf + scatter(c, nchunks, mode) * (uint64_t)(chunk_kb * 64) + lane;
#pragma unroll 4
        for (int k = 0; k < chunk_kb; k++) {
            if (WRITE) { u32x4 v = {(unsigned)c, (unsigned)k, (unsigned)lane, 7u}; p[k * 64] = v; }
            else { u32x4 v = __builtin_nontemporal_load(p + k * 64); acc ^= v; }
        }
    }
    if (!WRITE && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
    const uint64_t bytes = 4ull << 30;
    u32x4* buf; unsigned* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int kbs[] = {1, 2, 4, 8, 16, 64, 256};
    for (int wr = 0; wr < 2; wr++)
        for (int mode = 0; mode < 2; mode++)
            for (int ki = 0; ki < 7; ki++) {
                const int kb = kbs[ki];
                const uint64_t nchunks = bytes / ((uint64_t)kb * 1024);
                float best = 1e9f;
                for (int rep = 0; rep < 4; rep++) {
                    hipEventRecord(e0);
                    if (wr) chunk_kernel<1><<<2048, 512>>>(buf, nchunks, kb, mode, sink);
                    else chunk_kernel<0><<<2048, 512>>>(buf, nchunks, kb, mode, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                printf("%s %-9s chunk %4d KiB per wave: %8.1f GB/s\n", wr ? "write" : "read ", mode ? "scattered" : "linear", kb, bytes / best / 1e6);
            }
    return 0;
}
