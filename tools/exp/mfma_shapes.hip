// Sustained bf16 MFMA rate under the power cap: v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16, operands in registers
// (no memory traffic at all), N(0,1)-like operand bits vs zeros.  One wave per SIMD (4 per CU), 256 CUs, each wave issues independent
// accumulator chains back to back for ~0.5 s while rocm-smi style clocks can be sampled from outside.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
// Question: would the 32x32x16 form -- half the operand-register reads per FLOP -- run faster at the 1.4 kW cap?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ bf16x8 operand(unsigned seed, int zero) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        seed = seed * 1664525u + 1013904223u;
        // a bf16 with random sign / mantissa and an exponent around 1.0: bit activity like real activations
        v[j] = zero ? (short)0 : (short)(((seed >> 16) & 0x807F) | 0x3F00);
    }
    return v;
}

// (the loop bodies are inline asm on fixed registers: left to hipcc, the 16x16 form came out with accumulator copies and s_nops between the MFMAs)
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_kernel(float* sink, int iters, int zero) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = operand(tid * 8 + i, zero); b[i] = operand(tid * 8 + 4 + i, zero); }
    // accumulators a[0:63] zeroed; operands in v[64:79] (A) and v[80:95] (B)
    asm volatile(
        "v_mov_b32 v64, %0\n v_mov_b32 v65, %1\n v_mov_b32 v66, %2\n v_mov_b32 v67, %3\n"
        "v_mov_b32 v68, %4\n v_mov_b32 v69, %5\n v_mov_b32 v70, %6\n v_mov_b32 v71, %7\n"
        "v_mov_b32 v72, %8\n v_mov_b32 v73, %9\n v_mov_b32 v74, %10\n v_mov_b32 v75, %11\n"
        "v_mov_b32 v76, %12\n v_mov_b32 v77, %13\n v_mov_b32 v78, %14\n v_mov_b32 v79, %15\n"
        :: "v"(((int*)&a[0])[0]), "v"(((int*)&a[0])[1]), "v"(((int*)&a[0])[2]), "v"(((int*)&a[0])[3]),
           "v"(((int*)&a[1])[0]), "v"(((int*)&a[1])[1]), "v"(((int*)&a[1])[2]), "v"(((int*)&a[1])[3]),
           "v"(((int*)&a[2])[0]), "v"(((int*)&a[2])[1]), "v"(((int*)&a[2])[2]), "v"(((int*)&a[2])[3]),
           "v"(((int*)&a[3])[0]), "v"(((int*)&a[3])[1]), "v"(((int*)&a[3])[2]), "v"(((int*)&a[3])[3])
        : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79");
    asm volatile(
        "v_mov_b32 v80, %0\n v_mov_b32 v81, %1\n v_mov_b32 v82, %2\n v_mov_b32 v83, %3\n"
        "v_mov_b32 v84, %4\n v_mov_b32 v85, %5\n v_mov_b32 v86, %6\n v_mov_b32 v87, %7\n"
        "v_mov_b32 v88, %8\n v_mov_b32 v89, %9\n v_mov_b32 v90, %10\n v_mov_b32 v91, %11\n"
        "v_mov_b32 v92, %12\n v_mov_b32 v93, %13\n v_mov_b32 v94, %14\n v_mov_b32 v95, %15\n"
        :: "v"(((int*)&b[0])[0]), "v"(((int*)&b[0])[1]), "v"(((int*)&b[0])[2]), "v"(((int*)&b[0])[3]),
           "v"(((int*)&b[1])[0]), "v"(((int*)&b[1])[1]), "v"(((int*)&b[1])[2]), "v"(((int*)&b[1])[3]),
           "v"(((int*)&b[2])[0]), "v"(((int*)&b[2])[1]), "v"(((int*)&b[2])[2]), "v"(((int*)&b[2])[3]),
           "v"(((int*)&b[3])[0]), "v"(((int*)&b[3])[1]), "v"(((int*)&b[3])[2]), "v"(((int*)&b[3])[3])
        : "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95");
#define Z4(n) "v_accvgpr_write_b32 a" #n ", 0\n"
    asm volatile(Z4(0) Z4(1) Z4(2) Z4(3) Z4(4) Z4(5) Z4(6) Z4(7) Z4(8) Z4(9) Z4(10) Z4(11) Z4(12) Z4(13) Z4(14) Z4(15)
                 Z4(16) Z4(17) Z4(18) Z4(19) Z4(20) Z4(21) Z4(22) Z4(23) Z4(24) Z4(25) Z4(26) Z4(27) Z4(28) Z4(29) Z4(30) Z4(31)
                 Z4(32) Z4(33) Z4(34) Z4(35) Z4(36) Z4(37) Z4(38) Z4(39) Z4(40) Z4(41) Z4(42) Z4(43) Z4(44) Z4(45) Z4(46) Z4(47)
                 Z4(48) Z4(49) Z4(50) Z4(51) Z4(52) Z4(53) Z4(54) Z4(55) Z4(56) Z4(57) Z4(58) Z4(59) Z4(60) Z4(61) Z4(62) Z4(63)
                 ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23",
                     "a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47",
                     "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63");
    for (int it = 0; it < iters; it++) {
        if (SHAPE == 16) {
            // 16 independent 16x16x32 MFMAs: acc a[4c:4c+3], A = v[64 + 4i ..], B = v[80 + 4j ..]
#define M16(c, i, j) "v_mfma_f32_16x16x32_bf16 a[" #c ":" #c "+3], v[80+4*" #j ":80+4*" #j "+3], v[64+4*" #i ":64+4*" #i "+3], a[" #c ":" #c "+3]\n"
            asm volatile(M16(0,0,0) M16(4,0,1) M16(8,0,2) M16(12,0,3) M16(16,1,0) M16(20,1,1) M16(24,1,2) M16(28,1,3)
                         M16(32,2,0) M16(36,2,1) M16(40,2,2) M16(44,2,3) M16(48,3,0) M16(52,3,1) M16(56,3,2) M16(60,3,3) ::: "memory");
        } else {
            // 8 x 32x32x16 (same FLOPs): 4 accumulators a[16c:16c+15], two k-steps each
#define M32(c, i, j) "v_mfma_f32_32x32x16_bf16 a[" #c ":" #c "+15], v[80+4*" #j ":80+4*" #j "+3], v[64+4*" #i ":64+4*" #i "+3], a[" #c ":" #c "+15]\n"
            asm volatile(M32(0,0,0) M32(16,0,1) M32(32,1,0) M32(48,1,1) M32(0,2,2) M32(16,2,3) M32(32,3,2) M32(48,3,3) ::: "memory");
        }
    }
    float s;
    asm volatile("s_nop 7\n s_nop 7\n v_accvgpr_read_b32 %0, a0" : "=v"(s));
    if (s == 12345.678f) sink[tid] = s;
}

int main() {
    float* sink; hipMalloc(&sink, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 2;     // two workgroups of 4 waves per CU: 2 waves per SIMD (like the 8-wave kernels); 1 per SIMD: grid 256
    for (int occ = 1; occ <= 2; occ++)
        for (int zero = 0; zero < 2; zero++)
            for (int shape = 16; shape <= 32; shape += 16) {
                const int iters = 400000;
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(e0);
                    if (shape == 16) mfma_kernel<16><<<256 * occ, 256>>>(sink, iters, zero);
                    else mfma_kernel<32><<<256 * occ, 256>>>(sink, iters, zero);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double flops = (double)256 * occ * 4 * iters * 16 * 16384.0;
                printf("%d wave(s)/SIMD  %-7s  v_mfma_f32_%s_bf16: %7.1f ms  %7.1f TFLOP/s\n", occ, zero ? "zeros" : "random", shape == 16 ? "16x16x32" : "32x32x16",
                       best, flops / best / 1e9);
            }
    (void)grid;
    return 0;
}
