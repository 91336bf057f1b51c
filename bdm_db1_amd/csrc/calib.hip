// Box calibration for bench.py: the sustained bf16 MFMA rate of THIS box with the operands in registers -- no memory
// traffic at all -- on operand bits that toggle like real activations.  The boxes of the pool differ by +-2-4 % in what they sustain under
// the 1.4 kW cap, which is as much as a round's gain: the bench line carries this number next to its tokens/s so that a
// delta between two records can be attributed to the box or to the code.  Declared in include/db1_hip_test.h (measurement tooling, not
// part of the drop-in surface).  The loop is tools/exp/mfma_shapes.hip's 16x16x32 case (profiles/r03_mfma_shapes.txt: 2205 TFLOP/s on the
// box it was first measured on, one wave per SIMD).
#include "db1_common.h"

typedef __attribute__((ext_vector_type(8))) short cal_bf16x8;

__device__ __forceinline__ cal_bf16x8 cal_operand(unsigned seed, int random_bits) {
    cal_bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        seed = seed * 1664525u + 1013904223u;
        v[j] = random_bits ? (short)(((seed >> 16) & 0x807F) | 0x3F00) : (short)0;   // random sign / mantissa around 1.0, or zeros
    }
    return v;
}

// (the loop body is inline asm on fixed registers: left to hipcc the 16 accumulator chains came out with copies and s_nops between the MFMAs)
__global__ __launch_bounds__(256) void db1_mfma_calibration_kernel(float* sink, int iters, int random_bits) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    union { cal_bf16x8 v; int w[4]; } a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i].v = cal_operand(tid * 8 + i, random_bits); b[i].v = cal_operand(tid * 8 + 4 + i, random_bits); }
    asm volatile(
        "v_mov_b32 v64, %0\n v_mov_b32 v65, %1\n v_mov_b32 v66, %2\n v_mov_b32 v67, %3\n"
        "v_mov_b32 v68, %4\n v_mov_b32 v69, %5\n v_mov_b32 v70, %6\n v_mov_b32 v71, %7\n"
        "v_mov_b32 v72, %8\n v_mov_b32 v73, %9\n v_mov_b32 v74, %10\n v_mov_b32 v75, %11\n"
        "v_mov_b32 v76, %12\n v_mov_b32 v77, %13\n v_mov_b32 v78, %14\n v_mov_b32 v79, %15\n"
        :: "v"(a[0].w[0]), "v"(a[0].w[1]), "v"(a[0].w[2]), "v"(a[0].w[3]), "v"(a[1].w[0]), "v"(a[1].w[1]), "v"(a[1].w[2]), "v"(a[1].w[3]),
           "v"(a[2].w[0]), "v"(a[2].w[1]), "v"(a[2].w[2]), "v"(a[2].w[3]), "v"(a[3].w[0]), "v"(a[3].w[1]), "v"(a[3].w[2]), "v"(a[3].w[3])
        : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79");
    asm volatile(
        "v_mov_b32 v80, %0\n v_mov_b32 v81, %1\n v_mov_b32 v82, %2\n v_mov_b32 v83, %3\n"
        "v_mov_b32 v84, %4\n v_mov_b32 v85, %5\n v_mov_b32 v86, %6\n v_mov_b32 v87, %7\n"
        "v_mov_b32 v88, %8\n v_mov_b32 v89, %9\n v_mov_b32 v90, %10\n v_mov_b32 v91, %11\n"
        "v_mov_b32 v92, %12\n v_mov_b32 v93, %13\n v_mov_b32 v94, %14\n v_mov_b32 v95, %15\n"
        :: "v"(b[0].w[0]), "v"(b[0].w[1]), "v"(b[0].w[2]), "v"(b[0].w[3]), "v"(b[1].w[0]), "v"(b[1].w[1]), "v"(b[1].w[2]), "v"(b[1].w[3]),
           "v"(b[2].w[0]), "v"(b[2].w[1]), "v"(b[2].w[2]), "v"(b[2].w[3]), "v"(b[3].w[0]), "v"(b[3].w[1]), "v"(b[3].w[2]), "v"(b[3].w[3])
        : "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95");
#define CAL_Z(n) "v_accvgpr_write_b32 a" #n ", 0\n"
    asm volatile(CAL_Z(0) CAL_Z(1) CAL_Z(2) CAL_Z(3) CAL_Z(4) CAL_Z(5) CAL_Z(6) CAL_Z(7) CAL_Z(8) CAL_Z(9) CAL_Z(10) CAL_Z(11) CAL_Z(12) CAL_Z(13) CAL_Z(14) CAL_Z(15)
                 CAL_Z(16) CAL_Z(17) CAL_Z(18) CAL_Z(19) CAL_Z(20) CAL_Z(21) CAL_Z(22) CAL_Z(23) CAL_Z(24) CAL_Z(25) CAL_Z(26) CAL_Z(27) CAL_Z(28) CAL_Z(29) CAL_Z(30) CAL_Z(31)
                 CAL_Z(32) CAL_Z(33) CAL_Z(34) CAL_Z(35) CAL_Z(36) CAL_Z(37) CAL_Z(38) CAL_Z(39) CAL_Z(40) CAL_Z(41) CAL_Z(42) CAL_Z(43) CAL_Z(44) CAL_Z(45) CAL_Z(46) CAL_Z(47)
                 CAL_Z(48) CAL_Z(49) CAL_Z(50) CAL_Z(51) CAL_Z(52) CAL_Z(53) CAL_Z(54) CAL_Z(55) CAL_Z(56) CAL_Z(57) CAL_Z(58) CAL_Z(59) CAL_Z(60) CAL_Z(61) CAL_Z(62) CAL_Z(63)
                 ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23",
                     "a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47",
                     "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63");
    for (int it = 0; it < iters; it++) {
        // 16 independent 16x16x32 MFMAs: accumulator a[4c:4c+3], operands v[64 + 4i ..] and v[80 + 4j ..]
#define CAL_M(c, i, j) "v_mfma_f32_16x16x32_bf16 a[" #c ":" #c "+3], v[80+4*" #j ":80+4*" #j "+3], v[64+4*" #i ":64+4*" #i "+3], a[" #c ":" #c "+3]\n"
        asm volatile(CAL_M(0,0,0) CAL_M(4,0,1) CAL_M(8,0,2) CAL_M(12,0,3) CAL_M(16,1,0) CAL_M(20,1,1) CAL_M(24,1,2) CAL_M(28,1,3)
                     CAL_M(32,2,0) CAL_M(36,2,1) CAL_M(40,2,2) CAL_M(44,2,3) CAL_M(48,3,0) CAL_M(52,3,1) CAL_M(56,3,2) CAL_M(60,3,3) ::: "memory");
    }
    float s;
    asm volatile("s_nop 7\n s_nop 7\n v_accvgpr_read_b32 %0, a0" : "=v"(s));
    if (s == 12345.678f) sink[tid] = s;     // (never true: keeps the loop alive)
}

// `iters` rounds of 16 MFMAs per wave on 256 workgroups x 4 waves (one wave per SIMD of every CU): 256 * 4 * iters * 16 * 16 384 FLOP.
// Asynchronous on `stream`; the caller times it with events.  sink: >= 256 * 256 floats (never written).
extern "C" int db1_test_mfma_calibration(float* sink, int iters, int random_bits, void* stream) {
    if (!sink || iters <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "mfma_calibration: sink / iters");
    db1_mfma_calibration_kernel<<<256, 256, 0, (hipStream_t)stream>>>(sink, iters, random_bits);
    DB1_CHECK_LAUNCH("mfma_calibration");
    return DB1_OK;
}
