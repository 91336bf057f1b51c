// Argument block of the strided GEMM kernel (shared by gemm.hip and gemm_strided.hip).
#pragma once
#include <stdint.h>
struct GemmStridedArgs {
    const void* A; const void* B; void* C; const void* bias;
    int M, N, K;
    int64_t a_rs, a_cs, b_rs, b_cs, c_rs, c_cs;
    int batch1;
    int64_t a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;
    float alpha, beta;
};
