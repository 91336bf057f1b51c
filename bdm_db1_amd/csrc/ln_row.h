// One LayerNorm row held in the registers of ONE wave (d == 64 * V * NV): y = LN(alpha * x + r) * gamma + beta, the statistics in fp32 over the
// row as the reference stores it (rounded to T).  Shared by the row kernel (elementwise.hip: ln_fwd_reg_kernel, which adds dropout on r and
// keeps s / mean / rstd for the backward) and by the inference linear layer that finishes with the LayerNorm of its rows (gemm_skinny.hip).
#pragma once
#include "db1_common.h"

template <typename T, int NV>
__device__ __forceinline__ void ln_row_stats(Vec16<T> (&a)[NV], int d, float eps, float& mu, float& rs) {
    constexpr int V = Vec16<T>::N;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int j = 0; j < V; j++) {
            if (sizeof(T) == 2) a[k].v[j] = bf2f(f2bf(a[k].v[j]));  // s is a tensor of dtype T in the reference
            sum += a[k].v[j];
        }
    mu = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int j = 0; j < V; j++) { const float c = a[k].v[j] - mu; sq += c * c; }
    rs = rsqrtf(wave_sum(sq) / (float)d + eps);
}

template <typename T, typename TP, int NV>
__device__ __forceinline__ void ln_row_store(const Vec16<T> (&a)[NV], float mu, float rs, const TP* __restrict__ gamma, const TP* __restrict__ beta,
                                             T* __restrict__ yrow, int lane) {
    constexpr int V = Vec16<T>::N;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int i = (k * 64 + lane) * V;
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < V; j++) o.v[j] = (a[k].v[j] - mu) * rs * ldf(gamma + i + j) + ldf(beta + i + j);
        o.store(yrow + i);
    }
}
