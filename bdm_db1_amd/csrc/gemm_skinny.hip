// Skinny bf16 GEMM for inference with Transformer-XL memory (evaluate_rl.py:157-266: 1 .. ~50 new tokens per call):
//     y[M, N] = alpha * x[M, K] . W[N, K]^T + beta * y + bias[n],     M <= 64, both operands K-major (the NT form of nn.Linear).
// With so few rows the product is a stream over W (HBM-bound: 2 N K bytes): the 128x128 tile kernels would run N / 128
// workgroups with an almost empty tile each, the generic strided kernel took 135-250 us per projection at M = 1.
// One workgroup = 16 output columns; its 4 waves split the contraction 4 ways (partials added in wave order through LDS:
// deterministic).  Lane (r = lane & 15, g = lane >> 4) reads 32 contiguous bytes of W row n0 + r per 64-wide k-step (the four
// lanes of a row cover one 128-byte line) and the same 32 bytes of each x row; the two 8-element halves are the two
// v_mfma_f32_16x16x32_bf16 k-steps (any k permutation is fine as long as x and W use the same one).  Operands are swapped in
// the MFMA like in the tile kernels, so a lane ends with 4 consecutive n of row m = lane & 15 and reuses store_frag.
#include "gemm_tile.h"

struct GemmSkinnyArgs {
    const bf16_t* x; const bf16_t* w; void* y; const void* bias;
    int M, N, K;
    int64_t ldx, ldw, ldy;
    float alpha, beta;
};

template <typename TC, typename TBIAS, int MT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmSkinnyArgs p) {
    __shared__ f32x4 red[3][MT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int ksteps = p.K / 64, per = (ksteps + 3) / 4;
    const int ks0 = wave * per, ks1 = ks0 + per < ksteps ? ks0 + per : ksteps;
    const bf16_t* wrow = p.w + (int64_t)(n0 + r) * p.ldw + g * 16;
    const bf16_t* xrow[MT];
    bool xok[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        xok[t] = t * 16 + r < p.M;
        xrow[t] = p.x + (int64_t)(xok[t] ? t * 16 + r : 0) * p.ldx + g * 16;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
    // The kernel is a latency problem (a workgroup streams only 64-256 KB of W): the W pieces of EIGHT k-steps (16 loads of 16 bytes per
    // lane, 32 KB per workgroup) are requested before the first one is used, so a wave's share of K = 2048 is ONE round trip instead of
    // two (unroll 4) -- and the loads are non-temporal: every byte of W is read once per call, by one workgroup.
    constexpr int KB = 8;
    for (int kb = ks0; kb < ks1; kb += KB) {
        bf16x8_t b0[KB], b1[KB];
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const bool ok = kb + u < ks1;
            b0[u] = ok ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + (kb + u) * 64)) : zero;
            b1[u] = ok ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + (kb + u) * 64 + 8)) : zero;
        }
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const bool ok = kb + u < ks1;
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const bf16x8_t a0 = (xok[t] && ok) ? *reinterpret_cast<const bf16x8_t*>(xrow[t] + (kb + u) * 64) : zero;
                const bf16x8_t a1 = (xok[t] && ok) ? *reinterpret_cast<const bf16x8_t*>(xrow[t] + (kb + u) * 64 + 8) : zero;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[u], a0, acc[t], 0, 0, 0);  // swapped: D[n][m]
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[u], a1, acc[t], 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; t++) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave == 0) {
        TC* Y = (TC*)p.y;
#pragma unroll
        for (int t = 0; t < MT; t++) {
            f32x4 s = acc[t];
#pragma unroll
            for (int w = 0; w < 3; w++) {
                const f32x4 v = red[w][t][lane];
                s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
            }
            const int m = t * 16 + r;
            if (m < p.M) store_frag<TC, TBIAS>(s, Y, p.ldy, m, n0 + g * 4, p.alpha, p.beta, p.bias);
        }
    }
}

template <int MT>
static void launch_skinny(const GemmSkinnyArgs& a, int dtC, int dtBias, hipStream_t st) {
    const dim3 grid((unsigned)(a.N / 16));
    if (dtC == DB1_F32) {
        if (dtBias == DB1_BF16) gemm_skinny_kernel<float, bf16_t, MT><<<grid, 256, 0, st>>>(a);
        else gemm_skinny_kernel<float, float, MT><<<grid, 256, 0, st>>>(a);
    } else {
        if (dtBias == DB1_BF16) gemm_skinny_kernel<bf16_t, bf16_t, MT><<<grid, 256, 0, st>>>(a);
        else gemm_skinny_kernel<bf16_t, float, MT><<<grid, 256, 0, st>>>(a);
    }
}

// caller has checked: bf16 operands, both K-major, M <= 64, N % 16 == 0, K % 64 == 0, 16-byte alignment, ld % 8 == 0
int db1_gemm_skinny_launch(const bf16_t* x, const bf16_t* w, void* y, const void* bias, int M, int N, int K, int64_t ldx, int64_t ldw,
                           int64_t ldy, float alpha, float beta, int dtC, int dtBias, hipStream_t st) {
    GemmSkinnyArgs a;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.alpha = alpha; a.beta = beta;
    if (M <= 16) launch_skinny<1>(a, dtC, dtBias, st);
    else if (M <= 32) launch_skinny<2>(a, dtC, dtBias, st);
    else if (M <= 48) launch_skinny<3>(a, dtC, dtBias, st);
    else launch_skinny<4>(a, dtC, dtBias, st);
    DB1_CHECK_LAUNCH("gemm_skinny");
    return DB1_OK;
}
