// Fully strided, batched GEMM on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//   C[z][m,n] = alpha * sum_k A[z][m,k] B[z][k,n] + beta * C[z][m,n] + bias[n]
// This is the fp32 parity-gate instantiation and the catch-all for shapes the bf16 tile kernels do
// not take (tiny models, head sizes != 128, unaligned vocabularies, score-matrix contractions of the
// materialised attention path).  bf16 operands are widened to fp32 in LDS, so products are exact and
// accumulation is a k-ordered fp32 fma chain (MI355X_MICROARCH.md: f32-input MFMA == fmaf chain).
// Tile: 64x64x16 per 256-thread workgroup, 2x2 waves, each wave 32x32 = 2x2 MFMA fragments.
// LDS tiles are k-major ([k][m], row stride 80 floats) so fragment reads are conflict-free ds_read_b32.
#include "db1_common.h"
#include "gemm_args.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GS_BM 64
#define GS_BN 64
#define GS_BK 16
#define GS_LD 80  // 64 + 16: rows 16 lanes apart land on disjoint bank halves


template <typename TA, typename TB, typename TC, typename TBIAS>
__global__ __launch_bounds__(256) void gemm_strided_kernel(GemmStridedArgs p) {
    __shared__ float As[GS_BK * GS_LD];
    __shared__ float Bs[GS_BK * GS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z, z0 = z / p.batch1, z1 = z % p.batch1;
    const TA* A = (const TA*)p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const TB* B = (const TB*)p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    TC* C = (TC*)p.C + z0 * p.c_bs0 + z1 * p.c_bs1;
    const int m0 = blockIdx.x * GS_BM, n0 = blockIdx.y * GS_BN;  // M on grid.x: the conv1 GEMM of a large RL batch has millions of rows
    // loader index maps: pick the one whose fastest index follows the unit stride
    const bool a_kfast = (p.a_cs == 1);            // k contiguous in memory
    const bool b_kfast = (p.b_rs == 1 && p.b_cs != 1);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < p.K; k0 += GS_BK) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            int mm, kk;
            if (a_kfast) { kk = tid & 15; mm = (tid >> 4) + 16 * it; }
            else { mm = tid & 63; kk = (tid >> 6) + 4 * it; }
            float v = 0.f;
            if (m0 + mm < p.M && k0 + kk < p.K) v = ldf(A + (int64_t)(m0 + mm) * p.a_rs + (int64_t)(k0 + kk) * p.a_cs);
            As[kk * GS_LD + mm] = v;
            int nn, k2;
            if (b_kfast) { k2 = tid & 15; nn = (tid >> 4) + 16 * it; }
            else { nn = tid & 63; k2 = (tid >> 6) + 4 * it; }
            float w = 0.f;
            if (n0 + nn < p.N && k0 + k2 < p.K) w = ldf(B + (int64_t)(k0 + k2) * p.b_rs + (int64_t)(n0 + nn) * p.b_cs);
            Bs[k2 * GS_LD + nn] = w;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GS_BK / 4; ks++) {
            const int kr = ks * 4 + (lane >> 4);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = As[kr * GS_LD + wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = Bs[kr * GS_LD + wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: C fragment layout col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = n0 + wn * 32 + j * 16 + (lane & 15);
            if (n >= p.N) continue;
            const float bv = p.bias ? ldf((const TBIAS*)p.bias + n) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (m >= p.M) continue;
                TC* c = C + (int64_t)m * p.c_rs + (int64_t)n * p.c_cs;
                float v = p.alpha * acc[i][j][r] + bv;
                if (p.beta != 0.f) v += p.beta * ldf(c);
                stf(c, v);
            }
        }
}

template <typename TA, typename TB, typename TC>
static void launch_strided(const GemmStridedArgs& a, int dtBias, dim3 grid, hipStream_t st) {
    if (dtBias == DB1_BF16) gemm_strided_kernel<TA, TB, TC, bf16_t><<<grid, 256, 0, st>>>(a);
    else gemm_strided_kernel<TA, TB, TC, float><<<grid, 256, 0, st>>>(a);
}

int db1_gemm_strided_generic(const GemmStridedArgs& a, int dtA, int dtB, int dtC, int dtBias, int batch, hipStream_t st) {
    dim3 grid((unsigned)((a.M + GS_BM - 1) / GS_BM), (unsigned)((a.N + GS_BN - 1) / GS_BN), (unsigned)batch);
    if (grid.y > 65535 || grid.z > 65535) DB1_FAIL(DB1_ERR_BAD_SHAPE, "gemm_strided: grid too large (N=%d batch=%d)", a.N, batch);
    const int key = (dtA << 2) | (dtB << 1) | dtC;
    switch (key) {
        case 0: launch_strided<float, float, float>(a, dtBias, grid, st); break;
        case 1: launch_strided<float, float, bf16_t>(a, dtBias, grid, st); break;
        case 2: launch_strided<float, bf16_t, float>(a, dtBias, grid, st); break;
        case 3: launch_strided<float, bf16_t, bf16_t>(a, dtBias, grid, st); break;
        case 4: launch_strided<bf16_t, float, float>(a, dtBias, grid, st); break;
        case 5: launch_strided<bf16_t, float, bf16_t>(a, dtBias, grid, st); break;
        case 6: launch_strided<bf16_t, bf16_t, float>(a, dtBias, grid, st); break;
        default: launch_strided<bf16_t, bf16_t, bf16_t>(a, dtBias, grid, st); break;
    }
    DB1_CHECK_LAUNCH("gemm_strided");
    return DB1_OK;
}
