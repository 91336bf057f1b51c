// bf16 MFMA GEMM, 256x256x64 workgroup tile, 8 waves in two "ping-pong" groups, 2 LDS stages (128 KiB).
//
// PMC on the 128x128 and 256x128 kernels (profiles/) showed the matrix pipe ~45 % busy: the two waves that share
// a SIMD ran in lockstep (both reading fragments, then both queueing MFMAs, then both at the barrier).  Here the waves of a
// SIMD pair (w, w+4) belong to different groups that run the SAME interval sequence half a step apart:
//
//      interval :   2j            2j+1           2j+2          2j+3
//      group 0  :   LOAD(j)       COMPUTE(j)     LOAD(j+1)     COMPUTE(j+1)
//      group 1  :   COMPUTE(j-1)  LOAD(j)        COMPUTE(j)    LOAD(j+1)
//
// so one wave's fragment reads / LDS-DMA issue overlap its partner's 32 MFMAs.  "Set" j = (k-tile j/2, k-substep j%2):
// LOAD(j) = 12 ds_read_b128 (or 24 ds_read_b64_tr_b16) through inline asm (invisible to hipcc's waitcnt pass, which would
// otherwise drain vmcnt(0) before every LDS read), COMPUTE(j) = 32 x v_mfma_f32_16x16x32_bf16 into a 128x64 wave tile.
// One s_barrier ends every interval (group 1 starts with one extra, group 0 ends with one extra: equal counts).
// A k-tile is read during intervals 4t .. 4t+3; k-tile t+2 is DMA'd into the same stage from interval 4t+4 (group 0) /
// 4t+5 (group 1) and each wave waits for its own pieces (vmcnt(0), nothing else is in flight then) just before the barrier
// that ends interval 4t+7: every load has >= 2 intervals (~1 us) to land.
// Measured (tools/bench_kernels.py gemm): with the MFMAs removed the same loop still takes 65-85 % of the full
// time and with the LDS-DMA removed 70-85 %: the kernel sits where the L2 -> LDS stream (64 KiB per k-tile per CU, one k-tile
// of prefetch: all that fits beside two stages in 160 KiB) and the matrix pipe are about equally long.
// Group g owns output rows [128g, 128g+128); wave u of a group owns columns [64u, 64u+64).  LDS stage = A0 | A1 | B0 | B1
// (four 128-row sub-tiles in the K-major / M-major images of gemm_tile.h).
#include "gemm_tile.h"
#ifndef DB1_KROT
#define DB1_KROT 1
#endif
#include <stdlib.h>
#include <type_traits>

#define PP_STAGE_BYTES (4 * TILE_BYTES)
#define PP_LDS_BYTES (2 * PP_STAGE_BYTES)

template <bool KMAJOR>
__device__ __forceinline__ int pp_frag_off(int rbase, int ks, int lane, int h) {
    const int i = lane & 15, g = lane >> 4;
    if (KMAJOR) {
        const int row = rbase + i;
        return row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4);
    } else {
        const int kr = ks * 32 + g * 8 + h * 4 + (i >> 2);
        const int q = (rbase >> 2) + (i & 3);
        const int f = ((kr & 3) << 1) | (kr & 8);
        return kr * 256 + (((q >> 1) ^ f) << 4) + (q & 1) * 8;
    }
}
template <bool KMAJOR> struct PFrag;
template <> struct PFrag<true> {
    bf16x8_t v;
    __device__ __forceinline__ void read(unsigned a0, unsigned) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a0)); }
    __device__ __forceinline__ bf16x8_t get() const { return v; }
};
template <> struct PFrag<false> {
    bf16x4_t lo, hi;
    __device__ __forceinline__ void read(unsigned a0, unsigned a1) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
    }
    __device__ __forceinline__ bf16x8_t get() const { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
};

template <bool A_KMAJOR, bool B_KMAJOR, typename TC, typename TBIAS>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp_kernel(GemmTileArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, u = wave & 3;
    int tm, tn;
    tile_coords(blockIdx.x, p.tiles_m, p.tiles_n, 4, tm, tn);
    const int z = blockIdx.y, z0 = z / p.batch1, z1 = z % p.batch1;
    const bf16_t* A = p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const bf16_t* B = p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    const int m0 = tm * 256, n0 = tn * 256;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);

    // lane-constant fragment addresses in stage 0: A sub-tile = grp, B sub-tile = u >> 1
    unsigned aoff[2][8][2], boff[2][4][2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int h = 0; h < 2; h++) aoff[ks][i][h] = lds0 + grp * TILE_BYTES + pp_frag_off<A_KMAJOR>(i * 16, ks, lane, h);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int h = 0; h < 2; h++) boff[ks][j][h] = lds0 + (2 + (u >> 1)) * TILE_BYTES + pp_frag_off<B_KMAJOR>((u & 1) * 64 + j * 16, ks, lane, h);
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS-DMA sources as (wave-uniform base pointer) + (lane-constant 32-bit byte offset): no per-tile vector address math.
    // piece q = 2 * wave + it of a 16 KiB sub-tile; the two sub-tiles of an operand differ by a uniform 128-row step.
    unsigned voffA[2], voffB[2];
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int q = wave * 2 + it;
        if (A_KMAJOR) { const int r = q * 8 + (lane >> 3); voffA[it] = (unsigned)((r * p.lda + (((lane & 7) ^ (r & 7)) << 3)) * 2); }
        else { const int kr = q * 4 + (lane >> 4); voffA[it] = (unsigned)((kr * p.lda + (((lane & 15) ^ (((kr & 3) << 1) | (kr & 8))) << 3)) * 2); }
        if (B_KMAJOR) { const int r = q * 8 + (lane >> 3); voffB[it] = (unsigned)((r * p.ldb + (((lane & 7) ^ (r & 7)) << 3)) * 2); }
        else { const int kr = q * 4 + (lane >> 4); voffB[it] = (unsigned)((kr * p.ldb + (((lane & 15) ^ (((kr & 3) << 1) | (kr & 8))) << 3)) * 2); }
    }
    const char* Abase = (const char*)(A_KMAJOR ? A + (int64_t)m0 * p.lda : A + m0);
    const char* Bbase = (const char*)(B_KMAJOR ? B + (int64_t)n0 * p.ldb : B + n0);
    const int64_t a_kstep = (A_KMAJOR ? (int64_t)TBK : (int64_t)TBK * p.lda) * 2, a_half = (A_KMAJOR ? 128 * p.lda : (int64_t)128) * 2;
    const int64_t b_kstep = (B_KMAJOR ? (int64_t)TBK : (int64_t)TBK * p.ldb) * 2, b_half = (B_KMAJOR ? 128 * p.ldb : (int64_t)128) * 2;
    // k-tile rotation: workgroup (tm, tn) walks the contraction starting at a different k-tile (and wraps).  With a row stride
    // that is a multiple of 4 KiB (K = 2048 bf16) all 512 rows of a k-tile map to the same memory channel; unrotated, every
    // workgroup of the chip hits the same channel at the same time (measured: qkv NT 909 -> 1067 TFLOP/s with a padded ld).
    const int nt_ = p.K / TBK;
    const int rot = DB1_KROT ? (int)((blockIdx.x & 7) * nt_) >> 3 : 0;  // per XCD: workgroups that share panels in one L2 stay in step
    auto krot = [&](int t) __attribute__((always_inline)) { const int k = t + rot; return k >= nt_ ? k - nt_ : k; };
    auto stage = [&](int t) __attribute__((always_inline)) {  // 8 global_load_lds per wave
        char* s = smem + (t & 1) * PP_STAGE_BYTES + wave * 2048;
        const int kt = krot(t);
        const char* a0 = Abase + kt * a_kstep;
        const char* b0 = Bbase + kt * b_kstep;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            __builtin_amdgcn_global_load_lds(a0 + voffA[it], LDS_PTR(void, s + it * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(a0 + a_half + voffA[it], LDS_PTR(void, s + TILE_BYTES + it * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(b0 + voffB[it], LDS_PTR(void, s + 2 * TILE_BYTES + it * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(b0 + b_half + voffB[it], LDS_PTR(void, s + 3 * TILE_BYTES + it * 1024), 16, 0, 0);
        }
    };
    const int nt = p.K / TBK, nsets = 2 * nt;
    stage(0);
    if (nt > 1) {
        stage(1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // k-tile 0 is in LDS

    PFrag<A_KMAJOR> af[8];
    PFrag<B_KMAJOR> bfr[4];
    auto load_set = [&](int j) __attribute__((always_inline)) {
        const int ks = j & 1;
        const unsigned sb = ((j >> 1) & 1) * PP_STAGE_BYTES;
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) af[i].read(aoff[0][i][0] + sb, aoff[0][i][1] + sb);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) bfr[jj].read(boff[0][jj][0] + sb, boff[0][jj][1] + sb);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) af[i].read(aoff[1][i][0] + sb, aoff[1][i][1] + sb);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) bfr[jj].read(boff[1][jj][0] + sb, boff[1][jj][1] + sb);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0 && j >= 2 && (j >> 1) + 1 < nt) stage((j >> 1) + 1);  // the stage read two k-tiles ago is free again
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jj].get(), af[i].get(), acc[i][jj], 0, 0, 0);  // swapped: D[n][m]
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (grp == 0) {
        for (int j = 0; j < nsets; j++) {
            load_set(j);
            __builtin_amdgcn_s_barrier();
            compute();
            if (j & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of the next k-tile (issued >= 2 intervals ago)
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __builtin_amdgcn_s_barrier();
        for (int j = 0; j < nsets; j++) {
            load_set(j);
            if (j & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute();
            __builtin_amdgcn_s_barrier();
        }
    }
    if (A_KMAJOR && B_KMAJOR && sizeof(TC) == 2 && p.split_n > 0 && n0 < p.split_n) {  // (workgroup-uniform: split_n is a multiple of the tile width)
        __syncthreads();
        store_wave_tile_bf16<TBIAS>(acc, smem + wave * 16384, (bf16_t*)p.Cu, p.ld_uv, m0 + grp * 128, n0 + u * 64, p.alpha, p.bias_u, lane);
        store_wave_tile_bf16<TBIAS>(acc, smem + wave * 16384, (bf16_t*)p.Cv, p.ld_uv, m0 + grp * 128, n0 + u * 64, p.alpha, p.bias_v, lane);
        return;
    }
    TC* C = (TC*)p.C + z0 * p.c_bs0 + z1 * p.c_bs1;
    if (sizeof(TC) == 2 && p.beta == 0.f) {  // bf16 output, nothing to accumulate: full-line stores through the (now free) operand stages
        __syncthreads();
        store_wave_tile_bf16<TBIAS>(acc, smem + wave * 16384, (bf16_t*)C, p.ldc, m0 + grp * 128, n0 + u * 64, p.alpha, p.bias, lane);
        return;
    }
    if (sizeof(TC) == 2) {  // bf16 output accumulated onto C: fp32 staging, two halves of 64 rows
        __syncthreads();
        store_wave_half_bf16_beta<TBIAS, 0>(acc, smem + wave * 16384, (bf16_t*)C, p.ldc, m0 + grp * 128, n0 + u * 64, p.alpha, p.beta, p.bias, lane);
        store_wave_half_bf16_beta<TBIAS, 4>(acc, smem + wave * 16384, (bf16_t*)C, p.ldc, m0 + grp * 128, n0 + u * 64, p.alpha, p.beta, p.bias, lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            store_frag<TC, TBIAS>(acc[i][j], C, p.ldc, m0 + grp * 128 + i * 16 + (lane & 15), n0 + u * 64 + j * 16 + (lane >> 4) * 4, p.alpha, p.beta, p.bias);
}

template <bool AK, bool BK_>
static void launch_pp(const GemmTileArgs& t, int dtC, int dtBias, dim3 grid, hipStream_t st) {
    static Db1PerDeviceOnce attr_once;   // dynamic LDS above 64 KiB needs the opt-in attribute: once per device, every instantiation
    attr_once.run([] {
#define SET_ATTR(TC, TB) hipFuncSetAttribute((const void*)gemm_bf16_pp_kernel<AK, BK_, TC, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES)
        SET_ATTR(float, float); SET_ATTR(float, bf16_t); SET_ATTR(bf16_t, float); SET_ATTR(bf16_t, bf16_t);
#undef SET_ATTR
    });
    if (dtC == DB1_F32) {
        if (dtBias == DB1_BF16) gemm_bf16_pp_kernel<AK, BK_, float, bf16_t><<<grid, 512, PP_LDS_BYTES, st>>>(t);
        else gemm_bf16_pp_kernel<AK, BK_, float, float><<<grid, 512, PP_LDS_BYTES, st>>>(t);
    } else {
        if (dtBias == DB1_BF16) gemm_bf16_pp_kernel<AK, BK_, bf16_t, bf16_t><<<grid, 512, PP_LDS_BYTES, st>>>(t);
        else gemm_bf16_pp_kernel<AK, BK_, bf16_t, float><<<grid, 512, PP_LDS_BYTES, st>>>(t);
    }
}

int db1_gemm_pp_launch(const GemmTileArgs& t_in, int fa, int fb, int dtC, int dtBias, int batch, hipStream_t st) {
    if (db1_gemm_w4_supported(t_in, fa, fb, dtC, batch)) return db1_gemm_w4_launch(t_in, fa, fb, dtC, dtBias, batch, st);
    GemmTileArgs t = t_in;
    t.tiles_m = t.M / 256;
    t.tiles_n = t.N / 256;
    t.ksplit = 1;
    dim3 grid((unsigned)(t.tiles_m * t.tiles_n), (unsigned)batch);
    if (fa == 0 && fb == 0) launch_pp<true, true>(t, dtC, dtBias, grid, st);
    else if (fa == 0 && fb == 1) launch_pp<true, false>(t, dtC, dtBias, grid, st);
    else launch_pp<false, false>(t, dtC, dtBias, grid, st);
    DB1_CHECK_LAUNCH("gemm_bf16_pp");
    return DB1_OK;
}
