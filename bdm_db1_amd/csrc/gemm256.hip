// bf16 MFMA GEMM, 256x128x64 workgroup tile, 3-stage LDS ring, 8 waves as 4 x 2 (64x64 wave tiles, 2 waves/SIMD).
// Used where the 256x256 ping-pong kernels do not apply (N % 256 != 0, e.g. the per-head N = 128 contractions over dT) or would
// leave the chip under-filled.  Variants measured and dropped: 4 waves with 128x64 wave tiles (slower), and a staggered schedule
// in which the second wave half defers one MFMA batch past the barrier (no gain; the idea lives on as gemm_pp.hip's ping-pong).
//
// Why this shape on gfx950: one k-step of the 128x128 kernel is ~0.25 us of MFMA work per workgroup while a
// global_load_lds round trip is 1-2 us, so with a 2-stage buffer every k-step ends in a vmcnt(0) drain.  Here the
// stages are 48 KiB (A 256x64 + B 128x64), three of them (144 KiB of the 160 KiB LDS), loads are issued TWO tiles
// ahead, and the per-tile wait is a COUNTED vmcnt (the newest tile stays in flight across the barrier) with a raw
// s_barrier.  hipcc (ROCm 7.2) cannot prove that an in-flight LDS-DMA and a ds_read touch different stages and would
// put `s_waitcnt vmcnt(0)` in front of every fragment read (measured: that made this kernel slower than the 2-stage
// one), so the fragment reads are issued through inline asm with hand-counted lgkmcnt waits + sched_barrier
// (cdna_hip_programming.md 5.7): reads of k-substep 1 are in flight under the MFMAs of k-substep 0.
// One barrier per k-step.  The 64x64 wave tile needs 1 KiB of fragment reads per 16 MFMAs (LDS read bandwidth ~= MFMA time).
// Operand forms, swizzles and the swapped-operand epilogue are those of gemm.hip (gemm_tile.h); an A tile is staged as two
// independent 128-row sub-tiles.
#include "gemm_tile.h"
#include <stdlib.h>

#define T256_STAGE_BYTES (3 * TILE_BYTES)   // A0 | A1 | B
#define T256_LDS_BYTES (3 * T256_STAGE_BYTES)

// byte offset (inside one 16 KiB sub-tile) of the h-th read of the fragment "rows rbase + lane&15, k-substep ks"
template <bool KMAJOR>
__device__ __forceinline__ int frag_off(int rbase, int ks, int lane, int h) {
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
// the reads themselves: hidden from the compiler's waitcnt bookkeeping on purpose.  The destination registers must not be
// touched (not even copied) before the hand-placed lgkmcnt wait, so a transposed fragment stays as two 64-bit halves until
// get() is called AFTER the wait.
template <bool KMAJOR> struct Frag;
template <> struct Frag<true> {
    bf16x8_t v;
    __device__ __forceinline__ void read(unsigned addr0, unsigned) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr0)); }
    __device__ __forceinline__ bf16x8_t get() const { return v; }
};
template <> struct Frag<false> {
    bf16x4_t lo, hi;
    __device__ __forceinline__ void read(unsigned addr0, unsigned addr1) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(addr1));
    }
    __device__ __forceinline__ bf16x8_t get() const { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
};

template <bool A_KMAJOR, bool B_KMAJOR, typename TC, typename TBIAS>
__global__ __launch_bounds__(512, 1) void gemm_bf16_tile256_kernel(GemmTileArgs p) {
    constexpr int NWM = 4, NWN = 2, NW = NWM * NWN, NI = 256 / NWM / 16, NJ = 128 / NWN / 16, PIECES = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    int tm, tn;
    tile_coords(blockIdx.x, p.tiles_m, p.tiles_n, 4, tm, tn);
    const int z = blockIdx.y, z0 = z / p.batch1, z1 = z % p.batch1;
    const bf16_t* A = p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const bf16_t* B = p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    const int m0 = tm * 256, n0 = tn * TBN;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);

    // lane-constant fragment addresses inside stage 0: [ks][frag][half]
    unsigned aoff[2][NI][2], boff[2][NJ][2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int row = wm * (256 / NWM) + i * 16;
#pragma unroll
            for (int h = 0; h < 2; h++) aoff[ks][i][h] = lds0 + (row >> 7) * TILE_BYTES + frag_off<A_KMAJOR>(row & 127, ks, lane, h);
        }
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int h = 0; h < 2; h++) boff[ks][j][h] = lds0 + 2 * TILE_BYTES + frag_off<B_KMAJOR>(wn * (128 / NWN) + j * 16, ks, lane, h);
    }

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // k-tiles that can hold non-zeros of A (structural-zero hint, gemm_tile.h): t -> k-tile index
    int nt = p.K / TBK, tri_lo = 0, tri_cnt = 1, tri_per = 1;
    if (p.tri_mode == 1) {
        const int lim = m0 / TBK + 256 / TBK;
        nt = nt < lim ? nt : lim;
    } else if (p.tri_mode == 2) {
        tri_per = p.tri_period / TBK;
        tri_lo = m0 / TBK < tri_per ? m0 / TBK : tri_per;
        tri_cnt = tri_per - tri_lo;
        nt = (nt / tri_per) * tri_cnt;
    }
    auto ktile = [&](int t) { return p.tri_mode == 2 ? (t / tri_cnt) * tri_per + tri_lo + t % tri_cnt : t; };
    auto stage = [&](int t, int buf) {  // 3 * PIECES global_load_lds per wave
        char* s = smem + buf * T256_STAGE_BYTES;
        const int k0 = ktile(t) * TBK;
        stage_tile<A_KMAJOR, PIECES>(A, p.lda, m0, k0, s, wave, lane);
        stage_tile<A_KMAJOR, PIECES>(A, p.lda, m0 + 128, k0, s + TILE_BYTES, wave, lane);
        stage_tile<B_KMAJOR, PIECES>(B, p.ldb, n0, k0, s + 2 * TILE_BYTES, wave, lane);
    };
    if (nt > 0) stage(0, 0);
    if (nt > 1) stage(1, 1);
    auto reads = [&](int ks, Frag<A_KMAJOR>* af, Frag<B_KMAJOR>* bf, unsigned sb) {
#pragma unroll
        for (int i = 0; i < NI; i++) af[i].read(aoff[ks][i][0] + sb, aoff[ks][i][1] + sb);
#pragma unroll
        for (int j = 0; j < NJ; j++) bf[j].read(boff[ks][j][0] + sb, boff[ks][j][1] + sb);
    };
    auto mfmas = [&](const Frag<A_KMAJOR>* af, const Frag<B_KMAJOR>* bf) {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j].get(), af[i].get(), acc[i][j], 0, 0, 0);  // swapped: D[n][m]
    };
    auto tile_ready = [&](int t) {  // tile t has landed once at most the loads of tile t+1 are still outstanding
        if (t + 1 < nt) {
            static_assert(PIECES == 2, "the counted wait below assumes 3 sub-tiles x 2 pieces per wave and k-tile");
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // everyone's pieces of tile t are in LDS; everyone is done reading tile t-1
    };
    int buf = 0;
    for (int t = 0; t < nt; t++) {  // [fragment reads of both k-substeps | 2 MFMA batches]
        tile_ready(t);
        if (t + 2 < nt) stage(t + 2, buf == 0 ? 2 : buf - 1);  // overwrites the stage read during iteration t-1
        const unsigned sb = buf * T256_STAGE_BYTES;
        Frag<A_KMAJOR> af0[NI], af1[NI];
        Frag<B_KMAJOR> bf0[NJ], bf1[NJ];
        reads(0, af0, bf0, sb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        reads(1, af1, bf1, sb);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(af0, bf0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfmas(af1, bf1);
        buf = buf == 2 ? 0 : buf + 1;
    }
    TC* C = (TC*)p.C + z0 * p.c_bs0 + z1 * p.c_bs1;
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
            store_frag<TC, TBIAS>(acc[i][j], C, p.ldc, m0 + wm * (256 / NWM) + i * 16 + (lane & 15), n0 + wn * (128 / NWN) + j * 16 + (lane >> 4) * 4,
                                  p.alpha, p.beta, p.bias);
}

template <bool AK, bool BK_>
static void launch256(const GemmTileArgs& t, int dtC, int dtBias, dim3 grid, hipStream_t st) {
    static Db1PerDeviceOnce attr_once;   // dynamic LDS above 64 KiB needs the opt-in attribute: once per device, every instantiation
    attr_once.run([] {
#define SET_ATTR(TC, TB) hipFuncSetAttribute((const void*)gemm_bf16_tile256_kernel<AK, BK_, TC, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, T256_LDS_BYTES)
        SET_ATTR(float, float); SET_ATTR(float, bf16_t); SET_ATTR(bf16_t, float); SET_ATTR(bf16_t, bf16_t);
#undef SET_ATTR
    });
    if (dtC == DB1_F32) {
        if (dtBias == DB1_BF16) gemm_bf16_tile256_kernel<AK, BK_, float, bf16_t><<<grid, 512, T256_LDS_BYTES, st>>>(t);
        else gemm_bf16_tile256_kernel<AK, BK_, float, float><<<grid, 512, T256_LDS_BYTES, st>>>(t);
    } else {
        if (dtBias == DB1_BF16) gemm_bf16_tile256_kernel<AK, BK_, bf16_t, bf16_t><<<grid, 512, T256_LDS_BYTES, st>>>(t);
        else gemm_bf16_tile256_kernel<AK, BK_, bf16_t, float><<<grid, 512, T256_LDS_BYTES, st>>>(t);
    }
}

int db1_gemm_tile256_launch(const GemmTileArgs& t_in, int fa, int fb, int dtC, int dtBias, int batch, hipStream_t st) {
    GemmTileArgs t = t_in;
    t.tiles_m = t.M / 256;
    t.tiles_n = t.N / TBN;
    t.ksplit = 1;
    dim3 grid((unsigned)(t.tiles_m * t.tiles_n), (unsigned)batch);
    if (fa == 0 && fb == 0) launch256<true, true>(t, dtC, dtBias, grid, st);
    else if (fa == 0 && fb == 1) launch256<true, false>(t, dtC, dtBias, grid, st);
    else launch256<false, false>(t, dtC, dtBias, grid, st);
    DB1_CHECK_LAUNCH("gemm_bf16_tile256");
    return DB1_OK;
}
