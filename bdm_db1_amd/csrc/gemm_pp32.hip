// bf16 MFMA GEMM, 256x256 workgroup tile, 8 waves in two ping-pong groups (see gemm_pp.hip), k-tiles of 32 in a ring of NS stages.
//
// Same interval scheme as gemm_pp.hip -- group 0: LOAD(j) | COMPUTE(j), group 1 the same one interval later, one s_barrier per
// interval -- but the LDS holds NS = 5 (or 4) stages of 32 KiB instead of 2 of 64 KiB: a k-tile is requested NS - 1 tiles
// (2 NS - 3 intervals) before its first read instead of one 64-wide tile (2-3 intervals), i.e. up to 128 KiB per CU are in
// flight from L2 instead of 64 KiB, and every LOAD interval issues the same 4 LDS-DMA pieces per wave.  Waits are counted:
// a wave allows min(NS - 2, tiles left) of its own tiles to stay outstanding.
//   K-major sub-tile : [128 rows][64 B]; 16-B chunk c of row r sits at position c ^ SW[(r >> 2) & 3], SW = {0, 2, 3, 1}
//                      (conflict-free for ds_read_b128's 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...)
//   M-major sub-tile : [32 k][256 B], the image of gemm_tile.h restricted to 32 k-rows.
#include "gemm_tile.h"
#ifndef DB1_KROT
#define DB1_KROT 1
#endif
#include <stdlib.h>

#define P32_SUB_BYTES 8192             // one 128-row operand sub-tile of a 32-wide k-tile
#define P32_STAGE_BYTES (4 * P32_SUB_BYTES)

__device__ __forceinline__ int p32_sw(int r) { return (0x78 >> (((r >> 2) & 3) * 2)) & 3; }  // {0, 2, 3, 1}[(r >> 2) & 3]

template <bool KMAJOR>
__device__ __forceinline__ int p32_frag_off(int rbase, int lane, int h) {
    const int i = lane & 15, g = lane >> 4;
    if (KMAJOR) {
        const int row = rbase + i;
        return row * 64 + ((g ^ p32_sw(row)) << 4);
    } else {
        const int kr = g * 8 + h * 4 + (i >> 2);
        const int q = (rbase >> 2) + (i & 3);
        const int f = ((kr & 3) << 1) | (kr & 8);
        return kr * 256 + (((q >> 1) ^ f) << 4) + (q & 1) * 8;
    }
}
// lane-constant byte offset of this lane's 16 B inside piece `wave` of a sub-tile, relative to the sub-tile's first element
template <bool KMAJOR>
__device__ __forceinline__ unsigned p32_src_off(int wave, int lane, int64_t ld) {
    if (KMAJOR) {
        const int r = wave * 16 + (lane >> 2);
        return (unsigned)((r * ld + (((lane & 3) ^ p32_sw(r)) << 3)) * 2);
    } else {
        const int kr = wave * 4 + (lane >> 4);
        return (unsigned)((kr * ld + (((lane & 15) ^ (((kr & 3) << 1) | (kr & 8))) << 3)) * 2);
    }
}
template <bool KMAJOR> struct P32Frag;
template <> struct P32Frag<true> {
    bf16x8_t v;
    __device__ __forceinline__ void read(unsigned a0, unsigned) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a0)); }
    __device__ __forceinline__ bf16x8_t get() const { return v; }
};
template <> struct P32Frag<false> {
    bf16x4_t lo, hi;
    __device__ __forceinline__ void read(unsigned a0, unsigned a1) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
    }
    __device__ __forceinline__ bf16x8_t get() const { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
};

template <bool A_KMAJOR, bool B_KMAJOR, typename TC, typename TBIAS, int NS>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp32_kernel(GemmTileArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int P = NS - 1;  // prefetch distance in k-tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, u = wave & 3;
    int tm, tn;
    tile_coords(blockIdx.x, p.tiles_m, p.tiles_n, 4, tm, tn);
    const int z = blockIdx.y, z0 = z / p.batch1, z1 = z % p.batch1;
    const bf16_t* A = p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const bf16_t* B = p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    const int m0 = tm * 256, n0 = tn * 256;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);

    unsigned aoff[8][2], boff[4][2];  // fragment addresses in stage 0: A sub-tile = grp, B sub-tile = u >> 1
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int h = 0; h < 2; h++) aoff[i][h] = lds0 + grp * P32_SUB_BYTES + p32_frag_off<A_KMAJOR>(i * 16, lane, h);
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int h = 0; h < 2; h++) boff[j][h] = lds0 + (2 + (u >> 1)) * P32_SUB_BYTES + p32_frag_off<B_KMAJOR>((u & 1) * 64 + j * 16, lane, h);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned voffA = p32_src_off<A_KMAJOR>(wave, lane, p.lda), voffB = p32_src_off<B_KMAJOR>(wave, lane, p.ldb);
    const char* Abase = (const char*)(A_KMAJOR ? A + (int64_t)m0 * p.lda : A + m0);
    const char* Bbase = (const char*)(B_KMAJOR ? B + (int64_t)n0 * p.ldb : B + n0);
    const int64_t a_kstep = (A_KMAJOR ? (int64_t)32 : (int64_t)32 * p.lda) * 2, a_half = (A_KMAJOR ? 128 * p.lda : (int64_t)128) * 2;
    const int64_t b_kstep = (B_KMAJOR ? (int64_t)32 : (int64_t)32 * p.ldb) * 2, b_half = (B_KMAJOR ? 128 * p.ldb : (int64_t)128) * 2;
    const int nt_ = p.K / 32;  // k-tile rotation per workgroup against memory-channel camping (see gemm_pp.hip)
    const int rot = 0;  // measured: rotation only helps NT (gemm_pp.hip); NN -1 %, TN -3..-9 %
    auto krot = [&](int t) __attribute__((always_inline)) { const int k = t + rot; return k >= nt_ ? k - nt_ : k; };
    auto stage = [&](int t) __attribute__((always_inline)) {  // this wave's piece of each of the four sub-tiles of k-tile t
        char* s = smem + (t % NS) * P32_STAGE_BYTES + wave * 1024;
        const int kt = krot(t);
        const char* a0 = Abase + kt * a_kstep;
        const char* b0 = Bbase + kt * b_kstep;
        __builtin_amdgcn_global_load_lds(a0 + voffA, LDS_PTR(void, s), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(a0 + a_half + voffA, LDS_PTR(void, s + P32_SUB_BYTES), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b0 + voffB, LDS_PTR(void, s + 2 * P32_SUB_BYTES), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b0 + b_half + voffB, LDS_PTR(void, s + 3 * P32_SUB_BYTES), 16, 0, 0);
    };
    // wait until at most `tiles` of this wave's k-tiles (4 loads each) are still outstanding
    auto wait_outstanding = [&](int tiles) __attribute__((always_inline)) {
        if (tiles >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (tiles == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (tiles == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int nt = p.K / 32;
    const int npro = nt < P ? nt : P;
    for (int t = 0; t < npro; t++) stage(t);
    wait_outstanding(npro - 1 < P - 1 ? npro - 1 : P - 1);
    __builtin_amdgcn_s_barrier();  // k-tile 0 is in LDS

    P32Frag<A_KMAJOR> af[8];
    P32Frag<B_KMAJOR> bfr[4];
    auto load_set = [&](int j) __attribute__((always_inline)) {
        const unsigned sb = (unsigned)(j % NS) * P32_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 8; i++) af[i].read(aoff[i][0] + sb, aoff[i][1] + sb);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) bfr[jj].read(boff[jj][0] + sb, boff[jj][1] + sb);
        __builtin_amdgcn_sched_barrier(0);
        if (j + P < nt) stage(j + P);  // into the stage of k-tile j-1, whose last reads ended before the previous barrier
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
    // before the barrier that ends interval 2j+1 every wave makes sure its pieces of k-tile j+1 have landed: of the tiles it
    // has requested (up to j+P) only those after j+1 may stay in flight
    auto wait_next = [&](int j) __attribute__((always_inline)) {
        if (j + 1 < nt) {
            const int last = j + P < nt - 1 ? j + P : nt - 1;
            wait_outstanding(last - (j + 1));
        }
    };
    if (grp == 0) {
        for (int j = 0; j < nt; j++) {
            load_set(j);
            __builtin_amdgcn_s_barrier();
            compute();
            wait_next(j);
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __builtin_amdgcn_s_barrier();
        for (int j = 0; j < nt; j++) {
            load_set(j);
            wait_next(j);
            __builtin_amdgcn_s_barrier();
            compute();
            __builtin_amdgcn_s_barrier();
        }
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

template <bool AK, bool BK_, int NS>
static void launch_pp32(const GemmTileArgs& t, int dtC, int dtBias, dim3 grid, hipStream_t st) {
    constexpr int LDS = NS * P32_STAGE_BYTES;
    static Db1PerDeviceOnce attr_once;   // dynamic LDS above 64 KiB needs the opt-in attribute: once per device, every instantiation
    attr_once.run([] {
#define SET_ATTR(TC, TB) hipFuncSetAttribute((const void*)gemm_bf16_pp32_kernel<AK, BK_, TC, TB, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
        SET_ATTR(float, float); SET_ATTR(float, bf16_t); SET_ATTR(bf16_t, float); SET_ATTR(bf16_t, bf16_t);
#undef SET_ATTR
    });
    if (dtC == DB1_F32) {
        if (dtBias == DB1_BF16) gemm_bf16_pp32_kernel<AK, BK_, float, bf16_t, NS><<<grid, 512, LDS, st>>>(t);
        else gemm_bf16_pp32_kernel<AK, BK_, float, float, NS><<<grid, 512, LDS, st>>>(t);
    } else {
        if (dtBias == DB1_BF16) gemm_bf16_pp32_kernel<AK, BK_, bf16_t, bf16_t, NS><<<grid, 512, LDS, st>>>(t);
        else gemm_bf16_pp32_kernel<AK, BK_, bf16_t, float, NS><<<grid, 512, LDS, st>>>(t);
    }
}

int db1_gemm_pp32_launch(const GemmTileArgs& t_in, int fa, int fb, int dtC, int dtBias, int batch, hipStream_t st) {
    if (db1_gemm_w4_supported(t_in, fa, fb, dtC, batch)) return db1_gemm_w4_launch(t_in, fa, fb, dtC, dtBias, batch, st);
    GemmTileArgs t = t_in;
    t.tiles_m = t.M / 256;
    t.tiles_n = t.N / 256;
    t.ksplit = 1;
    dim3 grid((unsigned)(t.tiles_m * t.tiles_n), (unsigned)batch);
    const int ns = db1_knob(DB1_KNOB_PP32_STAGES, 4);   // A/B knob (4 | 5); measured: 4 stages (128 KiB) beat 5 (160 KiB) on every shape
#define FORMS(NS_)                                                                            \
    if (fa == 0 && fb == 0) launch_pp32<true, true, NS_>(t, dtC, dtBias, grid, st);           \
    else if (fa == 0 && fb == 1) launch_pp32<true, false, NS_>(t, dtC, dtBias, grid, st);     \
    else launch_pp32<false, false, NS_>(t, dtC, dtBias, grid, st);
    if (ns == 5) { FORMS(5) } else { FORMS(4) }
#undef FORMS
    DB1_CHECK_LAUNCH("gemm_bf16_pp32");
    return DB1_OK;
}
