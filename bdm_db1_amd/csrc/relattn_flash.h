// Shared pieces of the relative-position flash attention kernels (relattn_flash.hip, relattn_flash_fwd2.hip): argument block, fragment /
// LDS layout conventions (described at the top of relattn_flash.hip), lane-constant addresses, LDS-DMA staging helpers.
#pragma once
#include "db1_common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define LOG2E 1.4426950408889634f

#define FA_D 128
#define FA_BQ 128     // rows (queries, or keys in bwd_kv) per workgroup
#define FA_BK 32      // columns per block
#define FA_RING 256    // rows of the LDS ring of R

struct FlashArgs {
    const bf16_t* qu; const bf16_t* qv; const bf16_t* k; const bf16_t* v; const bf16_t* R;
    const bf16_t* out; const bf16_t* dout; const float* lse; float* delta;
    bf16_t* o; float* lse_out;
    bf16_t* dq; bf16_t* dk; bf16_t* dv; bf16_t* dT;
    bf16_t* pt; float* mblk;       // forward-stored p~ = exp2(s - m_blk) images (same layout as pbuf) and the running maxima they refer to, [B*H][L/32][L] (x scale*log2 e), or null
    float* fblk;                   // exp2(m_blk c2 - lse log2 e) per (key block, query): written by bwd_q2, read by kv2<true>; same shape as mblk (+ 64 floats)
    bf16_t* pbuf; bf16_t* dsbuf;  // P and dS as fragment images [B*H][L/32 key blocks][L/16 query tiles][64 lanes][8] bf16 (stored-probabilities backward), or null
    int64_t kv_rs, kv_bs;     // row / batch strides (elements) of k and v (they live inside the packed qkv activations)
    int64_t dq_rs, dq_bs;     // same for dq / dk / dv
    int B, L, H, shift;
    float scale;
};

// The fragment images (p~ kept by the forward; P / dS of the scratch mode): 1 KiB per (key block jb of 32, 16-query tile qt) and (batch, head),
// [64 lanes][8] bf16.  A tile exists only on or below the causal diagonal (qt >= 2 jb: every window the kernels support is inside it), so the
// tiles of a (batch, head) are stored as a TRIANGLE, key block major: index = qt + jb (NT - 1 - jb), NT = L / 16 -- row jb starts where row
// jb - 1 ended, tiles of one key block stay adjacent in qt.  1056 instead of 2048 tiles at L = 1024: 25 instead of 49 GiB of kept
// probabilities for DB1-1.3B at 64 x 1024 tokens (round 6).  A read of a tile above the diagonal (the key-side kernels' prefetch at the edge
// of their walk) lands on some other tile of the same buffer: finite or NaN garbage that the edge blocks SELECT away, as before.
static __host__ __device__ inline int64_t flash_pt_tiles(int L) { const int64_t nkb = L / FA_BK, nt = L / 16; return nkb * nt - nkb * (nkb - 1); }
static __host__ __device__ inline int64_t flash_pt_index(int jb, int qt, int nt) { return (int64_t)qt + (int64_t)jb * (nt - 1 - jb); }

// {lse[i0 .. i0+32), delta[i0 .. i0+32)} -> 64 floats in LDS, one 4-byte LDS-DMA per lane of ONE wave
__device__ __forceinline__ void glds_stat(const float* lse, const float* delta, int i0, float* dst_lds, int lane) {
    const float* src = (lane < 32 ? lse : delta - 32) + i0 + lane;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(float, dst_lds));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {  // v_cvt_pk_bf16_f32 (round to nearest even)
    f32x2_t v = {lo, hi};
    bf16x2_hw r = __builtin_convertvector(v, bf16x2_hw);
    return *reinterpret_cast<unsigned*>(&r);
}
__device__ __forceinline__ bf16x8_t pack8(const float* p) {
    union { unsigned u[4]; bf16x8_t v; } o;
#pragma unroll
    for (int t = 0; t < 4; t++) o.u[t] = pk_bf16(p[2 * t], p[2 * t + 1]);
    return o.v;
}

// Workgroup id -> (tile rank, head, batch).  Hardware hands consecutive workgroup ids to the 8 XCDs round-robin and a free
// CU takes the next id, so the id order is the schedule:
//   * all tiles of one (batch, head) are given ids of ONE XCD, so its K / V / Q rows are shared in that XCD's L2;
//   * inside a chunk of 4 (batch, head) pairs per XCD (= 256 workgroups chip-wide at L = 1024) the ids go heaviest tile first
//     (rank 0 = the tile with the longest loop), so the light tiles fill the tail instead of one 32-block tile ending alone.
// returns false for the padding ids of a ragged last chunk
__device__ __forceinline__ bool flash_wg_coords(int ntile, int H, int B, int& rank, int& h, int& b) {
    const int i = blockIdx.x, xcd = i & 7, j = i >> 3;
    const int per_chunk = 4 * ntile, chunk = j / per_chunk, jl = j % per_chunk;
    rank = jl >> 2;
    const int bh = (chunk * 4 + (jl & 3)) * 8 + xcd;
    if (bh >= B * H) return false;
    h = bh % H;
    b = bh / H;
    return true;
}
static inline unsigned flash_grid(int ntile, int H, int B) {
    const int per_xcd = (B * H + 7) / 8, chunks = (per_xcd + 3) / 4;
    return (unsigned)(8 * chunks * 4 * ntile);
}


#define W16_WAVES 8
#define W16_STAGES 3
#define W16_TP 68            // scratch row pitch (words): 64 columns + 4 -> the column-wise writes and the skewed reads are both conflict-free
#define W16_TW_BYTES 4352    // per-wave scratch: T [16][68] f32 or the [16][136] bf16 output staging tile (both 4352 B)
#define W16_DP 72            // bwd_q dS scratch row pitch (bytes): [16 q][32 keys, reversed] bf16 + 8
#define W16_OFF_K 0          // three stages of 8 KiB
#define W16_OFF_V (W16_STAGES * 8192)
#define W16_OFF_R (2 * W16_STAGES * 8192)      // 256 ring rows of 256 B
#define W16_OFF_T (W16_OFF_R + FA_RING * 256)
#define W16_OFF_D (W16_OFF_T + W16_WAVES * W16_TW_BYTES)
#define W16_FWD_LDS W16_OFF_D
#define W16_BQ_LDS (W16_OFF_D + W16_WAVES * 16 * W16_DP)

__device__ __forceinline__ int swz_kv(int row) { return (((row >> 4) & 1) << 3) | ((row & 3) << 1) | ((row >> 2) & 1); }
__device__ __forceinline__ int swz_ring(int slot) { return ((((slot & 7) ^ ((slot & 8) >> 1))) << 1) | ((slot >> 3) & 1); }
__device__ __forceinline__ int kk16(int t, int g) { return 16 * ((g & 1) ^ t) + 8 * (g >> 1) + 4 * t; }

// LDS accesses go through absolute 32-bit LDS addresses kept in VGPRs (lane constants) plus immediate offsets: with pointer
// arithmetic on the dynamic-LDS symbol hipcc emitted one `v_add_u32 v, 0, v` per access (a third of the loop's VALU work).
typedef __attribute__((address_space(3))) const bf16x8_t* lds_b128_ptr;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;
typedef __attribute__((address_space(3))) float* lds_f32_ptr;
typedef __attribute__((address_space(3))) unsigned* lds_u32_ptr;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((address_space(3))) u32x2_t* lds_u64_ptr;
__device__ __forceinline__ bf16x8_t lds_ld128(unsigned addr) { return *(lds_b128_ptr)(size_t)addr; }
__device__ __forceinline__ float lds_ldf(unsigned addr) { return *(lds_f32_ptr)(size_t)addr; }
__device__ __forceinline__ void lds_stf(unsigned addr, float v) { *(lds_f32_ptr)(size_t)addr = v; }
__device__ __forceinline__ bf16x8_t lds_tr_pair(unsigned a0, unsigned a1) {
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)(size_t)a0);
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)(size_t)a1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// one 1 KiB LDS-DMA piece = 4 rows of 256 B, lane -> (row = lane >> 4, destination chunk = lane & 15); src already points at the
// lane's (swizzled) 16-byte source chunk, dst_lds is the wave-uniform LDS address of the piece.  Issued from inline asm on purpose:
// with the builtin, hipcc knows an LDS write is pending on vmcnt and puts `s_waitcnt vmcnt(0)` in front of the first LDS access it
// cannot disambiguate (tr reads, scratch writes), i.e. it drains the prefetch in the middle of the block.  The kernels wait for
// their own prefetch explicitly (counted vmcnt + barrier at the end of every block).
__device__ __forceinline__ void glds16(const void* src, unsigned dst_lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst_lds) : "memory");
}
__device__ __forceinline__ float vmax3(float x, float y, float z) {  // no NaN canonicalisation (fmaxf costs a v_max x,x per input)
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}
__device__ __forceinline__ float vmax2(float x, float y) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
    return d;
}
__device__ __forceinline__ float max_x16(float x) {  // max with lane ^ 16 (v_permlane16_swap: no LDS round trip)
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_x32(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_x16(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_x32(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void zero4(f32x4& x) { x[0] = 0.f; x[1] = 0.f; x[2] = 0.f; x[3] = 0.f; }

// lane-constant absolute LDS addresses (tile addresses are those of a tile at LDS offset 0: add W16_OFF_x + stage * 8192 as immediates)
struct W16Lane {
    unsigned rowf[2][4];  // A-operand row fragment of 16-row tile t, k-step ks: MFMA row a <-> tile row kk(t, a >> 2) + (a & 3)
    unsigned tr[2][8];    // ds_read_b64_tr_b16 of tile rows kk(t, g) + 0..3, d-tile db
    unsigned ring[4];     // natural-order row fragment (row & 15 == a) of a ring-swizzled image at LDS offset 0: + offset + (row16 << 8)
    unsigned tsk[2][8];   // scratch element (a, a - key + 32), key = kk(t, g) + r, for both parities
    unsigned twr;         // scratch write base: rows 4g + r, column a  (+ (r * W16_TP + (16 tile ^ 32 parity)) * 4)
};
// keep every address in its own VGPR: left alone, hipcc re-associates them into (common part) + (lane part) and re-adds per access
#define W16_OPAQUE(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ void w16_lane_init(W16Lane& o, unsigned lds0, unsigned tw0, int lane) {  // tw0: LDS address of the wave scratch
    const int a = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int rho = kk16(t, a >> 2) + (a & 3);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) { o.rowf[t][ks] = lds0 + rho * 256 + (((4 * ks + g) ^ swz_kv(rho)) << 4); W16_OPAQUE(o.rowf[t][ks]); }
        const int row = kk16(t, g) + (a >> 2);
#pragma unroll
        for (int db = 0; db < 8; db++) {
            o.tr[t][db] = lds0 + row * 256 + (((2 * db + ((a & 3) >> 1)) ^ swz_kv(row)) << 4) + (a & 1) * 8;
            W16_OPAQUE(o.tr[t][db]);
        }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) { o.ring[ks] = lds0 + (a << 8) + (((4 * ks + g) ^ swz_ring(a)) << 4); W16_OPAQUE(o.ring[ks]); }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int c = a - kk16(t, g) - r + 32;
            o.tsk[0][t * 4 + r] = tw0 + (a * W16_TP + c) * 4;
            o.tsk[1][t * 4 + r] = tw0 + (a * W16_TP + (c ^ 32)) * 4;
            W16_OPAQUE(o.tsk[0][t * 4 + r]);
            W16_OPAQUE(o.tsk[1][t * 4 + r]);
        }
    o.twr = tw0 + ((4 * g) * W16_TP + a) * 4;
    W16_OPAQUE(o.twr);
}
// staging state of one wave: every wave moves one 1 KiB piece (4 rows) of the K tile, of the V tile and of the 32 new ring rows per block
struct W16Stage {
    const bf16_t* kptr; const bf16_t* vptr;  // this lane's source chunk of the NEXT tile to stage (tiles are staged in key order)
    const bf16_t* rbase;                     // R + head + this lane's ring source chunk
    int64_t kv_step, r_rs;
    int srow, L;
    unsigned lds0;
};
__device__ __forceinline__ void w16_stage_init(W16Stage& s, const bf16_t* kg, const bf16_t* vg, const bf16_t* Rg, int64_t kv_rs, int64_t r_rs, int row0,
                                               int L, unsigned lds0, int wave, int lane) {
    s.srow = wave * 4 + (lane >> 4);
    const int schunk = ((lane & 15) ^ swz_kv(s.srow)) << 3;
    s.kptr = kg + (int64_t)(row0 + s.srow) * kv_rs + schunk;
    s.vptr = vg + (int64_t)(row0 + s.srow) * kv_rs + schunk;
    s.rbase = Rg + (((lane & 15) ^ swz_ring(s.srow & 15)) << 3);  // slot & 15 == srow & 15: distances are staged in multiples of 16
    s.kv_step = (int64_t)FA_BK * kv_rs;
    s.r_rs = r_rs;
    s.L = L;
    s.lds0 = lds0;
}
__device__ __forceinline__ void w16_stage_k(W16Stage& s, int stage, int wave) {
    glds16(s.kptr, s.lds0 + W16_OFF_K + stage * 8192 + wave * 1024);
    s.kptr += s.kv_step;
}
__device__ __forceinline__ void w16_stage_v(W16Stage& s, int stage, int wave) {
    glds16(s.vptr, s.lds0 + W16_OFF_V + stage * 8192 + wave * 1024);
    s.vptr += s.kv_step;
}
__device__ __forceinline__ void w16_stage_kv(W16Stage& s, int stage, int wave) {
    w16_stage_k(s, stage, wave);
    w16_stage_v(s, stage, wave);
}
template <int N> __device__ __forceinline__ void w16_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most n vector-memory operations are outstanding, n wave-uniform at run time (the instruction takes an immediate)
__device__ __forceinline__ void w16_vmcnt_dyn(int n) {
    switch (n) {
#define W16_VMC(k) case k: w16_vmcnt<k>(); break;
        W16_VMC(0) W16_VMC(1) W16_VMC(2) W16_VMC(3) W16_VMC(4) W16_VMC(5) W16_VMC(6) W16_VMC(7) W16_VMC(8) W16_VMC(9) W16_VMC(10)
        W16_VMC(11) W16_VMC(12) W16_VMC(13) W16_VMC(14) W16_VMC(15) W16_VMC(16) W16_VMC(17) W16_VMC(18) W16_VMC(19) W16_VMC(20)
        W16_VMC(21) W16_VMC(22) W16_VMC(23)
#undef W16_VMC
        default: w16_vmcnt<24>(); break;
    }
}
__device__ __forceinline__ void w16_stage_ring(const W16Stage& s, int dist0, int wave) {  // distances dist0 .. dist0+31 (dist0 a multiple of 16)
    const int slot0 = (dist0 + wave * 4) & (FA_RING - 1);
    const int dist = dist0 + s.srow;
    const int gr = dist < 0 ? 0 : (dist > s.L - 1 ? s.L - 1 : dist);  // out-of-range distances belong to masked pairs
    glds16(s.rbase + (int64_t)gr * s.r_rs, s.lds0 + W16_OFF_R + slot0 * 256);
}
// relative-term tile: distances dist16 .. dist16+15 (dist16 a multiple of 16) for the wave's 16 queries -> scratch columns col .. col+15
__device__ __forceinline__ void w16_rel_tile(const bf16x8_t* fqv, const W16Lane& o, int dist16, int col) {
    const unsigned rs = (unsigned)(dist16 & (FA_RING - 1)) << 8;  // wave-uniform
    f32x4 acc;
    zero4(acc);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) acc = MFMA16(fqv[ks], lds_ld128(o.ring[ks] + rs + W16_OFF_R), acc);
#pragma unroll
    for (int r = 0; r < 4; r++) lds_stf(o.twr + (r * W16_TP + col) * 4, acc[r]);
}
// acc^T [8 d-tiles](rows = d, col = lane & 15) -> bf16 rows [16][128] at dst (row stride rs), staged through the wave scratch
__device__ __forceinline__ void store_acc_t16(const f32x4* acc, float mul, bf16_t* Ow, bf16_t* dst, int64_t rs, int lane) {
    const int a = lane & 15, g = lane >> 4;
#pragma unroll
    for (int db = 0; db < 8; db++) {
        uint2 o;
        o.x = pk_bf16(acc[db][0] * mul, acc[db][1] * mul);
        o.y = pk_bf16(acc[db][2] * mul, acc[db][3] * mul);
        *reinterpret_cast<uint2*>(Ow + a * 136 + 16 * db + 4 * g) = o;
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int row = it * 4 + (lane >> 4), ch = lane & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(Ow + row * 136 + ch * 8);
        *reinterpret_cast<uint4*>(dst + (int64_t)row * rs + ch * 8) = v;
    }
}
#define W16_BLOCK_LOOP(block)                                                                                       \
    for (int jb = jb_lo; jb <= jb_hi; jb += 6) {                                                                    \
        block(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, jb);                              \
        if (jb + 1 <= jb_hi) block(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, jb + 1);     \
        if (jb + 2 <= jb_hi) block(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, jb + 2);     \
        if (jb + 3 <= jb_hi) block(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, jb + 3);     \
        if (jb + 4 <= jb_hi) block(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, jb + 4);     \
        if (jb + 5 <= jb_hi) block(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, jb + 5);     \
    }
