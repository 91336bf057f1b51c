// Shared pieces of the relative-position flash attention kernels (relattn_flash.hip, relattn_flash16.hip).
#pragma once
#include "db1_common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#define FA_D 128
#define FA_BQ 128
#define FA_BK 32
#define FA_RING 256
#define FA_TW_BYTES 8704   // per-wave scratch: T [32][64] f32 (8192 B) or a [32][136] bf16 output staging tile (8704 B)

struct FlashArgs {
    const bf16_t* qu; const bf16_t* qv; const bf16_t* k; const bf16_t* v; const bf16_t* R;
    const bf16_t* out; const bf16_t* dout; const float* lse; float* delta;
    bf16_t* o; float* lse_out;
    bf16_t* dq; bf16_t* dk; bf16_t* dv; bf16_t* dT;
    int64_t kv_rs, kv_bs;     // row / batch strides (elements) of k and v (they live inside the packed qkv activations)
    int64_t dq_rs, dq_bs;     // same for dq / dk / dv
    int B, L, H, shift;
    float scale;
};

__device__ __forceinline__ int crow(int r, int hb) { return (r & 3) + 8 * (r >> 2) + 4 * hb; }  // C-layout row of register r
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// one 1 KiB global_load_lds piece = 4 rows of 256 B; lane -> (row = lane >> 4, chunk position = lane & 15)
// The LDS-DMA is issued from inline asm on purpose: with the builtin, hipcc knows an LDS write is pending on vmcnt and puts
// `s_waitcnt vmcnt(0)` in front of the first LDS access it cannot disambiguate (the tr reads of V / K, the scratch writes
// of the backward kernels), i.e. the prefetch of the NEXT block was drained in the middle of (bwd_kv: at the start of) the
// current one.  The kernels wait for their own prefetch explicitly (vmcnt(0) + barrier at the end of every block).
__device__ __forceinline__ void glds_row(const bf16_t* row_ptr, int row_for_swz, char* lds_piece, int lane) {
    const int c = (lane & 15) ^ swz(row_for_swz);
    const bf16_t* src = row_ptr + c * 8;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(char, lds_piece));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
// {lse[i0 .. i0+32), delta[i0 .. i0+32)} -> 64 floats in LDS, one 4-byte LDS-DMA per lane of ONE wave
__device__ __forceinline__ void glds_stat(const float* lse, const float* delta, int i0, float* dst_lds, int lane) {
    const float* src = (lane < 32 ? lse : delta - 32) + i0 + lane;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(float, dst_lds));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
// stage a [32][128] bf16 tile whose global rows are row0 .. row0+31 (stride rs): 8 pieces, 2 per wave
__device__ __forceinline__ void stage_tile32(const bf16_t* g, int64_t rs, int row0, char* tile, int wave, int lane) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int piece = wave * 2 + it;
        const int r = piece * 4 + (lane >> 4);
        glds_row(g + (int64_t)(row0 + r) * rs, r, tile + piece * 1024, lane);
    }
}
// ring rows for 32 consecutive distances starting at dist0 (multiple of 4); out-of-range rows are clamped (they are masked)
__device__ __forceinline__ void stage_ring32(const bf16_t* Rg, int64_t rs, int dist0, int L, char* ring, int wave, int lane) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int piece = wave * 2 + it;
        const int slot0 = (dist0 + piece * 4) & (FA_RING - 1);  // wave-uniform
        const int dist = dist0 + piece * 4 + (lane >> 4);
        const int gr = dist < 0 ? 0 : (dist > L - 1 ? L - 1 : dist);
        glds_row(Rg + (int64_t)gr * rs, slot0 + (lane >> 4), ring + slot0 * 256, lane);
    }
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
static unsigned flash_grid(int ntile, int H, int B) {
    const int per_xcd = (B * H + 7) / 8, chunks = (per_xcd + 3) / 4;
    return (unsigned)(8 * chunks * 4 * ntile);
}
#define LOG2E 1.4426950408889634f
