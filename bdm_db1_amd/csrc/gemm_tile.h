// Shared pieces of the bf16 MFMA tile GEMMs (gemm.hip: 128x128 tile / 4 waves; gemm256.hip: 256x128 tile / 8 waves).
#pragma once
#include "db1_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

#define TBM 128
#define TBN 128
#define TBK 64
#define TILE_BYTES (128 * 64 * 2)  // 16 KiB: one 128-row operand (sub-)tile per stage

struct GemmTileArgs {
    const bf16_t* A; const bf16_t* B; void* C; const void* bias;
    int M, N, K;
    int64_t lda, ldb, ldc;  // leading dimensions in elements
    int batch1;
    int64_t a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;
    float alpha, beta;
    int tiles_m, tiles_n;
    int ksplit;  // > 1: blockIdx.z owns K / ksplit of the contraction and accumulates into fp32 C with atomics (beta must be 1) ...
    int64_t c_zs = 0;   // ... or, when != 0, STORES its partial sum at C + blockIdx.z * c_zs (a workspace; splitk_reduce_kernel adds the slices in order)
    // structural-zero hint for A (256x128 kernel only; elsewhere ignored, the zeros are simply multiplied):
    //   1: A[m, k] == 0 for k > m                      -> k-tiles beyond the tile's last row are skipped (dq_r = dT . R)
    //   2: A[m, k] == 0 for (k mod tri_period) < m     -> per period only the k-tiles from the tile's first row on (dR = dT^T . qv)
    int tri_mode, tri_period;
    int tri_walk;   // mode 2 on one column of tiles: hand the tile rows out heaviest first (gemm_w4.hip; only read when tri_mode == 2)
    // head-bias epilogue of the attention input projection (ping-pong NT kernel only): columns n < split_n are written TWICE, as
    // acc + bias_u[n] to Cu and acc + bias_v[n] to Cv (row stride ld_uv) instead of to C: q + r_w_bias and q + r_r_bias straight from
    // the accumulators (db1_gemm_nt_headbias); 0 = off
    int split_n;
    void* Cu; void* Cv; const void* bias_u; const void* bias_v;
    int64_t ld_uv;
    // GEGLU epilogues of the 4-wave kernel (gemm_w4.hip; 0 = off).  Forward (NT, db1_gemm_nt_geglu): B = W1 [2 dff, K], C = z [M, 2 dff] and
    // Cact = z[:, :dff] * gelu(z[:, dff:]) [M, dff] leave the same accumulators.  Backward (NN, db1_gemm_nn_geglu_bwd): the product is
    // dact [M, dff] = dy W2; the epilogue reads Zin = z and writes C = dz [M, 2 dff] plus the column sums of dz per 128-row block to colpart.
    int geglu_dff = 0;
    void* Cact = nullptr; int64_t ld_act = 0;
    const bf16_t* Zin = nullptr; int64_t ld_z = 0;
    float* colpart = nullptr;
    int rot = 1;        // NT: per-XCD rotation of the k-tile walk (measured without effect at the five forward shapes, profiles/r05_nt_vs_nn.txt)
    int band = 4;       // tile rows per band of the XCD-aware walk of the 4-wave kernels (3 .. 8 measured flat, profiles/r05_w4_band_sweep.txt)
};

// ---- staging of one 16 KiB operand (sub-)tile = 16 wave-instructions of 1 KiB, PIECES per wave (wave w takes w*PIECES ..)
//   K-major : memory [row][k]   -> LDS image [128 rows][128 B];  piece q = rows [8q, 8q+8):   lane -> (r = lane/8,  cp = lane%8),  holds chunk cp ^ (r & 7)
//   M-major : memory [k][row]   -> LDS image [64 k][256 B];      piece q = k-rows [4q, 4q+4): lane -> (kr = lane/16, cp = lane%16), holds chunk cp ^ f(kr)
// rows_valid (< 128 only in a tail tile): rows beyond it are loaded from the last valid row / 8-row chunk instead (their
// products land in accumulator rows / columns that the masked epilogue never stores).
template <bool KMAJOR, int PIECES>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int64_t ld, int row0, int k0, char* lds, int wave, int lane,
                                           int rows_valid = 128) {
#pragma unroll
    for (int it = 0; it < PIECES; it++) {
        const int q = wave * PIECES + it;
        const bf16_t* src;
        if (KMAJOR) {
            const int r = q * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            const int re = r < rows_valid ? r : rows_valid - 1;
            src = g + (int64_t)(row0 + re) * ld + k0 + c * 8;
        } else {
            const int kr = q * 4 + (lane >> 4);
            const int f = ((kr & 3) << 1) | (kr & 8);
            int c = (lane & 15) ^ f;
            if (c * 8 + 8 > rows_valid) c = (rows_valid >> 3) - 1;
            src = g + (int64_t)(k0 + kr) * ld + row0 + c * 8;
        }
        __builtin_amdgcn_global_load_lds(src, LDS_PTR(void, lds + q * 1024), 16, 0, 0);
    }
}

// ---- fragment: 8 consecutive k (k = ks*32 + g*8 + 0..7) for tile row (rbase + lane&15); 16x16x32 A/B operand image
template <bool KMAJOR>
__device__ __forceinline__ bf16x8_t load_frag(const char* lds, int rbase, int ks, int lane) {
    const int i = lane & 15, g = lane >> 4;
    if (KMAJOR) {
        const int row = rbase + i;
        const int chunk = (ks * 4 + g) ^ (row & 7);
        return *reinterpret_cast<const bf16x8_t*>(lds + row * 128 + chunk * 16);
    } else {
        // tr16_b64: lane t of a 16-lane group passes the address of (k-row kb + t/4, 4 row-elements at rbase + (t%4)*4)
        // and receives k-rows kb..kb+3 of tile row rbase + t  (lane map verified in profiles/r01_probe_gfx950_layouts.txt)
        const int kb = ks * 32 + g * 8;
        const int q = (rbase >> 2) + (i & 3);  // 8-byte granule index inside the k-row
        bf16x8_t out;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int kr = kb + h * 4 + (i >> 2);
            const int f = ((kr & 3) << 1) | (kr & 8);
            const int off = kr * 256 + ((((q >> 1) ^ f)) << 4) + (q & 1) * 8;
            bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4_t, const_cast<char*>(lds) + off));
            out[h * 4 + 0] = v[0]; out[h * 4 + 1] = v[1]; out[h * 4 + 2] = v[2]; out[h * 4 + 3] = v[3];
        }
        return out;
    }
}

// ---- epilogue for the swapped-operand accumulators: lane holds m = lane & 15, n = (lane >> 4) * 4 + r of each 16x16 fragment
template <typename TC, typename TBIAS>
__device__ __forceinline__ void store_frag(const f32x4& acc, TC* C, int64_t ldc, int m, int n, float alpha, float beta, const void* bias) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = alpha * acc[r];
    if (bias) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] += ldf((const TBIAS*)bias + n + r);
    }
    TC* c = C + (int64_t)m * ldc + n;
    if (sizeof(TC) == 4) {
        if (beta != 0.f) {
            float4 o = *reinterpret_cast<const float4*>(c);
            v[0] += beta * o.x; v[1] += beta * o.y; v[2] += beta * o.z; v[3] += beta * o.w;
        }
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        if (beta != 0.f) {
            uint2 o = *reinterpret_cast<const uint2*>(c);
            v[0] += beta * __uint_as_float(o.x << 16); v[1] += beta * __uint_as_float(o.x & 0xffff0000u);
            v[2] += beta * __uint_as_float(o.y << 16); v[3] += beta * __uint_as_float(o.y & 0xffff0000u);
        }
        uint2 o;
        o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(c) = o;
    }
}

// ---- bf16 epilogue of a 128 x 64 wave tile (8 x 4 swapped-layout fragments) through a 16 KiB LDS region private to the wave.
// The per-fragment store (store_frag: 8 bytes per lane = 16 rows x 32-byte pieces per instruction) made the epilogue store-ISSUE bound:
// 256 store instructions per 256 x 256 tile at ~60 cycles each = 13-20 % of a K = 2048 GEMM, exposed because the tile's workgroup is
// alone on its CU (measured by skipping the stores: ff1 NT 1867 -> 1625 us).  Staged, a wave issues 16 stores of 8 rows x one full
// 128-byte line.  LDS image: row r at r * 128 bytes, 8-byte granule q at q ^ (r & 14) (conflict-free for the fragment writes and the
// 16-byte row reads, and it keeps the two granules of a 16-byte chunk in order).  beta must be 0 (the staged value is already rounded).
typedef __attribute__((ext_vector_type(2))) unsigned gt_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned gt_u32x4;
template <typename TBIAS>
__device__ __forceinline__ void store_wave_tile_bf16(const f32x4 (&acc)[8][4], char* lds_wave, bf16_t* C, int64_t ldc, int m_base, int n_base,
                                                     float alpha, const void* bias, int lane) {
    const int m = lane & 15, g = lane >> 4;
    const unsigned w0 = (unsigned)(size_t)LDS_PTR(char, lds_wave);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; r++) b[r] = ldf((const TBIAS*)bias + n_base + j * 16 + g * 4 + r);
        }
        const unsigned wa = w0 + m * 128 + (((4 * j + g) ^ (m & 14)) << 3);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            gt_u32x2 o;
            o[0] = f2bf_pk(alpha * acc[i][j][0] + b[0], alpha * acc[i][j][1] + b[1]);
            o[1] = f2bf_pk(alpha * acc[i][j][2] + b[2], alpha * acc[i][j][3] + b[3]);
            *(__attribute__((address_space(3))) gt_u32x2*)(size_t)(wa + i * 2048) = o;
        }
    }
    const int rr = lane >> 3, c = lane & 7;
    bf16_t* dst = C + (int64_t)(m_base + rr) * ldc + n_base + c * 8;
#pragma unroll
    for (int it = 0; it < 16; it++) {  // rows 8 it + rr:  (8 it + rr) & 14 == rr & 14 | (8 it & 14) -> the XOR term changes with it & 1
        const unsigned a = w0 + (8 * it + rr) * 128 + (((2 * c) ^ ((8 * it + rr) & 14)) << 3);
        const gt_u32x4 v = *(__attribute__((address_space(3))) const gt_u32x4*)(size_t)a;
        *reinterpret_cast<gt_u32x4*>(dst + (int64_t)(8 * it) * ldc) = v;
    }
}

// The same for beta != 0 (bf16 C is read, scaled and added before the ONE rounding): the alpha * acc + bias values are staged in fp32,
// 64 rows at a time (16 KiB per wave: row r at r * 256 bytes, 16-byte chunk q at q ^ (r & 15)), and every lane then handles 4 consecutive
// columns of a row: 8-byte C loads and stores that cover 4 rows x one full 128-byte line per instruction.
template <typename TBIAS, int I0>
__device__ __forceinline__ void store_wave_half_bf16_beta(const f32x4 (&acc)[8][4], char* lds_wave, bf16_t* C, int64_t ldc, int m_base, int n_base,
                                                          float alpha, float beta, const void* bias, int lane) {
    const int m = lane & 15, g = lane >> 4;
    const unsigned w0 = (unsigned)(size_t)LDS_PTR(char, lds_wave);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; r++) b[r] = ldf((const TBIAS*)bias + n_base + j * 16 + g * 4 + r);
        }
        const unsigned wa = w0 + m * 256 + (((4 * j + g) ^ m) << 4);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = alpha * acc[I0 + i][j][r] + b[r];
            *(__attribute__((address_space(3))) f32x4*)(size_t)(wa + i * 4096) = o;
        }
    }
    const int rr = lane >> 4, c = lane & 15;
    bf16_t* dst = C + (int64_t)(m_base + I0 * 16 + rr) * ldc + n_base + c * 4;
    const int64_t step = 4 * ldc;
#pragma unroll 4
    for (int it = 0; it < 16; it++) {  // rows 4 it + rr
        const int row = 4 * it + rr;
        const f32x4 v = *(__attribute__((address_space(3))) const f32x4*)(size_t)(w0 + row * 256 + ((c ^ (row & 15)) << 4));
        const gt_u32x2 o = *reinterpret_cast<const gt_u32x2*>(dst);
        gt_u32x2 w;
        w[0] = f2bf_pk(v[0] + beta * __uint_as_float(o[0] << 16), v[1] + beta * __uint_as_float(o[0] & 0xffff0000u));
        w[1] = f2bf_pk(v[2] + beta * __uint_as_float(o[1] << 16), v[3] + beta * __uint_as_float(o[1] & 0xffff0000u));
        *reinterpret_cast<gt_u32x2*>(dst) = w;
        dst += step;
    }
}

// XCD-aware tile walk: hardware block ids round-robin over the 8 XCDs (private L2s); give each XCD a contiguous span of
// tiles, walked column-major inside bands of `band` tile-rows so neighbouring workgroups share A and B panels in L2.
__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int band, int& tm, int& tn) {
    const int nblk = tiles_m * tiles_n;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tiles_per_band = band * tiles_n;
    const int b0 = bid / tiles_per_band, rem = bid % tiles_per_band;
    const int band_rows = (tiles_m - b0 * band) < band ? (tiles_m - b0 * band) : band;
    tm = b0 * band + rem % band_rows;
    tn = rem / band_rows;
}

int db1_gemm_tile256_launch(const GemmTileArgs& t, int fa, int fb, int dtC, int dtBias, int batch, hipStream_t st);
int db1_gemm_pp_launch(const GemmTileArgs& t, int fa, int fb, int dtC, int dtBias, int batch, hipStream_t st);
bool db1_gemm_w4_supported(const GemmTileArgs& t, int fa, int fb, int dtC, int batch);
int db1_gemm_w4_launch(const GemmTileArgs& t, int fa, int fb, int dtC, int dtBias, int batch, hipStream_t st);
bool db1_gemm_w4n_supported(const GemmTileArgs& t, int fa, int fb, int dtC, int batch);
int db1_gemm_w4n_launch(const GemmTileArgs& t, int fa, int fb, int dtC, int dtBias, int batch, hipStream_t st);
int db1_gemm_pp32_launch(const GemmTileArgs& t, int fa, int fb, int dtC, int dtBias, int batch, hipStream_t st);
bool db1_gemm_w4_geglu_supported(int M, int dff, int K, int64_t lda, int64_t ldw, int64_t ldz, int64_t ld_other, bool fwd);
int db1_gemm_w4_geglu_fwd_launch(const GemmTileArgs& t, int dtBias, hipStream_t st);
int db1_gemm_w4_geglu_bwd_launch(const GemmTileArgs& t, hipStream_t st);
int db1_gemm_skinny_launch(const bf16_t* x, const bf16_t* w, void* y, const void* bias, int M, int N, int K, int64_t ldx, int64_t ldw,
                           int64_t ldy, float alpha, float beta, int dtC, int dtBias, hipStream_t st);
