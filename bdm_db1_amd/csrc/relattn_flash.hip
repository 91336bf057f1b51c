// Relative-position flash attention for gfx950 (bf16, d_head = 128): Transformer-XL scores
//     s[i,j] = ((q_i+u).k_j + (q_i+v).R[i-j]) / sqrt(d),   visible iff  i - shift < j <= i
// (closed form of AC + _rel_shift(BD) + mask, transformer_xl.py:98-110,160-209,551-567) with online softmax
// and P.V fused, never materialising an (L x L) tensor in HBM in the forward.
//
// Common machinery (all three kernels; v_mfma_f32_32x32x16_bf16, 4 waves x 32 rows per workgroup, 32-column blocks):
//   * "swapped" products put the query (fwd, bwd_q) or the key (bwd_kv) on the LANE axis of the accumulator, so
//     per-row softmax statistics are lane-local and P / dS feed the next MFMA as the B operand straight from registers;
//   * the relative term T = Qv.Rband^T is computed for the 64 distances a 32x32 block can touch, written to a per-wave
//     LDS scratch [32 q][64 dist] and read back SKEWED (element (a, a - b + 31)); write (lanes = consecutive distances)
//     and read (lane stride 65 or -1 words) are both bank-conflict free;
//   * transposed operands (V^T, K^T, dO^T, Qu^T) come from row-major LDS tiles through ds_read_b64_tr_b16;
//   * tiles and a 256-row ring of R rows (the band of distances slides by 32 per block) are staged with global_load_lds
//     (16 B/lane, lane-linear destination); 16-B chunks are XOR-swizzled on the SOURCE side with
//     swz(row) = ((row & 3) << 2) | ((row >> 2) & 3), which makes BOTH the ds_read_b128 row fragments and the tr reads
//     conflict-free on the same image.
// Backward = delta pre-pass + two kernels without atomics (deterministic):
//   bwd_q : per 128 queries, loop keys  -> dq_k = dS.K (the (q+u).k branch) and dT = dS re-indexed by distance (bf16, HBM);
//   bwd_kv: per 128 keys,    loop queries -> dV = P^T.dO, dK = dS^T.Qu.
// dq_r = dT.R and dR = dT^T.Qv are plain batched GEMMs on dT (exact causal FLOPs, no band overhead) run by the caller.
// Inputs qu = q+u and qv = q+v_bias are materialised once per layer by db1_relattn_add_head_bias.
#include "relattn_flash.h"
#include <cstdlib>
// transposed A-fragment from a row-major [row][128] tile: d-block db (32 columns), 16 tile rows starting at row0.
// slot t of lane (d = lane & 31, hb) <-> tile row row0 + (t & 3) + 8 * (t >> 2) + 4 * hb == the C-layout row order.
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int row0, int db, int lane) {
    const int g4 = lane >> 4, t = lane & 15, hb = g4 >> 1;
    const int gran = ((32 * db + 16 * (g4 & 1)) >> 2) + (t & 3);  // 8-byte granule inside the 256-B row
    bf16x8_t out;
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
        const int row = row0 + 4 * hb + 8 * h2 + (t >> 2);
        const int off = row * 256 + (((gran >> 1) ^ swz(row)) << 4) + (gran & 1) * 8;
        bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4_t, const_cast<char*>(tile) + off));
        out[h2 * 4 + 0] = v[0]; out[h2 * 4 + 1] = v[1]; out[h2 * 4 + 2] = v[2]; out[h2 * 4 + 3] = v[3];
    }
    return out;
}
// fragment "tile row `row`, k = ks*16 + (lane>>5)*8 .. +8" (A or B operand image) from a swizzled row-major tile
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int row, int ks, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(tile + row * 256 + (((ks * 2 + (lane >> 5)) ^ swz(row)) << 4));
}
__device__ __forceinline__ void zero16(f32x16& x) {
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = 0.f;
}
// acc^T [4 d-blocks](rows = d, col = lane) -> bf16 rows [32][128] at dst (row stride rs), staged through the wave scratch
__device__ __forceinline__ void store_acc_t(const f32x16* acc, float mul, bf16_t* Ow, bf16_t* dst, int64_t rs, int lane) {
    const int a = lane & 31, hb = lane >> 5;
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
            uint2 o;
            o.x = pk_bf16(acc[db][rq * 4 + 0] * mul, acc[db][rq * 4 + 1] * mul);
            o.y = pk_bf16(acc[db][rq * 4 + 2] * mul, acc[db][rq * 4 + 3] * mul);
            *reinterpret_cast<uint2*>(Ow + a * 136 + 32 * db + 8 * rq + 4 * hb) = o;
        }
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int row = it * 4 + (lane >> 4), ch = lane & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(Ow + row * 136 + ch * 8);
        *reinterpret_cast<uint4*>(dst + (int64_t)row * rs + ch * 8) = v;
    }
}
// lane-constant LDS byte offsets (computed once per kernel: the loops then issue LDS reads with almost no address VALU)
struct LaneOffs {
    int row[8];    // row_frag of tile row (lane & 31):   a*256 + (((ks*2 + hb) ^ swz(a)) << 4)
    int ring[8];   // chunk part of a ring row_frag:      ((ks*2 + hb) ^ swz((a + 1) & 15)) << 4   (slot & 15 never changes)
    int tr[4][2];  // tr_frag(tile, row0 = 0, db) halves: add 4096 for row0 = 16
};
__device__ __forceinline__ void make_offs(LaneOffs& o, int lane) {
    const int a = lane & 31, hb = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
        o.row[ks] = a * 256 + (((ks * 2 + hb) ^ swz(a)) << 4);
        o.ring[ks] = ((ks * 2 + hb) ^ swz((a + 1) & 15)) << 4;
    }
    const int g4 = lane >> 4, t = lane & 15, hb4 = g4 >> 1;
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
            const int gran = ((32 * db + 16 * (g4 & 1)) >> 2) + (t & 3);
            const int row = 4 * hb4 + 8 * h2 + (t >> 2);
            o.tr[db][h2] = row * 256 + (((gran >> 1) ^ swz(row)) << 4) + (gran & 1) * 8;
        }
}
__device__ __forceinline__ bf16x8_t rowf(const char* tile, const LaneOffs& o, int ks) { return *reinterpret_cast<const bf16x8_t*>(tile + o.row[ks]); }
__device__ __forceinline__ bf16x8_t trf(const char* tile, const LaneOffs& o, int row0, int db) {
    bf16x8_t out;
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
        bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4_t, const_cast<char*>(tile) + o.tr[db][h2] + row0 * 256));
        out[h2 * 4 + 0] = v[0]; out[h2 * 4 + 1] = v[1]; out[h2 * 4 + 2] = v[2]; out[h2 * 4 + 3] = v[3];
    }
    return out;
}
// T = Arows . Rband^T for the 64 distances starting at dist_lo (dist_lo + a == 1 mod 16 by construction), into Tw[32][64]
template <bool A_REGS>
__device__ __forceinline__ void rel_band_to_lds(const bf16x8_t* fa_regs, const char* a_tile, const LaneOffs& o, const char* ring, int dist_lo,
                                                float* Tw, int lane) {
    const int a = lane & 31, hb = lane >> 5;
    const int rr0 = ((dist_lo + a) & (FA_RING - 1)) << 8;
    float* tw = Tw + hb * 256 + a;
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        f32x16 acc_t;
        zero16(acc_t);
        const char* rrow = ring + ((rr0 + blk * 8192) & (FA_RING * 256 - 1));
#pragma unroll
        for (int ks = 0; ks < 8; ks++)
            acc_t = MFMA32(A_REGS ? fa_regs[ks] : rowf(a_tile, o, ks), *reinterpret_cast<const bf16x8_t*>(rrow + o.ring[ks]), acc_t);
#pragma unroll
        for (int r = 0; r < 16; r++) tw[((r & 3) + 8 * (r >> 2)) * 64 + 32 * blk] = acc_t[r];
    }
}
// Incremental form for the forward, whose waves keep their 32 queries and walk the key blocks upwards (in bwd_kv the queries
// change every block, so nothing carries over; bwd_q reuses the scratch for the dS re-indexing): the band of block jb+1 is the band of block jb moved down
// by 32 distances, so its upper half is the lower half just computed.  The scratch is used as a 2-slot ring with an XOR parity:
// logical column c (0..63, distance dist_lo + c) lives at physical column c ^ (32 * PAR); consecutive processed blocks alternate
// PAR (= the unrolled block instance), so the previous block's lower half IS this block's upper half without moving anything.
// Only the new lower 32 distances are computed (8 MFMAs instead of 16, half the scratch writes and ring reads) unless the wave
// did not process the previous block (`both`).
template <int PAR>
__device__ __forceinline__ void rel_band_incr_to_lds(const bf16x8_t* fa_regs, const LaneOffs& o, const char* ring, int dist_lo, float* Tw, int lane,
                                                     bool both) {
    const int a = lane & 31, hb = lane >> 5;
    const int rr0 = ((dist_lo + a) & (FA_RING - 1)) << 8;
    float* tw = Tw + hb * 256 + a;
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {  // blk 0 = the new lower 32 distances; blk 1 only for a wave's first processed block
        if (blk == 1 && !both) break;
        f32x16 acc_t;
        zero16(acc_t);
        const char* rrow = ring + ((rr0 + blk * 8192) & (FA_RING * 256 - 1));
#pragma unroll
        for (int ks = 0; ks < 8; ks++) acc_t = MFMA32(fa_regs[ks], *reinterpret_cast<const bf16x8_t*>(rrow + o.ring[ks]), acc_t);
        const int phys = 32 * (blk ^ PAR);  // logical half blk lives in physical half blk ^ PAR
#pragma unroll
        for (int r = 0; r < 16; r++) tw[((r & 3) + 8 * (r >> 2)) * 64 + phys] = acc_t[r];
    }
}

// ======================================================================================= forward
#define FWD_OFF_K 0          // two stages of 8 KiB
#define FWD_OFF_V 16384      // two stages of 8 KiB
#define FWD_OFF_R 32768
#define FWD_OFF_T (32768 + FA_RING * 256)
#define FWD_LDS_BYTES (FWD_OFF_T + 4 * FA_TW_BYTES)
#define BQ_WAVE_BYTES 16384                 // bwd_q: two [32][64] f32 scratches per wave (also holds the [32][136] bf16 output tile)
#define BQ_LDS_BYTES (FWD_OFF_T + 4 * BQ_WAVE_BYTES)   // = 160 KiB, all of the LDS

__global__ __launch_bounds__(256, 1) void relattn_flash_fwd_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rank, h, b;
    if (!flash_wg_coords(p.L / FA_BQ, p.H, p.B, rank, h, b)) return;
    const int qt = p.L / FA_BQ - 1 - rank;  // late query tiles have the longest key loops
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int i0 = qt * FA_BQ, iw = i0 + 32 * wave;
    const int a = lane & 31, hb = lane >> 5;
    char* Ks0 = smem + FWD_OFF_K;
    char* Vs0 = smem + FWD_OFF_V;
    char* Rr = smem + FWD_OFF_R;
    float* Tw = reinterpret_cast<float*>(smem + FWD_OFF_T + wave * FA_TW_BYTES);
    const float* twr = Tw + a * 65 + 31 - 4 * hb;  // skewed read base: element (a, a - crow(r,hb) + 31)
    const bf16_t* qu = p.qu + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* qv = p.qv + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* kg = p.k + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* vg = p.v + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* Rg = p.R + h * FA_D;
    LaneOffs offs;
    make_offs(offs, lane);

    bf16x8_t fqu[8], fqv[8];  // row iw + a, k = ks*16 + hb*8 (A and B operand images coincide)
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
        fqu[ks] = *reinterpret_cast<const bf16x8_t*>(qu + (int64_t)(iw + a) * HD + ks * 16 + hb * 8);
        fqv[ks] = *reinterpret_cast<const bf16x8_t*>(qv + (int64_t)(iw + a) * HD + ks * 16 + hb * 8);
    }
    int jlo = i0 - p.shift + 1;
    if (jlo < 0) jlo = 0;
    const int jb_lo = jlo / FA_BK, jb_hi = (i0 + FA_BQ - 1) / FA_BK;
    // prologue: ring rows for distances [i0-j0lo-32, i0-j0lo+128) and the first K/V tiles
    for (int c4 = -1; c4 < 4; c4++) stage_ring32(Rg, HD, i0 - jb_lo * FA_BK + 32 * c4, L, Rr, wave, lane);
    stage_tile32(kg, p.kv_rs, jb_lo * FA_BK, Ks0, wave, lane);
    stage_tile32(vg, p.kv_rs, jb_lo * FA_BK, Vs0, wave, lane);
    f32x16 acc_o[4];
#pragma unroll
    for (int db = 0; db < 4; db++) zero16(acc_o[db]);
    const float c2 = p.scale * LOG2E;
    float m_i = -1.0e30f, l_i = 0.f;  // m_i in RAW score units (before the 1/sqrt(d) scale)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), as a builtin so that hipcc's own bookkeeping sees the fragment loads retired
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int toff1[16];  // odd-parity read offsets: a * 64 + ((a - crow(r, hb) + 31) ^ 32)
#pragma unroll
    for (int r = 0; r < 16; r++) toff1[r] = a * 64 + ((a - crow(r, hb) + 31) ^ 32);
    bool have_prev = false;  // did this wave process the previous key block?  (wave-uniform)
    auto block = [&](auto CUR, int jb) {
        constexpr int cur = decltype(CUR)::value;
        const int j0 = jb * FA_BK;
        const char* Ks = Ks0 + cur * 8192;
        const char* Vs = Vs0 + cur * 8192;
        if (jb < jb_hi) {  // prefetch the next block's tiles and ring rows; they land while this block is computed
            stage_tile32(kg, p.kv_rs, j0 + FA_BK, Ks0 + (cur ^ 1) * 8192, wave, lane);
            stage_tile32(vg, p.kv_rs, j0 + FA_BK, Vs0 + (cur ^ 1) * 8192, wave, lane);
            stage_ring32(Rg, HD, i0 - j0 - 64, L, Rr, wave, lane);
        }
        if (j0 > iw + 31 || j0 + 31 <= iw - p.shift) have_prev = false;
        else {  // wave-uniform: blocks entirely outside this wave's window are skipped
            f32x16 acc_s;
            zero16(acc_s);
#pragma unroll
            for (int ks = 0; ks < 8; ks++) acc_s = MFMA32(rowf(Ks, offs, ks), fqu[ks], acc_s);  // S^T[key][query]
            rel_band_incr_to_lds<cur>(fqv, offs, Rr, iw - j0 - 31, Tw, lane, !have_prev);
            have_prev = true;
            float s[16];  // this lane: query iw+a; register r: key j0+crow(r,hb); T element (a, (a - crow + 31) ^ (32 * cur))
#pragma unroll
            for (int r = 0; r < 16; r++) s[r] = acc_s[r] + (cur == 0 ? twr[-((r & 3) + 8 * (r >> 2))] : Tw[toff1[r]]);
            if (j0 + 31 > iw || j0 <= iw + 31 - p.shift) {  // only diagonal / window-edge blocks need the element mask
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int i = iw + a, j = j0 + crow(r, hb);
                    s[r] = ((j <= i) && (j > i - p.shift)) ? s[r] : -1.0e30f;
                }
            }
            float mblk = s[0];
#pragma unroll
            for (int r = 1; r < 16; r++) mblk = fmaxf(mblk, s[r]);
            mblk = fmaxf(mblk, __shfl_xor(mblk, 32, 64));
            const float m_new = fmaxf(m_i, mblk);
            if (!__all(m_new == m_i)) {  // the running maxima rarely move after the first blocks: skip the O-wide rescale then
                const float alpha = __builtin_amdgcn_exp2f((m_i - m_new) * c2);
                l_i *= alpha;
#pragma unroll
                for (int db = 0; db < 4; db++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc_o[db][r] *= alpha;
                m_i = m_new;
            }
            const float mc = -m_i * c2;
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, mc)); rs += s[r]; }
            rs += __shfl_xor(rs, 32, 64);
            l_i += rs;
            const bf16x8_t pb0 = pack8(s), pb1 = pack8(s + 8);
#pragma unroll
            for (int db = 0; db < 4; db++) {  // O^T[d][query] += V^T . P^T
                acc_o[db] = MFMA32(trf(Vs, offs, 0, db), pb0, acc_o[db]);
                acc_o[db] = MFMA32(trf(Vs, offs, 16, db), pb1, acc_o[db]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // next tiles have landed; every wave is done reading the current ones
    };
    for (int jb = jb_lo; jb <= jb_hi; jb += 2) {
        block(std::integral_constant<int, 0>{}, jb);
        if (jb + 1 <= jb_hi) block(std::integral_constant<int, 1>{}, jb + 1);
    }
    store_acc_t(acc_o, 1.f / l_i, reinterpret_cast<bf16_t*>(Tw), p.o + ((int64_t)b * L + iw) * HD + h * FA_D, HD, lane);
    if (hb == 0) p.lse_out[((int64_t)b * H + h) * L + iw + a] = m_i * p.scale + logf(l_i);
}

// ======================================================================================= backward: delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void relattn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, float* __restrict__ delta,
                                                            int64_t n_rows, int L, int H) {
    // 16 lanes per (b, i, h) row of 128 elements (16 B per lane), 16 rows per 256-thread block
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= n_rows) return;
    const int sub = threadIdx.x & 15;
    const uint4 x = *reinterpret_cast<const uint4*>(o + row * FA_D + sub * 8);
    const uint4 y = *reinterpret_cast<const uint4*>(dout + row * FA_D + sub * 8);
    const unsigned xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
        s += __uint_as_float(xs[t] << 16) * __uint_as_float(ys[t] << 16) + __uint_as_float(xs[t] & 0xffff0000u) * __uint_as_float(ys[t] & 0xffff0000u);
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if (sub == 0) {
        const int64_t bi = row / H;
        const int hh = (int)(row - bi * H);
        const int64_t bb = bi / L, i = bi - bb * L;
        delta[(bb * H + hh) * L + i] = s;
    }
}

// ======================================================================================= backward w.r.t. queries (+ dT)
__global__ __launch_bounds__(256, 1) void relattn_flash_bwd_q_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rank, h, b;
    if (!flash_wg_coords(p.L / FA_BQ, p.H, p.B, rank, h, b)) return;
    const int qt = p.L / FA_BQ - 1 - rank;
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int i0 = qt * FA_BQ, iw = i0 + 32 * wave;
    const int a = lane & 31, hb = lane >> 5;
    char* Ks0 = smem + FWD_OFF_K;
    char* Vs0 = smem + FWD_OFF_V;
    char* Rr = smem + FWD_OFF_R;
    // per-wave scratch: T ring [32][64] f32 (kept across blocks: incremental band, see rel_band_incr_to_lds) + a second [32][64] for
    // the dS re-indexing (the two used to share one scratch, which forced the full 64-distance band every block)
    float* Tw = reinterpret_cast<float*>(smem + FWD_OFF_T + wave * BQ_WAVE_BYTES);
    float* Dw = Tw + 2048;
    const float* twr = Tw + a * 65 + 31 - 4 * hb;
    float* dwr = Dw + a * 65 + 31 - 4 * hb;
    const bf16_t* qu = p.qu + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* qv = p.qv + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* dog = p.dout + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* kg = p.k + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* vg = p.v + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* Rg = p.R + h * FA_D;
    bf16_t* dTg = p.dT + (((int64_t)h * p.B + b) * L) * L;
    LaneOffs offs;
    make_offs(offs, lane);

    bf16x8_t fqu[8], fqv[8], fdo[8];
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
        fqu[ks] = *reinterpret_cast<const bf16x8_t*>(qu + (int64_t)(iw + a) * HD + ks * 16 + hb * 8);
        fqv[ks] = *reinterpret_cast<const bf16x8_t*>(qv + (int64_t)(iw + a) * HD + ks * 16 + hb * 8);
        fdo[ks] = *reinterpret_cast<const bf16x8_t*>(dog + (int64_t)(iw + a) * HD + ks * 16 + hb * 8);
    }
    const float c2 = p.scale * LOG2E;
    const float nlse2 = -p.lse[((int64_t)b * H + h) * L + iw + a] * LOG2E;
    const float delta_a = p.delta[((int64_t)b * H + h) * L + iw + a];
    int jlo = i0 - p.shift + 1;
    if (jlo < 0) jlo = 0;
    const int jb_lo = jlo / FA_BK, jb_hi = (i0 + FA_BQ - 1) / FA_BK;
    for (int c4 = -1; c4 < 4; c4++) stage_ring32(Rg, HD, i0 - jb_lo * FA_BK + 32 * c4, L, Rr, wave, lane);
    stage_tile32(kg, p.kv_rs, jb_lo * FA_BK, Ks0, wave, lane);
    stage_tile32(vg, p.kv_rs, jb_lo * FA_BK, Vs0, wave, lane);
    f32x16 acc_dq[4];
#pragma unroll
    for (int db = 0; db < 4; db++) zero16(acc_dq[db]);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), as a builtin so that hipcc's own bookkeeping sees the fragment loads retired
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int toff1[16];  // odd-parity read offsets of the T ring: a * 64 + ((a - crow(r, hb) + 31) ^ 32)
#pragma unroll
    for (int r = 0; r < 16; r++) toff1[r] = a * 64 + ((a - crow(r, hb) + 31) ^ 32);
    bool have_prev = false;  // did this wave process the previous key block?  (wave-uniform)
    auto block = [&](auto CUR, int jb) {
        constexpr int cur = decltype(CUR)::value;
        const int j0 = jb * FA_BK;
        const char* Ks = Ks0 + cur * 8192;
        const char* Vs = Vs0 + cur * 8192;
        if (jb < jb_hi) {
            stage_tile32(kg, p.kv_rs, j0 + FA_BK, Ks0 + (cur ^ 1) * 8192, wave, lane);
            stage_tile32(vg, p.kv_rs, j0 + FA_BK, Vs0 + (cur ^ 1) * 8192, wave, lane);
            stage_ring32(Rg, HD, i0 - j0 - 64, L, Rr, wave, lane);
        }
        if (j0 > iw + 31 || j0 + 31 <= iw - p.shift) have_prev = false;
        else {
            f32x16 acc_s, acc_dp;
            zero16(acc_s);
            zero16(acc_dp);
#pragma unroll
            for (int ks = 0; ks < 8; ks++) acc_s = MFMA32(rowf(Ks, offs, ks), fqu[ks], acc_s);    // S^T[key][query]
#pragma unroll
            for (int ks = 0; ks < 8; ks++) acc_dp = MFMA32(rowf(Vs, offs, ks), fdo[ks], acc_dp);  // dP^T[key][query]
            rel_band_incr_to_lds<cur>(fqv, offs, Rr, iw - j0 - 31, Tw, lane, !have_prev);
            have_prev = true;
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; r++)
                ds[r] = __builtin_amdgcn_exp2f(fmaf(acc_s[r] + (cur == 0 ? twr[-((r & 3) + 8 * (r >> 2))] : Tw[toff1[r]]), c2, nlse2));
            if (j0 + 31 > iw || j0 <= iw + 31 - p.shift) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int i = iw + a, j = j0 + crow(r, hb);
                    ds[r] = ((j <= i) && (j > i - p.shift)) ? ds[r] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) ds[r] = ds[r] * (acc_dp[r] - delta_a) * p.scale;
            // dS re-indexed by distance: write into the scratch at (a, a - b + 31), then rows go out as 32 contiguous bf16
#pragma unroll
            for (int r = 0; r < 16; r++) dwr[-((r & 3) + 8 * (r >> 2))] = ds[r];
            const bf16x8_t db0 = pack8(ds), db1 = pack8(ds + 8);
#pragma unroll
            for (int db = 0; db < 4; db++) {  // dq^T[d][query] += K^T . dS^T
                acc_dq[db] = MFMA32(trf(Ks, offs, 0, db), db0, acc_dq[db]);
                acc_dq[db] = MFMA32(trf(Ks, offs, 16, db), db1, acc_dq[db]);
            }
            {
                const int t = lane & 31;
                bf16_t* drow = dTg + (int64_t)(iw + hb) * L + (iw + hb - j0 - 31 + t);  // row = 2*it + hb, dist = i - j with j = j0 + 31 - t
                const float* trow = Dw + hb * 65 + t;
#pragma unroll
                for (int it = 0; it < 16; it++) {
                    const int dist = iw + 2 * it + hb - j0 - 31 + t;
                    if (dist >= 0) drow[(int64_t)(2 * it) * (L + 1)] = f2bf(trow[2 * it * 65]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int jb = jb_lo; jb <= jb_hi; jb += 2) {
        block(std::integral_constant<int, 0>{}, jb);
        if (jb + 1 <= jb_hi) block(std::integral_constant<int, 1>{}, jb + 1);
    }
    store_acc_t(acc_dq, 1.f, reinterpret_cast<bf16_t*>(Tw), p.dq + (int64_t)b * p.dq_bs + (int64_t)iw * p.dq_rs + h * FA_D, p.dq_rs, lane);
}

// ======================================================================================= backward w.r.t. keys / values
#define KV_OFF_QU 0                          // two stages of 8 KiB each for Qu, Qv, dO
#define KV_OFF_QV 16384
#define KV_OFF_DO 32768
#define KV_OFF_ST 49152                      // two stages of {lse[32], delta[32]} floats (raw, via LDS-DMA)
#define KV_OFF_R 49664
#define KV_OFF_T (KV_OFF_R + FA_RING * 256)
#define KV_LDS_BYTES (KV_OFF_T + 4 * FA_TW_BYTES)

__global__ __launch_bounds__(256, 1) void relattn_flash_bwd_kv_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int kt, h, b;  // early key tiles see the most queries: rank == tile index
    if (!flash_wg_coords(p.L / FA_BQ, p.H, p.B, kt, h, b)) return;
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int j0 = kt * FA_BQ, kw = j0 + 32 * wave;
    const int a = lane & 31, hb = lane >> 5;   // a = key column of this lane
    char* Qus0 = smem + KV_OFF_QU;
    char* Qvs0 = smem + KV_OFF_QV;
    char* dOs0 = smem + KV_OFF_DO;
    float* stat0 = reinterpret_cast<float*>(smem + KV_OFF_ST);
    char* Rr = smem + KV_OFF_R;
    float* Tw = reinterpret_cast<float*>(smem + KV_OFF_T + wave * FA_TW_BYTES);
    const float* twr = Tw + 4 * hb * 65 + 31 - a;  // element (aq, aq - a + 31) with aq = crow(r, hb)
    const bf16_t* qu = p.qu + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* qv = p.qv + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* dog = p.dout + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* kg = p.k + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* vg = p.v + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* Rg = p.R + h * FA_D;
    const float* lseg = p.lse + ((int64_t)b * H + h) * L;
    const float* delg = p.delta + ((int64_t)b * H + h) * L;
    LaneOffs offs;
    make_offs(offs, lane);

    bf16x8_t fk[8], fv[8];  // B-operand images: column = key kw + a
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
        fk[ks] = *reinterpret_cast<const bf16x8_t*>(kg + (int64_t)(kw + a) * p.kv_rs + ks * 16 + hb * 8);
        fv[ks] = *reinterpret_cast<const bf16x8_t*>(vg + (int64_t)(kw + a) * p.kv_rs + ks * 16 + hb * 8);
    }
    const float c2 = p.scale * LOG2E;
    const int ib_lo = j0 / FA_BK;
    int ihi = j0 + FA_BQ - 1 + p.shift - 1;  // last query that can see the last key of the tile
    if (ihi > L - 1) ihi = L - 1;
    const int ib_hi = ihi / FA_BK;
    // ring: distances [i0q - j0 - 128, i0q - j0 + 32) for the first block; every block prefetches the next 32
    for (int c4 = 0; c4 < 5; c4++) stage_ring32(Rg, HD, ib_lo * FA_BK - j0 - 128 + 32 * c4, L, Rr, wave, lane);
    stage_tile32(qu, HD, ib_lo * FA_BK, Qus0, wave, lane);
    stage_tile32(qv, HD, ib_lo * FA_BK, Qvs0, wave, lane);
    stage_tile32(dog, HD, ib_lo * FA_BK, dOs0, wave, lane);
    if (wave == 0) glds_stat(lseg, delg, ib_lo * FA_BK, stat0, lane);
    f32x16 acc_dk[4], acc_dv[4];
#pragma unroll
    for (int db = 0; db < 4; db++) { zero16(acc_dk[db]); zero16(acc_dv[db]); }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), as a builtin so that hipcc's own bookkeeping sees the fragment loads retired
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto block = [&](auto CUR, int ib) {
        constexpr int cur = decltype(CUR)::value;
        const int i0q = ib * FA_BK;
        const char* Qus = Qus0 + cur * 8192;
        const char* Qvs = Qvs0 + cur * 8192;
        const char* dOs = dOs0 + cur * 8192;
        const float* stat = stat0 + cur * 64 + 4 * hb;
        if (ib < ib_hi) {  // prefetch the next query block
            const int nx = i0q + FA_BK;
            stage_tile32(qu, HD, nx, Qus0 + (cur ^ 1) * 8192, wave, lane);
            stage_tile32(qv, HD, nx, Qvs0 + (cur ^ 1) * 8192, wave, lane);
            stage_tile32(dog, HD, nx, dOs0 + (cur ^ 1) * 8192, wave, lane);
            stage_ring32(Rg, HD, nx - j0, L, Rr, wave, lane);
            if (wave == 0) glds_stat(lseg, delg, nx, stat0 + (cur ^ 1) * 64, lane);
        }
        if (!(i0q + 31 < kw || i0q >= kw + 31 + p.shift)) {  // some (i, j) of this block pair is visible
            f32x16 acc_s, acc_dp;
            zero16(acc_s);
            zero16(acc_dp);
#pragma unroll
            for (int ks = 0; ks < 8; ks++) acc_s = MFMA32(rowf(Qus, offs, ks), fk[ks], acc_s);    // S[query][key]
#pragma unroll
            for (int ks = 0; ks < 8; ks++) acc_dp = MFMA32(rowf(dOs, offs, ks), fv[ks], acc_dp);  // dP[query][key]
            rel_band_to_lds<false>(nullptr, Qvs, offs, Rr, i0q - kw - 31, Tw, lane);  // (queries change every block: nothing to reuse)
            float pr[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {  // this lane: key kw + a; register r: query i0q + crow(r, hb)
                const int q8 = (r & 3) + 8 * (r >> 2);
                pr[r] = __builtin_amdgcn_exp2f(fmaf(acc_s[r] + twr[q8 * 65], c2, -LOG2E * stat[q8]));
            }
            if (i0q < kw + 31 || i0q + 31 >= kw + p.shift) {  // diagonal / window-edge block pairs need the element mask
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int i = i0q + crow(r, hb), j = kw + a;
                    pr[r] = ((j <= i) && (j > i - p.shift)) ? pr[r] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) ds[r] = pr[r] * (acc_dp[r] - stat[32 + (r & 3) + 8 * (r >> 2)]) * p.scale;
            const bf16x8_t pb0 = pack8(pr), pb1 = pack8(pr + 8), sb0 = pack8(ds), sb1 = pack8(ds + 8);
#pragma unroll
            for (int db = 0; db < 4; db++) {
                acc_dv[db] = MFMA32(trf(dOs, offs, 0, db), pb0, acc_dv[db]);   // dV^T[d][key] += dO^T . P
                acc_dv[db] = MFMA32(trf(dOs, offs, 16, db), pb1, acc_dv[db]);
                acc_dk[db] = MFMA32(trf(Qus, offs, 0, db), sb0, acc_dk[db]);   // dK^T[d][key] += Qu^T . dS
                acc_dk[db] = MFMA32(trf(Qus, offs, 16, db), sb1, acc_dk[db]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int ib = ib_lo; ib <= ib_hi; ib += 2) {
        block(std::integral_constant<int, 0>{}, ib);
        if (ib + 1 <= ib_hi) block(std::integral_constant<int, 1>{}, ib + 1);
    }
    bf16_t* Ow = reinterpret_cast<bf16_t*>(Tw);
    store_acc_t(acc_dk, 1.f, Ow, p.dk + (int64_t)b * p.dq_bs + (int64_t)kw * p.dq_rs + h * FA_D, p.dq_rs, lane);
    store_acc_t(acc_dv, 1.f, Ow, p.dv + (int64_t)b * p.dq_bs + (int64_t)kw * p.dq_rs + h * FA_D, p.dq_rs, lane);
}


// ======================================================================================= host side
extern "C" int db1_relattn_flash_supported(int B, int L, int H, int D, int dt) {
    return (dt == DB1_BF16 && D == FA_D && B > 0 && H > 0 && L >= FA_BQ && (L % FA_BQ) == 0 && B <= 65535 && H <= 65535) ? 1 : 0;
}

int db1_flash16_fwd_launch(const FlashArgs& a, hipStream_t st);
int db1_flash16_bwd_q_launch(const FlashArgs& a, hipStream_t st);
int db1_flash16_bwd_kv_launch(const FlashArgs& a, hipStream_t st);
static bool flash_impl16() {  // the 8-wave kernels (relattn_flash16.hip) are the default; DB1_FLASH_IMPL=32 selects the 4-wave ones (A/B timing)
    static int v = -1;
    if (v < 0) { const char* e = getenv("DB1_FLASH_IMPL"); v = (e && atoi(e) == 32) ? 0 : 1; }
    return v == 1;
}
static int flash_check(const FlashArgs& a, int D, const char* what) {
    if (!db1_relattn_flash_supported(a.B, a.L, a.H, D, DB1_BF16)) DB1_FAIL(DB1_ERR_UNSUPPORTED, "%s: needs bf16, d_head = 128, L %% 128 == 0 (got L=%d D=%d)", what, a.L, D);
    if (a.shift < 1) DB1_FAIL(DB1_ERR_BAD_SHAPE, "%s: empty attention window (shift=%d)", what, a.shift);
    if ((a.kv_rs % 8) || (a.kv_bs % 8)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "%s: k/v strides must be multiples of 8 elements", what);
    if (!db1_aligned16(a.qu) || !db1_aligned16(a.qv) || !db1_aligned16(a.k) || !db1_aligned16(a.v) || !db1_aligned16(a.R))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "%s: operands must be 16-byte aligned", what);
    return DB1_OK;
}

extern "C" int db1_relattn_flash_fwd(const void* qu, const void* qv, const void* k, const void* v, int64_t kv_row_stride,
                                     int64_t kv_batch_stride, const void* R, void* out, float* lse, int B, int L, int H, int D,
                                     int shift, float scale, void* stream) {
    FlashArgs a = {};
    a.qu = (const bf16_t*)qu; a.qv = (const bf16_t*)qv; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.R = (const bf16_t*)R;
    a.o = (bf16_t*)out; a.lse_out = lse; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride;
    a.B = B; a.L = L; a.H = H; a.shift = shift; a.scale = scale;
    int st = flash_check(a, D, "relattn_flash_fwd");
    if (st) return st;
    if (!db1_aligned16(out)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_flash_fwd: out alignment");
    if (flash_impl16()) return db1_flash16_fwd_launch(a, (hipStream_t)stream);
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)relattn_flash_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS_BYTES); attr = true; }
    dim3 grid(flash_grid(L / FA_BQ, H, B));
    relattn_flash_fwd_kernel<<<grid, 256, FWD_LDS_BYTES, (hipStream_t)stream>>>(a);
    DB1_CHECK_LAUNCH("relattn_flash_fwd");
    return DB1_OK;
}

extern "C" int db1_relattn_flash_bwd(const void* qu, const void* qv, const void* k, const void* v, int64_t kv_row_stride,
                                     int64_t kv_batch_stride, const void* R, const void* out, const void* dout, const float* lse,
                                     float* delta, void* dq, void* dk, void* dv, int64_t dqkv_row_stride, int64_t dqkv_batch_stride,
                                     void* dT, int B, int L, int H, int D, int shift, float scale, void* stream) {
    FlashArgs a = {};
    a.qu = (const bf16_t*)qu; a.qv = (const bf16_t*)qv; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.R = (const bf16_t*)R;
    a.out = (const bf16_t*)out; a.dout = (const bf16_t*)dout; a.lse = lse; a.delta = delta;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.dT = (bf16_t*)dT;
    a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride; a.dq_rs = dqkv_row_stride; a.dq_bs = dqkv_batch_stride;
    a.B = B; a.L = L; a.H = H; a.shift = shift; a.scale = scale;
    int st = flash_check(a, D, "relattn_flash_bwd");
    if (st) return st;
    if ((a.dq_rs % 8) || (a.dq_bs % 8) || !db1_aligned16(dq) || !db1_aligned16(dk) || !db1_aligned16(dv) || !db1_aligned16(out) || !db1_aligned16(dout))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_flash_bwd: alignment");
    if (!delta || !dT || !lse) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_flash_bwd: null buffer");
    hipStream_t s = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)relattn_flash_bwd_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BQ_LDS_BYTES);
        hipFuncSetAttribute((const void*)relattn_flash_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KV_LDS_BYTES);
        attr = true;
    }
    const int64_t n_rows = (int64_t)B * L * H;
    relattn_delta_kernel<<<(unsigned)((n_rows + 15) / 16), 256, 0, s>>>(a.out, a.dout, delta, n_rows, L, H);
    DB1_CHECK_LAUNCH("relattn_delta");
    dim3 grid(flash_grid(L / FA_BQ, H, B));
    if (flash_impl16()) {
        st = db1_flash16_bwd_q_launch(a, s);
        if (st) return st;
    } else {
        relattn_flash_bwd_q_kernel<<<grid, 256, BQ_LDS_BYTES, s>>>(a);
        DB1_CHECK_LAUNCH("relattn_flash_bwd_q");
    }
    if (flash_impl16() && !getenv("DB1_FLASH_KV32")) return db1_flash16_bwd_kv_launch(a, s);
    relattn_flash_bwd_kv_kernel<<<grid, 256, KV_LDS_BYTES, s>>>(a);
    DB1_CHECK_LAUNCH("relattn_flash_bwd_kv");
    return DB1_OK;
}
