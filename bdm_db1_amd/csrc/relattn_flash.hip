// Relative-position flash attention for gfx950 (bf16, d_head = 128): Transformer-XL scores
//     s[i,j] = ((q_i+u).k_j + (q_i+v).R[i-j]) / sqrt(d),   visible iff  i - shift < j <= i
// (closed form of AC + _rel_shift(BD) + mask, transformer_xl.py:98-110,160-209,551-567) with online softmax and P.V fused, never
// materialising an (L x L) tensor in HBM in the forward.  Inputs qu = q+u and qv = q+v_bias are materialised once per layer by
// db1_relattn_add_head_bias.  Backward = two kernels without atomics (deterministic):
//   bwd_q : per 128 queries, loop keys    -> delta = rowsum(dO * O), dq_k = dS.K (the (q+u).k branch) and dT = dS re-indexed by distance (bf16, HBM);
//   bwd_kv: per 128 keys,    loop queries -> dV = P^T.dO, dK = dS^T.Qu.
// dq_r = dT.R (relattn_dqr.hip, a stream over dT) and dR = dT^T.Qv (a batched tile GEMM) are run by the caller on dT: exact causal FLOPs, no band overhead.
//
// All three kernels: one workgroup = 128 rows of one (batch, head) = 8 waves x 16 rows (two waves per SIMD) on
// v_mfma_f32_16x16x32_bf16, 32-column blocks, a 256-row LDS ring of R rows (the band of distances slides by 32 per block), operand
// tiles staged with global_load_lds.  (The first version ran 4 waves x 32 rows on the 32x32x16 MFMA, one wave per SIMD with up to 352
// VGPRs: every wave was bound by its own serial chain LDS read -> MFMA -> scratch -> softmax -> MFMA, matrix pipe 14-21 % busy.  With
// two lighter waves per SIMD one wave's LDS / VALU phases overlap the other's MFMAs: bwd_q 2104 -> ~1050 us, bwd_kv 1589 -> ~1200 us,
// fwd 968 -> 945 us at B = 64.)
//
// Fragment conventions (16x16x32): A[m][k]: lane (m = lane & 15, g = lane >> 4) holds k = 8g .. 8g+7; B[k][n]: lane (n, g) the same
// k; C[m][n]: lane (n = lane & 15, g) holds rows m = 4g + r, r = 0..3.
//   * "swapped" products put the query (fwd, bwd_q) or the key (bwd_kv) on the LANE axis of the accumulator: S^T = K.Qu^T per 16-row
//     tile t (two per block); MFMA row 4g + r of tile t is block row   kk(t, g) + r,   kk(t, g) = 16 ((g & 1) ^ t) + 8 (g >> 1) + 4 t
//     (a bijection onto 0..31).  The permutation makes the skewed scratch read conflict-free: lanes g and g ^ 1 of one 32-lane LDS
//     group differ by 16 rows = 16 banks.  A lane's 8 scores (tile 0: r = 0..3, tile 1: r = 0..3) are exactly k-slots 8g .. 8g+7 of
//     the B operand of O^T += V^T.P^T, and the V^T fragment takes the same rows through ds_read_b64_tr_b16 (rows kk(t, g) + 0..3):
//     P / dS feed the next MFMA straight from registers, softmax statistics need two v_permlane swaps.
//   * relative term (fwd, bwd_q): T = Qv.Rband^T for the 48 distances [iw - j0 - 32, iw - j0 + 16) of a 16 x 32 block (47 are needed;
//     starting one lower keeps the ring slots of a tile 16-aligned), C[q][dist] written to a per-wave scratch [16 q][64 (+4)] and read
//     back SKEWED at (q, q - key + 32).  Incremental: the band of block jb+1 is the band of block jb moved down by 32, so two new
//     16-distance tiles are computed and the third is kept; logical column c lives at c ^ (32 * parity), parity = unrolled instance.
//   * K / V tiles are prefetched TWO blocks ahead through three LDS stages (counted vmcnt): all query tiles of one (batch, head) walk the
//     keys in step, so a tile's first touch is an HBM miss (~2 k cycles under load) that a one-block prefetch distance did not hide
//     (a staging-only loop took 478 us of the 965 us forward at B = 64; 329 us with two blocks of distance).
//   * LDS images are row-major with the 16-byte chunk XOR-swizzled on the source side of the LDS-DMA:
//       K / V (Qu / dO) tiles:  f(row)  = (b4 << 3) | (b1 << 2) | (b0 << 1) | b2     (conflict-free for the permuted row fragments AND the tr reads)
//       ring rows, Qv tiles:    fr(row) = (((row & 7) ^ ((row & 8) >> 1)) << 1) | ((row >> 3) & 1)      (natural-order row fragments)
#include "relattn_flash.h"
#include <stdlib.h>

// ======================================================================================= forward
// SAVE: every processed (16 queries x 32 keys) wave-block also leaves its UNNORMALISED probabilities p~ = exp2((s - m) c2) as a fragment
// image (the bf16x8 the lane feeds to the P.V MFMA: one global_store_dwordx4 per lane, 1 KiB contiguous per wave) and the running
// maximum m c2 they were computed against; P = p~ exp2(m c2 - lse log2 e) is what the stored-probabilities backward rebuilds.
template <bool SAVE>
__global__ __launch_bounds__(512, 1) void relattn_flash_fwd_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rank, h, b;
    if (!flash_wg_coords(p.L / FA_BQ, p.H, p.B, rank, h, b)) return;
    const int qt = p.L / FA_BQ - 1 - rank;  // late query tiles have the longest key loops
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int i0 = qt * FA_BQ, iw = i0 + 16 * wave;
    const int a = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
    const bf16_t* qu = p.qu + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* qv = p.qv + ((int64_t)b * L) * HD + h * FA_D;
    W16Lane ln;
    w16_lane_init(ln, lds0, lds0 + W16_OFF_T + wave * W16_TW_BYTES, lane);

    bf16x8_t fqu[4], fqv[4];  // row iw + a, k = 32 ks + 8 g .. +7 (A and B operand images coincide)
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        fqu[ks] = *reinterpret_cast<const bf16x8_t*>(qu + (int64_t)(iw + a) * HD + ks * 32 + g * 8);
        fqv[ks] = *reinterpret_cast<const bf16x8_t*>(qv + (int64_t)(iw + a) * HD + ks * 32 + g * 8);
    }
    int jlo = i0 - p.shift + 1;
    if (jlo < 0) jlo = 0;
    const int jb_lo = jlo / FA_BK, jb_hi = (i0 + FA_BQ - 1) / FA_BK;
    W16Stage sg;
    w16_stage_init(sg, p.k + (int64_t)b * p.kv_bs + h * FA_D, p.v + (int64_t)b * p.kv_bs + h * FA_D, p.R + h * FA_D, p.kv_rs, HD, jb_lo * FA_BK, L,
                   lds0, wave, lane);
    // prologue: ring rows for distances [i0-j0-64, i0-j0+128) (blocks jb_lo and jb_lo+1) and the first two K/V tiles
    for (int c4 = -2; c4 < 4; c4++) w16_stage_ring(sg, i0 - jb_lo * FA_BK + 32 * c4, wave);
    w16_stage_kv(sg, 0, wave);
    if (jb_lo + 1 <= jb_hi) w16_stage_kv(sg, 1, wave);
    f32x4 acc_o[8];
#pragma unroll
    for (int db = 0; db < 8; db++) zero4(acc_o[db]);
    const float c2 = p.scale * LOG2E;
    float m_i = -1.0e30f, l_i = 0.f;  // m_i in RAW score units; l_i is this lane's PARTIAL row sum (reduced over g at the end)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), as a builtin so that hipcc's own bookkeeping sees the fragment loads retired
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int64_t sv_tile = (((int64_t)b * H + h) * flash_pt_tiles(L) + (iw / 16)) * 512 + lane * 8;   // + jb (NT - 1 - jb) * 512: the triangle of relattn_flash.h
    const int sv_nt = L / 16;
    float* mrow = SAVE ? p.mblk + ((int64_t)b * H + h) * (L / FA_BK) * L + iw + a : nullptr;                   // + jb * L
    bool have_prev = false;  // did this wave process the previous key block?  (wave-uniform)
    auto block = [&](auto STG, auto PARC, int jb) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value, par = decltype(PARC)::value;
        const int j0 = jb * FA_BK;
        const bool pf = jb + 2 <= jb_hi;
        if (pf) {  // prefetch block jb+2: its tiles and the 32 ring rows it adds land while this block and the next are computed
            w16_stage_kv(sg, (stg + 2) % W16_STAGES, wave);
            w16_stage_ring(sg, i0 - j0 - 96, wave);
        }
        if (j0 > iw + 15 || j0 + 31 <= iw - p.shift) have_prev = false;
        else {  // wave-uniform: blocks entirely outside this wave's window are skipped
            // relative term: logical tile t = distances dist_lo + 16 t .. -> scratch columns (16 t ..) ^ (32 par); tile 2 is the one
            // kept from the previous block (computed only for a wave's first processed block)
            const int dist_lo = iw - j0 - 32;
            w16_rel_tile(fqv, ln, dist_lo, 0 ^ (32 * par));
            w16_rel_tile(fqv, ln, dist_lo + 16, 16 ^ (32 * par));
            if (!have_prev) w16_rel_tile(fqv, ln, dist_lo + 32, 32 ^ (32 * par));
            have_prev = true;
            f32x4 acc_s[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                zero4(acc_s[t]);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) acc_s[t] = MFMA16(lds_ld128(ln.rowf[t][ks] + W16_OFF_K + stg * 8192), fqu[ks], acc_s[t]);  // S^T[key][query]
            }
            // skewed read of the relative term, then the V^T fragments of the P.V product (in flight during the softmax)
            float s[8];  // this lane: query iw + a; s[4 t + r]: key j0 + kk(t, g) + r
#pragma unroll
            for (int r = 0; r < 8; r++) s[r] = lds_ldf(ln.tsk[par][r]);
            bf16x8_t vt[8];
#pragma unroll
            for (int db = 0; db < 8; db++) vt[db] = lds_tr_pair(ln.tr[0][db] + W16_OFF_V + stg * 8192, ln.tr[1][db] + W16_OFF_V + stg * 8192);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) s[t * 4 + r] += acc_s[t][r];
            if (j0 + 31 > iw || j0 <= iw + 15 - p.shift) {  // only diagonal / window-edge blocks need the element mask
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int i = iw + a, j = j0 + kk16(t, g) + r;
                        s[t * 4 + r] = ((j <= i) && (j > i - p.shift)) ? s[t * 4 + r] : -1.0e30f;
                    }
            }
            float mblk = vmax3(vmax3(s[0], s[1], s[2]), vmax3(s[3], s[4], s[5]), vmax2(s[6], s[7]));
            mblk = max_x32(max_x16(mblk));
            const float m_new = vmax2(m_i, mblk);
            if (!__all(m_new == m_i)) {  // the running maxima rarely move after the first blocks: skip the O-wide rescale then
                const float alpha = __builtin_amdgcn_exp2f((m_i - m_new) * c2);
                l_i *= alpha;
#pragma unroll
                for (int db = 0; db < 8; db++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc_o[db][r] *= alpha;
                m_i = m_new;
            }
            const float mc = -m_i * c2;
#pragma unroll
            for (int r = 0; r < 8; r++) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, mc)); l_i += s[r]; }
            const bf16x8_t pb = pack8(s);
            if (SAVE) {
                *reinterpret_cast<bf16x8_t*>(p.pt + sv_tile + (int64_t)jb * (sv_nt - 1 - jb) * 512) = pb;
                if (g == 0) mrow[(int64_t)jb * L] = -mc;
            }
#pragma unroll
            for (int db = 0; db < 8; db++) acc_o[db] = MFMA16(vt[db], pb, acc_o[db]);  // O^T[d][query] += V^T . P^T
            if (SAVE) { if (pf) w16_vmcnt<3 + 2>(); else w16_vmcnt<2>(); }   // (the two stores sit behind this block's prefetch pieces)
        }
        // block jb+1 must have landed: all but the three pieces issued at the top of this block
        if (!SAVE || !have_prev) { if (pf) w16_vmcnt<3>(); else w16_vmcnt<0>(); }
        __syncthreads();  // every wave is done reading stage stg (refilled by the next block's prefetch)
    };
    W16_BLOCK_LOOP(block)
    l_i = sum_x32(sum_x16(l_i));
    bf16_t* Ow = reinterpret_cast<bf16_t*>(smem + W16_OFF_T + wave * W16_TW_BYTES);
    store_acc_t16(acc_o, 1.f / l_i, Ow, p.o + ((int64_t)b * L + iw) * HD + h * FA_D, HD, lane);
    if (g == 0) p.lse_out[((int64_t)b * H + h) * L + iw + a] = m_i * p.scale + logf(l_i);
}

// ======================================================================================= backward w.r.t. queries (+ dT)
// per wave-block (16 queries x 32 keys): S^T = K.Qu^T, dP^T = V.dO^T, the relative-term band (incremental), dS^T -> dq^T += K^T.dS^T, and
// dT[i][i - j] = dS[i][j]: in row i the 32 keys of a block are 32 CONSECUTIVE distances in reverse key order, so the lane's packed
// bf16 pairs go to a small [16 q][32] scratch (two ds_write_b64) and leave as 64 contiguous bytes per row.
// SAVE: the block's P and dS (bf16, masked entries zero) also leave for relattn_flash_bwd_kv2_kernel, as FRAGMENT IMAGES: the wave's
// 16 x 32 tile is the 1 KiB [lane][8] the lanes hold anyway (lane (a, g): query a, keys kk(0, g) + 0..3, kk(1, g) + 0..3), stored with
// one global_store_dwordx4 per lane = eight full 128-byte lines per wave and matrix.  (Plain [query][key] rows cost 64-byte partial
// lines and an LDS round trip: +470 us per layer at B = 64 against +180 us for this form.)
typedef unsigned __attribute__((aligned(2))) u32_a2_t;
template <bool SAVE>
__global__ __launch_bounds__(512, 1) void relattn_flash_bwd_q_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rank, h, b;
    if (!flash_wg_coords(p.L / FA_BQ, p.H, p.B, rank, h, b)) return;
    const int qt = p.L / FA_BQ - 1 - rank;
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int i0 = qt * FA_BQ, iw = i0 + 16 * wave;
    const int a = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
    const bf16_t* qu = p.qu + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* qv = p.qv + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* dog = p.dout + ((int64_t)b * L) * HD + h * FA_D;
    W16Lane ln;
    w16_lane_init(ln, lds0, lds0 + W16_OFF_T + wave * W16_TW_BYTES, lane);
    // dS scratch: lane (query a, g) writes keys kk(t, g) + 0..3 as element 31 - key: dwords 14 - kk/2, 15 - kk/2 of row a
    const unsigned dw0 = lds0 + W16_OFF_D + wave * 16 * W16_DP;
    unsigned dsw[2], dsr;
#pragma unroll
    for (int t = 0; t < 2; t++) { dsw[t] = dw0 + a * W16_DP + (14 - kk16(t, g) / 2) * 4; W16_OPAQUE(dsw[t]); }
    dsr = dw0 + (lane >> 4) * W16_DP + (lane & 15) * 4;  // + it * 4 rows
    W16_OPAQUE(dsr);
    // fragment image of tile (key block jb, query tile iw / 16): + jb (NT - 1 - jb) * 512 elements (the triangle of relattn_flash.h)
    const int64_t sv_tile = (((int64_t)b * p.H + h) * flash_pt_tiles(p.L) + (iw / 16)) * 512 + lane * 8;
    const int sv_nt = p.L / 16;

    bf16x8_t fqu[4], fqv[4], fdo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        fqu[ks] = *reinterpret_cast<const bf16x8_t*>(qu + (int64_t)(iw + a) * HD + ks * 32 + g * 8);
        fqv[ks] = *reinterpret_cast<const bf16x8_t*>(qv + (int64_t)(iw + a) * HD + ks * 32 + g * 8);
        fdo[ks] = *reinterpret_cast<const bf16x8_t*>(dog + (int64_t)(iw + a) * HD + ks * 32 + g * 8);
    }
    const float c2 = p.scale * LOG2E;
    const float nlse2 = -p.lse[((int64_t)b * H + h) * L + iw + a] * LOG2E;
    // delta_i = sum_d dO[i][d] * O[i][d] (the softmax backward's row term): this wave's 16 queries, 32 of the 128 d per lane, reduced
    // over the four lane groups; bwd_q visits every (b, h, i) exactly once, so it also publishes delta for bwd_kv (launched after it)
    float delta_a = 0.f;
    {
        const bf16_t* og = p.out + ((int64_t)b * L + iw + a) * HD + h * FA_D + g * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const bf16x8_t fo = *reinterpret_cast<const bf16x8_t*>(og + ks * 32);
#pragma unroll
            for (int e = 0; e < 8; e++)
                delta_a += __uint_as_float((unsigned)(unsigned short)fo[e] << 16) * __uint_as_float((unsigned)(unsigned short)fdo[ks][e] << 16);
        }
        delta_a = sum_x32(sum_x16(delta_a));
        if (g == 0) p.delta[((int64_t)b * H + h) * L + iw + a] = delta_a;
    }
    // dT row (iw + row) starts at distance (iw + row) - j0 - 31 for a block: row = 4 it + (lane >> 4), this lane's pair = elements 2 col, 2 col + 1
    bf16_t* dtp = p.dT + (((int64_t)h * p.B + b) * L + iw + (lane >> 4)) * L + (iw + (lane >> 4) - 31 + 2 * (lane & 15));
    const int drow0 = iw + (lane >> 4) - 31 + 2 * (lane & 15);  // distance of the pair's first element for j0 = 0, it = 0
    int jlo = i0 - p.shift + 1;
    if (jlo < 0) jlo = 0;
    const int jb_lo = jlo / FA_BK, jb_hi = (i0 + FA_BQ - 1) / FA_BK;
    W16Stage sg;
    w16_stage_init(sg, p.k + (int64_t)b * p.kv_bs + h * FA_D, p.v + (int64_t)b * p.kv_bs + h * FA_D, p.R + h * FA_D, p.kv_rs, HD, jb_lo * FA_BK, L,
                   lds0, wave, lane);
    for (int c4 = -2; c4 < 4; c4++) w16_stage_ring(sg, i0 - jb_lo * FA_BK + 32 * c4, wave);
    w16_stage_kv(sg, 0, wave);
    if (jb_lo + 1 <= jb_hi) w16_stage_kv(sg, 1, wave);
    f32x4 acc_dq[8];
#pragma unroll
    for (int db = 0; db < 8; db++) zero4(acc_dq[db]);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bool have_prev = false;
    int prev_sure = 0;   // lower bound of the stores the previous block issued behind its prefetch pieces
    constexpr int sure_tiles = SAVE ? 2 : 0;
    auto block = [&](auto STG, auto PARC, int jb) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value, par = decltype(PARC)::value;
        const int j0 = jb * FA_BK;
        const bool pf = jb + 2 <= jb_hi;
        if (pf) {
            w16_stage_kv(sg, (stg + 2) % W16_STAGES, wave);
            w16_stage_ring(sg, i0 - j0 - 96, wave);
        }
        int cur_sure = 0;
        if (j0 > iw + 15 || j0 + 31 <= iw - p.shift) have_prev = false;
        else {
            const int dist_lo = iw - j0 - 32;
            w16_rel_tile(fqv, ln, dist_lo, 0 ^ (32 * par));
            w16_rel_tile(fqv, ln, dist_lo + 16, 16 ^ (32 * par));
            if (!have_prev) w16_rel_tile(fqv, ln, dist_lo + 32, 32 ^ (32 * par));
            have_prev = true;
            f32x4 acc_s[2], acc_dp[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                zero4(acc_s[t]);
                zero4(acc_dp[t]);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) acc_s[t] = MFMA16(lds_ld128(ln.rowf[t][ks] + W16_OFF_K + stg * 8192), fqu[ks], acc_s[t]);   // S^T[key][query]
#pragma unroll
                for (int ks = 0; ks < 4; ks++) acc_dp[t] = MFMA16(lds_ld128(ln.rowf[t][ks] + W16_OFF_V + stg * 8192), fdo[ks], acc_dp[t]);  // dP^T[key][query]
            }
            float ds[8];
#pragma unroll
            for (int r = 0; r < 8; r++) ds[r] = lds_ldf(ln.tsk[par][r]);
            bf16x8_t kt[8];
#pragma unroll
            for (int db = 0; db < 8; db++) kt[db] = lds_tr_pair(ln.tr[0][db] + W16_OFF_K + stg * 8192, ln.tr[1][db] + W16_OFF_K + stg * 8192);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) ds[t * 4 + r] = __builtin_amdgcn_exp2f(fmaf(ds[t * 4 + r] + acc_s[t][r], c2, nlse2));
            const bool edge = j0 + 31 > iw || j0 <= iw + 15 - p.shift;
            if (edge) {
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int i = iw + a, j = j0 + kk16(t, g) + r;
                        ds[t * 4 + r] = ((j <= i) && (j > i - p.shift)) ? ds[t * 4 + r] : 0.f;
                    }
            }
            if (SAVE) *reinterpret_cast<bf16x8_t*>(p.pbuf + sv_tile + (int64_t)jb * (sv_nt - 1 - jb) * 512) = pack8(ds);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) ds[t * 4 + r] = ds[t * 4 + r] * (acc_dp[t][r] - delta_a) * p.scale;
            const bf16x8_t db8 = pack8(ds);
            if (SAVE) *reinterpret_cast<bf16x8_t*>(p.dsbuf + sv_tile + (int64_t)jb * (sv_nt - 1 - jb) * 512) = db8;
#pragma unroll
            for (int db = 0; db < 8; db++) acc_dq[db] = MFMA16(kt[db], db8, acc_dq[db]);  // dq^T[d][query] += K^T . dS^T
            // dT: element 31 - key of row a; the pairs are (key+1, key) in memory order
#pragma unroll
            for (int t = 0; t < 2; t++) {
                u32x2_t w;
                w[0] = pk_bf16(ds[t * 4 + 3], ds[t * 4 + 2]);
                w[1] = pk_bf16(ds[t * 4 + 1], ds[t * 4 + 0]);
                *(lds_u64_ptr)(size_t)dsw[t] = w;
            }
            bf16_t* drow = dtp - j0;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const unsigned v = *(lds_u32_ptr)(size_t)(dsr + it * 4 * W16_DP);
                bf16_t* dst = drow + (int64_t)(4 * it) * (L + 1);
                if (!edge) *reinterpret_cast<u32_a2_t*>(dst) = v;
                else {
                    const int d = drow0 - j0 + 4 * it;  // distance of the pair's first element (negative: above the diagonal, never stored)
                    if (d >= 0) *reinterpret_cast<u32_a2_t*>(dst) = v;
                    else if (d == -1) dst[1] = (bf16_t)(v >> 16);
                }
            }
            cur_sure = edge ? sure_tiles : 4 + sure_tiles;
        }
        // Block jb+1 must have landed; its pieces were issued at the top of block jb-1.  vmcnt counts loads and stores alike and
        // retires them in issue order, so everything issued after those pieces may stay outstanding: the stores of block jb-1, the
        // three pieces of this block and this block's stores (a plain vmcnt(3) would wait for the acknowledgement of stores issued
        // a few cycles ago, and for a write to reach the L2 takes longer than a block under load).  The counts are LOWER bounds of
        // what was issued: an interior block stores its four dT dwords (+ the two saved images) unconditionally, an edge block's dT
        // stores are predicated and are not counted, a skipped block stores nothing.
        w16_vmcnt_dyn(prev_sure + (pf ? 3 : 0) + cur_sure);
        prev_sure = cur_sure;
        __syncthreads();
    };
    W16_BLOCK_LOOP(block)
    bf16_t* Ow = reinterpret_cast<bf16_t*>(smem + W16_OFF_T + wave * W16_TW_BYTES);
    store_acc_t16(acc_dq, 1.f, Ow, p.dq + (int64_t)b * p.dq_bs + (int64_t)iw * p.dq_rs + h * FA_D, p.dq_rs, lane);
}

// ======================================================================================= backward w.r.t. queries from forward-stored p~
// relattn_flash_fwd_kernel<true> left p~ = exp2((s - m_blk) c2) per (16 queries x 32 keys) wave-block in exactly the lane layout this
// kernel computes in, and m_blk c2 per (key block, query).  So nothing is recomputed: P = p~ f with f = exp2(m_blk c2 - lse log2 e)
// (one exp2 per query and block instead of 32 + the two score contractions + the relative-term band and its skew), dP^T = V.dO^T,
// dS = P (dP - delta) scale, dq^T += K^T.dS^T, dT as in relattn_flash_bwd_q_kernel; dS leaves as a fragment image and f as a float per
// (key block, query) for relattn_flash_bwd_kv2_kernel<true>.  No ring, no relative-term scratch: 57 KiB of LDS.  p~ and m come through
// plain global loads two blocks ahead (16 B + 4 B per lane); they and the stores share vmcnt with the LDS-DMA pieces, see the wait.
// dT leaves through a per-wave ring [16 queries][64 distances] indexed by the ABSOLUTE distance (column = d & 63): the skew is taken by
// the 2-byte LDS writes, and after block jb the aligned window [iw - j0 - 16, iw - j0 + 16) is complete in every row, so it goes out
// as 16 rows x 64 B = one ds_read_b128 + one 16-byte-aligned global_store_dwordx4 per lane.  (The direct form -- 4-byte stores at
// 2-byte-aligned addresses, four per lane and block -- cost 330 us of this kernel's 1330 at B = 64.)  Distances above the diagonal
// (d < 0) are not stored; entries of the window nobody wrote are the ring's initial zeros, which is what dT holds there anyway.
#define DTR_PITCH 192                          // bytes per ring row: 64 bf16 + 64 (144: the window read 2-way, the 2-byte writes up to 4-way conflicted -- 20.6 % of the LDS cycles)
#define Q2_OFF_D (2 * W16_STAGES * 8192)
#define Q2_LDS (Q2_OFF_D + W16_WAVES * 16 * DTR_PITCH)
typedef __attribute__((address_space(3))) unsigned short* lds_u16_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) const u32x4_t* lds_u128_ptr;
// (the p~ stream in and the dT stream out are touched once: non-temporal, so that they do not push the K / V tiles -- shared by the eight
//  query tiles of a (batch, head) -- out of the XCD's L2; DB1_Q2_NT=0 at build time restores plain accesses)
#ifndef DB1_Q2_NT
#define DB1_Q2_NT 1   /* bit 0: loads (-0.7 %), bit 1: stores (+14 %: non-temporal stores are slow here) */
#endif
#if DB1_Q2_NT & 1
#define Q2_NT_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define Q2_NT_LOAD(ptr) (*(ptr))
#endif
#if DB1_Q2_NT & 2
#define Q2_NT_STORE(v, ptr) __builtin_nontemporal_store(v, ptr)
#else
#define Q2_NT_STORE(v, ptr) (*(ptr) = (v))
#endif
__global__ __launch_bounds__(512, 1) void relattn_flash_bwd_q2_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rank, h, b;
    if (!flash_wg_coords(p.L / FA_BQ, p.H, p.B, rank, h, b)) return;
    const int qt = p.L / FA_BQ - 1 - rank;
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int i0 = qt * FA_BQ, iw = i0 + 16 * wave;
    const int a = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
    const bf16_t* dog = p.dout + ((int64_t)b * L) * HD + h * FA_D;
    W16Lane ln;
    w16_lane_init(ln, lds0, lds0, lane);
    const unsigned dw0 = lds0 + Q2_OFF_D + wave * 16 * DTR_PITCH;
    const unsigned dwrow = dw0 + a * DTR_PITCH;            // ring row of this lane's query
    const unsigned drrow = dw0 + (lane >> 2) * DTR_PITCH;  // flush: row lane >> 2, 16-byte piece lane & 3
    for (int o = lane * 4; o < 16 * DTR_PITCH; o += 256) *(lds_u32_ptr)(size_t)(dw0 + o) = 0u;
    const int64_t sv_tile = (((int64_t)b * H + h) * flash_pt_tiles(L) + (iw / 16)) * 512 + lane * 8;   // + jb (NT - 1 - jb) * 512
    const int sv_nt = L / 16;
    auto sv_off = [&](int jb) __attribute__((always_inline)) { return (int64_t)jb * (sv_nt - 1 - jb) * 512; };
    const int64_t mrow = ((int64_t)b * H + h) * (L / FA_BK) * L + iw + a;                                     // + jb * L

    bf16x8_t fdo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) fdo[ks] = *reinterpret_cast<const bf16x8_t*>(dog + (int64_t)(iw + a) * HD + ks * 32 + g * 8);
    const float lse2 = p.lse[((int64_t)b * H + h) * L + iw + a] * LOG2E;
    float delta_a = 0.f;
    {
        const bf16_t* og = p.out + ((int64_t)b * L + iw + a) * HD + h * FA_D + g * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const bf16x8_t fo = *reinterpret_cast<const bf16x8_t*>(og + ks * 32);
#pragma unroll
            for (int e = 0; e < 8; e++)
                delta_a += __uint_as_float((unsigned)(unsigned short)fo[e] << 16) * __uint_as_float((unsigned)(unsigned short)fdo[ks][e] << 16);
        }
        delta_a = sum_x32(sum_x16(delta_a));
        if (g == 0) p.delta[((int64_t)b * H + h) * L + iw + a] = delta_a;
    }
    bf16_t* dtp = p.dT + (((int64_t)h * p.B + b) * L + iw + (lane >> 2)) * L + 8 * (lane & 3);   // + window start
    int jlo = i0 - p.shift + 1;
    if (jlo < 0) jlo = 0;
    const int jb_lo = jlo / FA_BK, jb_hi = (i0 + FA_BQ - 1) / FA_BK;
    W16Stage sg;
    w16_stage_init(sg, p.k + (int64_t)b * p.kv_bs + h * FA_D, p.v + (int64_t)b * p.kv_bs + h * FA_D, p.R + h * FA_D, p.kv_rs, HD, jb_lo * FA_BK, L,
                   lds0, wave, lane);
    w16_stage_kv(sg, 0, wave);
    if (jb_lo + 1 <= jb_hi) w16_stage_kv(sg, 1, wave);
    bf16x8_t ptq[2];   // p~ of the current / next block (slot = position in the loop & 1), m likewise
    float mq[2];
    ptq[0] = Q2_NT_LOAD(reinterpret_cast<const bf16x8_t*>(p.pt + sv_tile + sv_off(jb_lo)));
    mq[0] = p.mblk[mrow + (int64_t)jb_lo * L];
    if (jb_lo + 1 <= jb_hi) {
        ptq[1] = Q2_NT_LOAD(reinterpret_cast<const bf16x8_t*>(p.pt + sv_tile + sv_off(jb_lo + 1)));
        mq[1] = p.mblk[mrow + (int64_t)(jb_lo + 1) * L];
    }
    f32x4 acc_dq[8];
#pragma unroll
    for (int db = 0; db < 8; db++) zero4(acc_dq[db]);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int prev_sure = 0;
    auto block = [&](auto STG, auto PARC, int jb) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value, par = decltype(PARC)::value;
        const int j0 = jb * FA_BK;
        const bool pf = jb + 2 <= jb_hi;
        const bf16x8_t pt = ptq[par];
        const float m2 = mq[par];
        if (pf) {   // block jb+2: K / V tiles (two LDS-DMA pieces), then its p~ image and maxima into the slot just read
            w16_stage_kv(sg, (stg + 2) % W16_STAGES, wave);
            ptq[par] = Q2_NT_LOAD(reinterpret_cast<const bf16x8_t*>(p.pt + sv_tile + sv_off(jb + 2)));
            mq[par] = p.mblk[mrow + (int64_t)(jb + 2) * L];
        }
        int cur_sure = 0;
        if (!(j0 > iw + 15 || j0 + 31 <= iw - p.shift)) {
            f32x4 acc_dp[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                zero4(acc_dp[t]);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) acc_dp[t] = MFMA16(lds_ld128(ln.rowf[t][ks] + W16_OFF_V + stg * 8192), fdo[ks], acc_dp[t]);  // dP^T[key][query]
            }
            bf16x8_t kt[8];
#pragma unroll
            for (int db = 0; db < 8; db++) kt[db] = lds_tr_pair(ln.tr[0][db] + W16_OFF_K + stg * 8192, ln.tr[1][db] + W16_OFF_K + stg * 8192);
            const float f = __builtin_amdgcn_exp2f(m2 - lse2);
            if (g == 0) p.fblk[mrow + (int64_t)jb * L] = f;
            const float fs = f * p.scale;
            float ds[8];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++)   // masked pairs: the forward stored p~ = 0 there
                    ds[t * 4 + r] = __uint_as_float((unsigned)(unsigned short)pt[t * 4 + r] << 16) * fs * (acc_dp[t][r] - delta_a);
            const bf16x8_t db8 = pack8(ds);
#pragma unroll
            for (int db = 0; db < 8; db++) acc_dq[db] = MFMA16(kt[db], db8, acc_dq[db]);  // dq^T[d][query] += K^T . dS^T
            {   // ring column of key kk(t, g) + r of this lane's query: (iw + a - j0 - kk - r) & 63
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const unsigned col = (unsigned)(iw + a - j0 - kk16(t, g) - r) & 63u;
                        *(lds_u16_ptr)(size_t)(dwrow + col * 2) = (unsigned short)db8[t * 4 + r];
                    }
                const int w0 = iw - j0 - 16;   // window start (a multiple of 16)
                const u32x4_t win = *(lds_u128_ptr)(size_t)(drrow + (((unsigned)(w0 + 8 * (lane & 3)) & 63u) * 2));
                if (w0 + 8 * (lane & 3) >= 0) Q2_NT_STORE(win, reinterpret_cast<u32x4_t*>(dtp + w0));
            }
            cur_sure = 2;   // f, the dT window
        }
        // The K / V pieces of block jb+1 were issued at the top of block jb-1; vmcnt retires in issue order, so what may stay outstanding
        // is everything issued after them: the two loads that followed them, the stores of block jb-1, this block's four requests and
        // its stores (all counted as lower bounds).  The last block drains: the epilogue reuses the stages.
        if (jb + 1 <= jb_hi) w16_vmcnt_dyn(2 + prev_sure + (pf ? 4 : 0) + cur_sure); else w16_vmcnt<0>();
        prev_sure = cur_sure;
        __syncthreads();
    };
    W16_BLOCK_LOOP(block)
    bf16_t* Ow = reinterpret_cast<bf16_t*>(smem + wave * W16_TW_BYTES);
    store_acc_t16(acc_dq, 1.f, Ow, p.dq + (int64_t)b * p.dq_bs + (int64_t)iw * p.dq_rs + h * FA_D, p.dq_rs, lane);
}

// ======================================================================================= backward w.r.t. keys / values
// One workgroup = 128 keys of one (batch, head), wave = 16 keys (K / V fragments in registers), loop over 32-query blocks whose Qu, Qv,
// dO tiles and {lse, delta} come through two LDS stages.  Per wave-block (32 queries x 16 keys):
//   S[q][key] = Qu.K^T and dP = dO.V^T (8 + 8 MFMA; MFMA row 4g + r of tile t is query kk(t, g) + r, like the keys of the other kernels);
//   the relative term for the 47 distances of the block: T tiles (q-tile tq, 16-distance tile td) are only needed for td - tq in {0, 1}
//   (16 MFMA, Qv rows in natural order against ring rows), written to a scratch [32 q][32] (row q holds distances dist_lo + 16 (q >> 4) ..)
//   and read back at (q, q - key + 16); P and dS feed dV^T += dO^T.P and dK^T += Qu^T.dS as B operands straight from registers.
#define KV16_OFF_QU 0                          // two stages of 8 KiB each for Qu, Qv, dO
#define KV16_OFF_QV 16384
#define KV16_OFF_DO 32768
#define KV16_OFF_ST 49152                      // two stages of {lse[32], delta[32]} floats (raw, via LDS-DMA)
#define KV16_OFF_R 49664
#define KV16_OFF_T (KV16_OFF_R + FA_RING * 256)
#define KV16_LDS (KV16_OFF_T + W16_WAVES * W16_TW_BYTES)
#define KV16_TP 33                             // scratch row pitch (words): odd, so that queries 16 apart (lanes g, g ^ 1) land 16 banks apart
typedef __attribute__((address_space(3))) const f32x4* lds_f32x4_ptr;

__global__ __launch_bounds__(512, 1) void relattn_flash_bwd_kv_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int kt, h, b;  // early key tiles see the most queries: rank == tile index
    if (!flash_wg_coords(p.L / FA_BQ, p.H, p.B, kt, h, b)) return;
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int j0 = kt * FA_BQ, kw = j0 + 16 * wave;
    const int a = lane & 15, g = lane >> 4;   // a = key column of this lane
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
    const unsigned tw0 = lds0 + KV16_OFF_T + wave * W16_TW_BYTES;
    const bf16_t* qu = p.qu + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* qv = p.qv + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* dog = p.dout + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* kg = p.k + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* vg = p.v + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* Rg = p.R + h * FA_D;
    const float* lseg = p.lse + ((int64_t)b * H + h) * L;
    const float* delg = p.delta + ((int64_t)b * H + h) * L;
    W16Lane ln;
    w16_lane_init(ln, lds0, tw0, lane);  // (its scratch addresses are not used here: the block is 32 x 16, see tsk / twr below)
    unsigned tsk[8];   // scratch element (q, q - a + 16) of row q = kk(t, g) + r, stored at column q - a + 16 - 16 (q >> 4)
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = kk16(t, g) + r;
            tsk[t * 4 + r] = tw0 + (q * KV16_TP + q - a + 16 - 16 * (q >> 4)) * 4;
            W16_OPAQUE(tsk[t * 4 + r]);
        }
    unsigned twr = tw0 + ((4 * g) * KV16_TP + a) * 4;  // tile (tq, td): rows 16 tq + 4g + r, columns 16 (td - tq) + a
    W16_OPAQUE(twr);
    unsigned sta[2];   // {lse, delta} of queries kk(t, g) .. +3 (one 16-byte read each)
#pragma unroll
    for (int t = 0; t < 2; t++) { sta[t] = lds0 + KV16_OFF_ST + kk16(t, g) * 4; W16_OPAQUE(sta[t]); }

    bf16x8_t fk[4], fv[4];  // B-operand images: column = key kw + a
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        fk[ks] = *reinterpret_cast<const bf16x8_t*>(kg + (int64_t)(kw + a) * p.kv_rs + ks * 32 + g * 8);
        fv[ks] = *reinterpret_cast<const bf16x8_t*>(vg + (int64_t)(kw + a) * p.kv_rs + ks * 32 + g * 8);
    }
    const float c2 = p.scale * LOG2E;
    const int ib_lo = j0 / FA_BK;
    int ihi = j0 + FA_BQ - 1 + p.shift - 1;  // last query that can see the last key of the tile
    if (ihi > L - 1) ihi = L - 1;
    const int ib_hi = ihi / FA_BK;
    // staging: one piece (4 rows) of each of the three tiles and of the 32 new ring rows per wave and block; Qu / dO images use the
    // K / V swizzle (permuted row fragments + tr reads), the Qv image the ring swizzle (natural-order row fragments)
    const int srow = wave * 4 + (lane >> 4);
    const int64_t soff_kv = (int64_t)(ib_lo * FA_BK + srow) * HD + (((lane & 15) ^ swz_kv(srow)) << 3);
    const int64_t soff_rg = (int64_t)(ib_lo * FA_BK + srow) * HD + (((lane & 15) ^ swz_ring(srow & 15)) << 3);
    const bf16_t* quptr = qu + soff_kv;
    const bf16_t* doptr = dog + soff_kv;
    const bf16_t* qvptr = qv + soff_rg;
    const bf16_t* rbase = Rg + (((lane & 15) ^ swz_ring(srow & 15)) << 3);
    const int64_t q_step = (int64_t)FA_BK * HD;
    auto stage_q = [&](int stage, int i0n) __attribute__((always_inline)) {  // the next query block (rows i0n ..) -> stage
        glds16(quptr, lds0 + KV16_OFF_QU + stage * 8192 + wave * 1024);
        glds16(qvptr, lds0 + KV16_OFF_QV + stage * 8192 + wave * 1024);
        glds16(doptr, lds0 + KV16_OFF_DO + stage * 8192 + wave * 1024);
        quptr += q_step; qvptr += q_step; doptr += q_step;
        if (wave == 0) glds_stat(lseg, delg, i0n, reinterpret_cast<float*>(smem + KV16_OFF_ST + stage * 256), lane);
    };
    auto stage_ring = [&](int dist0) __attribute__((always_inline)) {
        const int slot0 = (dist0 + wave * 4) & (FA_RING - 1);
        const int dist = dist0 + srow;
        const int gr = dist < 0 ? 0 : (dist > L - 1 ? L - 1 : dist);
        glds16(rbase + (int64_t)gr * HD, lds0 + KV16_OFF_R + slot0 * 256);
    };
    // ring: distances [i0q - j0 - 128, i0q - j0 + 32) for the first block; every block prefetches the next 32
    for (int c4 = 0; c4 < 5; c4++) stage_ring(ib_lo * FA_BK - j0 - 128 + 32 * c4);
    stage_q(0, ib_lo * FA_BK);
    f32x4 acc_dk[8], acc_dv[8];
#pragma unroll
    for (int db = 0; db < 8; db++) { zero4(acc_dk[db]); zero4(acc_dv[db]); }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto block = [&](auto CUR, int ib) __attribute__((always_inline)) {
        constexpr int cur = decltype(CUR)::value;
        const int i0q = ib * FA_BK;
        if (ib < ib_hi) {  // prefetch the next query block
            stage_q(cur ^ 1, i0q + FA_BK);
            stage_ring(i0q + FA_BK - j0);
        }
        if (!(i0q + 31 < kw || i0q >= kw + 15 + p.shift)) {  // some (i, j) of this block pair is visible
            // ---- relative term: distances dist_lo + 16 td .., dist_lo = i0q - kw - 16; (tq, td) in {(0,0), (0,1), (1,1), (1,2)}
            const int dist_lo = i0q - kw - 16;
            bf16x8_t qvf[2][4];
#pragma unroll
            for (int tq = 0; tq < 2; tq++)
#pragma unroll
                for (int ks = 0; ks < 4; ks++) qvf[tq][ks] = lds_ld128(ln.ring[ks] + KV16_OFF_QV + cur * 8192 + tq * 4096);
#pragma unroll
            for (int td = 0; td < 3; td++) {
                const unsigned rs = (unsigned)((dist_lo + 16 * td) & (FA_RING - 1)) << 8;
                bf16x8_t rf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ks++) rf[ks] = lds_ld128(ln.ring[ks] + rs + KV16_OFF_R);
#pragma unroll
                for (int tq = 0; tq < 2; tq++) {
                    if (td - tq != 0 && td - tq != 1) continue;
                    f32x4 acc;
                    zero4(acc);
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) acc = MFMA16(qvf[tq][ks], rf[ks], acc);
#pragma unroll
                    for (int r = 0; r < 4; r++) lds_stf(twr + ((16 * tq + r) * KV16_TP + 16 * (td - tq)) * 4, acc[r]);
                }
            }
            f32x4 acc_s[2], acc_dp[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                zero4(acc_s[t]);
                zero4(acc_dp[t]);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) acc_s[t] = MFMA16(lds_ld128(ln.rowf[t][ks] + KV16_OFF_QU + cur * 8192), fk[ks], acc_s[t]);    // S[query][key]
#pragma unroll
                for (int ks = 0; ks < 4; ks++) acc_dp[t] = MFMA16(lds_ld128(ln.rowf[t][ks] + KV16_OFF_DO + cur * 8192), fv[ks], acc_dp[t]);  // dP[query][key]
            }
            float pr[8], ds[8];
#pragma unroll
            for (int r = 0; r < 8; r++) pr[r] = lds_ldf(tsk[r]);
            f32x4 lse4[2], del4[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                lse4[t] = *(lds_f32x4_ptr)(size_t)(sta[t] + cur * 256);
                del4[t] = *(lds_f32x4_ptr)(size_t)(sta[t] + cur * 256 + 128);
            }
            bf16x8_t dot[8], qut[8];
#pragma unroll
            for (int db = 0; db < 8; db++) {
                dot[db] = lds_tr_pair(ln.tr[0][db] + KV16_OFF_DO + cur * 8192, ln.tr[1][db] + KV16_OFF_DO + cur * 8192);
                qut[db] = lds_tr_pair(ln.tr[0][db] + KV16_OFF_QU + cur * 8192, ln.tr[1][db] + KV16_OFF_QU + cur * 8192);
            }
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++)  // this lane: key kw + a; register 4t + r: query i0q + kk(t, g) + r
                    pr[t * 4 + r] = __builtin_amdgcn_exp2f(fmaf(pr[t * 4 + r] + acc_s[t][r], c2, -LOG2E * lse4[t][r]));
            if (i0q < kw + 15 || i0q + 31 >= kw + p.shift) {  // diagonal / window-edge block pairs need the element mask
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int i = i0q + kk16(t, g) + r, j = kw + a;
                        pr[t * 4 + r] = ((j <= i) && (j > i - p.shift)) ? pr[t * 4 + r] : 0.f;
                    }
            }
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) ds[t * 4 + r] = pr[t * 4 + r] * (acc_dp[t][r] - del4[t][r]) * p.scale;
            const bf16x8_t pb = pack8(pr), sb = pack8(ds);
#pragma unroll
            for (int db = 0; db < 8; db++) {
                acc_dv[db] = MFMA16(dot[db], pb, acc_dv[db]);   // dV^T[d][key] += dO^T . P
                acc_dk[db] = MFMA16(qut[db], sb, acc_dk[db]);   // dK^T[d][key] += Qu^T . dS
            }
        }
        w16_vmcnt<0>();
        __syncthreads();
    };
    for (int ib = ib_lo; ib <= ib_hi; ib += 2) {
        block(std::integral_constant<int, 0>{}, ib);
        if (ib + 1 <= ib_hi) block(std::integral_constant<int, 1>{}, ib + 1);
    }
    bf16_t* Ow = reinterpret_cast<bf16_t*>(smem + KV16_OFF_T + wave * W16_TW_BYTES);
    store_acc_t16(acc_dk, 1.f, Ow, p.dk + (int64_t)b * p.dq_bs + (int64_t)kw * p.dq_rs + h * FA_D, p.dq_rs, lane);
    store_acc_t16(acc_dv, 1.f, Ow, p.dv + (int64_t)b * p.dq_bs + (int64_t)kw * p.dq_rs + h * FA_D, p.dq_rs, lane);
}


// ======================================================================================= backward w.r.t. keys / values from stored P / dS
// The key side without recomputation: relattn_flash_bwd_q_kernel<true> leaves P and dS as fragment images (bf16, masked entries zero,
// tiles without a visible pair untouched) and this kernel is the pair of causal contractions dV^T += dO^T.P, dK^T += Qu^T.dS over them:
// 16 MFMAs per (32 queries x 16 keys) wave-block instead of 48, no relative term, no ring, no softmax.  One workgroup = 128 keys of one
// (batch, head), wave = 16 keys x all 128 d.  Per 32-query block the 2 x 4 (query tile, key block) images of P and of dS (1 KiB each,
// copied verbatim by one LDS-DMA per wave) and the [32][128 d] tiles of Qu and dO come through four LDS stages, requested three
// blocks ahead.  It is a stream over HBM: 2 x L x L x 2 B per (batch, head) over the causal half for 4 L^2 d FLOPs.  Every operand
// fragment is a ds_read_b64_tr_b16 pair: A = dO^T / Qu^T (d on the lane axis, as in the other kernels), B = the wave's 16 keys on the
// lane axis, k-slots = queries kk(t, g) + r for both.  In an image the 8-byte unit (query q, keys 4 n .. 4 n + 3) of a key block sits
// in chunk 16 g_f + (q & 15), half t_f, with kk(t_f, g_f) = 4 n; the copy permutes the 16-byte chunks (c -> c ^ 4 (bit 5 of c) ^
// 8 (query tile & 1)) so that the 32 lanes of a tr read hit 32 distinct bank pairs.
// Measured at B = 64 (8192 workgroups): 770 us for 3.7 GB on the HBM side (PMC): 4.8 TB/s, bandwidth-bound.  A seven-stage image ring
// with per-wave request roles (96 KiB in flight per CU), register staging through global_load_dwordx4 + ds_write_b128, and a rotated
// visiting order all measured 770-820 us.
#define KV2_STAGES 4
#define KV2_OFF_P 0
#define KV2_OFF_DS (KV2_STAGES * 8192)
#define KV2_OFF_QU (2 * KV2_STAGES * 8192)
#define KV2_OFF_DO (3 * KV2_STAGES * 8192)
#define KV2_LDS (4 * KV2_STAGES * 8192)        // the output staging of the epilogue reuses the stages

__device__ __forceinline__ int kv2_chunk_pos(int c, int qtl) { return c ^ (((c >> 5) & 1) << 2) ^ (qtl << 3); }

// FACT: the P images are the forward's p~; P = p~ f with f [key block][query] from relattn_flash_bwd_q2_kernel, and dS is formed HERE as
// P (dP - delta) scale from dP = dO.V^T (8 more MFMAs; V of the lane's key stationary in registers) instead of being read: 64 floats per
// wave and block -- f and delta of the block's 32 queries, one 4-byte LDS-DMA per lane -- into a per-wave slot of the stage.
#define KV2_OFF_F (4 * KV2_STAGES * 8192)       // [stage][wave][64] floats
#define KV2_LDS_F (KV2_OFF_F + KV2_STAGES * W16_WAVES * 256)
template <bool FACT>
__global__ __launch_bounds__(512, 1) void relattn_flash_bwd_kv2_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int kt, h, b;
    if (!flash_wg_coords(p.L / FA_BQ, p.H, p.B, kt, h, b)) return;
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int j0 = kt * FA_BQ, kw = j0 + 16 * wave;
    const int a = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
    W16Lane ln;
    w16_lane_init(ln, lds0, lds0, lane);   // the tr addresses of a [32][128] tile at LDS offset 0 (scratch addresses unused)
    unsigned btr[2];   // B fragments out of the images of key block (wave >> 1): k-slot group t <-> queries kk(t, g) + (a >> 2), keys 16 (wave & 1) + 4 (a & 3) ..
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int q = kk16(t, g) + (a >> 2), qtl = q >> 4, qa = q & 15;
        const int tf = a & 1, gh = (a >> 1) & 1, gl = (wave & 1) ^ tf;
        const int c = (2 * gh + gl) * 16 + qa;
        btr[t] = lds0 + ((wave >> 1) * 2 + qtl) * 1024 + kv2_chunk_pos(c, qtl) * 16 + tf * 8;
        W16_OPAQUE(btr[t]);
    }
    const int ib_lo = j0 / FA_BK;
    int ihi = j0 + FA_BQ - 1 + p.shift - 1;
    if (ihi > L - 1) ihi = L - 1;
    const int ib_hi = ihi / FA_BK;
    const int srow = wave * 4 + (lane >> 4);
    const int schunk = ((lane & 15) ^ swz_kv(srow)) << 3;
    const bf16_t* quptr = p.qu + ((int64_t)b * L + ib_lo * FA_BK + srow) * HD + h * FA_D + schunk;
    const bf16_t* doptr = p.dout + ((int64_t)b * L + ib_lo * FA_BK + srow) * HD + h * FA_D + schunk;
    // this wave copies image (key block j0 / 32 + (wave >> 1), query tile 2 ib + (wave & 1)); LDS chunk `lane` takes source chunk pos^-1 = pos
    const int64_t img = (((int64_t)b * H + h) * flash_pt_tiles(L) + flash_pt_index(j0 / FA_BK + (wave >> 1), 2 * ib_lo + (wave & 1), L / 16)) * 512 +
                        kv2_chunk_pos(lane, wave & 1) * 8;
    const bf16_t* pptr = (FACT ? p.pt : p.pbuf) + img;
    const bf16_t* sptr = p.dsbuf + img;
    const int64_t q_step = (int64_t)FA_BK * HD;
    // FACT: lanes 0-31 fetch f [key block][query], lanes 32-63 delta [query] of the block's 32 queries (one 4-byte LDS-DMA per lane)
    const float* fptr = !FACT ? nullptr
                        : (lane < 32 ? p.fblk + (((int64_t)b * H + h) * (L / FA_BK) + j0 / FA_BK + (wave >> 1)) * L + ib_lo * FA_BK + lane
                                     : p.delta + ((int64_t)b * H + h) * L + ib_lo * FA_BK + lane - 32);
    unsigned fld[2];   // f of queries kk(t, g) .. + 3 (delta: + 128 bytes)
#pragma unroll
    for (int t = 0; t < 2; t++) { fld[t] = lds0 + KV2_OFF_F + wave * 256 + kk16(t, g) * 4; W16_OPAQUE(fld[t]); }
    bf16x8_t fv[4];    // FACT: V of this lane's key (B operand of dP = dO.V^T)
    if (FACT) {
        const bf16_t* vg = p.v + (int64_t)b * p.kv_bs + h * FA_D;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) fv[ks] = *reinterpret_cast<const bf16x8_t*>(vg + (int64_t)(kw + a) * p.kv_rs + ks * 32 + g * 8);
    }
    constexpr int NP = 4;   // requests per wave and block
    auto stage = [&](int stg) __attribute__((always_inline)) {
        if (FACT) {
            const unsigned dstf = __builtin_amdgcn_readfirstlane(lds0 + KV2_OFF_F + stg * (W16_WAVES * 256) + wave * 256);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(fptr), "s"(dstf) : "memory");
            fptr += FA_BK;
        }
        glds16(pptr, lds0 + KV2_OFF_P + stg * 8192 + wave * 1024);
        if (!FACT) glds16(sptr, lds0 + KV2_OFF_DS + stg * 8192 + wave * 1024);
        glds16(quptr, lds0 + KV2_OFF_QU + stg * 8192 + wave * 1024);
        glds16(doptr, lds0 + KV2_OFF_DO + stg * 8192 + wave * 1024);
        pptr += 1024; sptr += 1024; quptr += q_step; doptr += q_step;   // the next 32 queries: two images further
    };
    stage(0);
    if (ib_lo + 1 <= ib_hi) stage(1);
    if (ib_lo + 2 <= ib_hi) stage(2);
    f32x4 acc_dk[8], acc_dv[8];
#pragma unroll
    for (int db = 0; db < 8; db++) { zero4(acc_dk[db]); zero4(acc_dv[db]); }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (the V fragments; as a builtin so that hipcc's own bookkeeping sees them retired)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto block = [&](auto STG, int ib) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value;
        const int i0q = ib * FA_BK;
        if (ib + 3 <= ib_hi) stage((stg + 3) % KV2_STAGES);
        if (!(i0q + 31 < kw || i0q >= kw + 15 + p.shift)) {  // some (i, j) of this block pair is visible
            bf16x8_t pb = lds_tr_pair(btr[0] + KV2_OFF_P + stg * 8192, btr[1] + KV2_OFF_P + stg * 8192);
            bf16x8_t sb;
            if (!FACT) sb = lds_tr_pair(btr[0] + KV2_OFF_DS + stg * 8192, btr[1] + KV2_OFF_DS + stg * 8192);
            else {
                // P = p~ f;  dP[query][key] = dO.V^T (MFMA row 4g + r of tile t is query kk(t, g) + r, like the k-slots);  dS = P (dP - delta) scale
                float pf32[8], ds32[8];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    f32x4 acc_dp;
                    zero4(acc_dp);
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) acc_dp = MFMA16(lds_ld128(ln.rowf[t][ks] + KV2_OFF_DO + stg * 8192), fv[ks], acc_dp);
                    const f32x4 f4 = *(lds_f32x4_ptr)(size_t)(fld[t] + stg * (W16_WAVES * 256));
                    const f32x4 d4 = *(lds_f32x4_ptr)(size_t)(fld[t] + stg * (W16_WAVES * 256) + 128);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        pf32[t * 4 + r] = __uint_as_float((unsigned)(unsigned short)pb[t * 4 + r] << 16) * f4[r];
                        ds32[t * 4 + r] = pf32[t * 4 + r] * (acc_dp[r] - d4[r]) * p.scale;
                    }
                }
                pb = pack8(pf32);
                sb = pack8(ds32);
            }
            if (i0q < kw + 15 || i0q + 31 >= kw + p.shift) {
                // diagonal / window-edge pairs: masked entries of a written tile are zero already, but a 16 x 32 tile without any
                // visible pair was never written by bwd_q -- select, do not multiply (the bytes there are arbitrary)
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int i = i0q + kk16(t, g) + r, j = kw + a;
                        const bool keep = (j <= i) && (j > i - p.shift);
                        pb[t * 4 + r] = keep ? pb[t * 4 + r] : (short)0;
                        sb[t * 4 + r] = keep ? sb[t * 4 + r] : (short)0;
                    }
            }
#pragma unroll
            for (int db = 0; db < 8; db++) {
                const bf16x8_t dot = lds_tr_pair(ln.tr[0][db] + KV2_OFF_DO + stg * 8192, ln.tr[1][db] + KV2_OFF_DO + stg * 8192);
                const bf16x8_t qut = lds_tr_pair(ln.tr[0][db] + KV2_OFF_QU + stg * 8192, ln.tr[1][db] + KV2_OFF_QU + stg * 8192);
                acc_dv[db] = MFMA16(dot, pb, acc_dv[db]);   // dV^T[d][key] += dO^T . P
                acc_dk[db] = MFMA16(qut, sb, acc_dk[db]);   // dK^T[d][key] += Qu^T . dS
            }
        }
        // block ib+1 must have landed: everything but the pieces of the (up to two) later blocks already requested
        if (ib + 3 <= ib_hi) w16_vmcnt<2 * NP>(); else if (ib + 2 <= ib_hi) w16_vmcnt<NP>(); else w16_vmcnt<0>();
        __syncthreads();
    };
    for (int ib = ib_lo; ib <= ib_hi; ib += 4) {
        block(std::integral_constant<int, 0>{}, ib);
        if (ib + 1 <= ib_hi) block(std::integral_constant<int, 1>{}, ib + 1);
        if (ib + 2 <= ib_hi) block(std::integral_constant<int, 2>{}, ib + 2);
        if (ib + 3 <= ib_hi) block(std::integral_constant<int, 3>{}, ib + 3);
    }
    bf16_t* Ow = reinterpret_cast<bf16_t*>(smem + wave * W16_TW_BYTES);   // (every wave is past the final barrier: the stages are free)
    store_acc_t16(acc_dk, 1.f, Ow, p.dk + (int64_t)b * p.dq_bs + (int64_t)kw * p.dq_rs + h * FA_D, p.dq_rs, lane);
    store_acc_t16(acc_dv, 1.f, Ow, p.dv + (int64_t)b * p.dq_bs + (int64_t)kw * p.dq_rs + h * FA_D, p.dq_rs, lane);
}

// ======================================================================================= key side, 32 keys per wave (forward-stored p~)
// relattn_flash_bwd_kv2_kernel<true> with a wave owning a whole 32-key block (two 16-key tiles) and a workgroup 256 keys: the A fragments
// of a 32-query block -- dO^T and Qu^T through 32 transposing reads, the dO rows of dP through 8 ds_read_b128 -- are per wave, so they
// now feed 48 MFMAs instead of 24 (the kernel is issue-bound: LDS reads were 44 of its ~130 instructions per wave-block).  A wave copies
// the two images (query tiles 2 ib, 2 ib + 1) of ITS key block and reads both key halves' B fragments from them.  Needs L % 256 == 0.
// 1734 -> 1652 us for the backward pair at B = 64.  (The same move on the query side -- 32 queries per wave, 256 per workgroup, V / K^T
// fragments feeding 32 MFMAs -- was built, is correct, and is 4 % SLOWER, 1658 -> 1718 us: the causal triangle leaves the low-query waves of
// a 256-query workgroup idle at the barriers of the late key blocks (82 % useful wave-blocks against 92 %), and twice the (batch, head)
// pairs are in flight per XCD.  Removed again.)
#define KV3_KEYS 256
#define KV3_STAGES 4
#define KV3_OFF_P 0                                   // [stage][8 key blocks][2 query tiles] images of 1 KiB
#define KV3_OFF_QU (KV3_STAGES * 16384)
#define KV3_OFF_DO (KV3_OFF_QU + KV3_STAGES * 8192)
#define KV3_OFF_F (KV3_OFF_DO + KV3_STAGES * 8192)    // [stage][wave][64] floats: f [32 queries], delta [32 queries]
#define KV3_LDS (KV3_OFF_F + KV3_STAGES * W16_WAVES * 256)
__global__ __launch_bounds__(512, 1) void relattn_flash_bwd_kv3_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int kt, h, b;
    if (!flash_wg_coords(p.L / KV3_KEYS, p.H, p.B, kt, h, b)) return;
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int j0 = kt * KV3_KEYS, kw = j0 + 32 * wave;
    const int a = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
    W16Lane ln;
    w16_lane_init(ln, lds0, lds0, lane);   // the tr addresses of a [32][128] tile at LDS offset 0 (scratch addresses unused)
    unsigned btr[2][2];   // [key half][k-slot group t]: queries kk(t, g) + (a >> 2), keys 16 half + 4 (a & 3) .. of this wave's key block
#pragma unroll
    for (int hf = 0; hf < 2; hf++)
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int q = kk16(t, g) + (a >> 2), qtl = q >> 4, qa = q & 15;
            const int tf = a & 1, gh = (a >> 1) & 1, gl = hf ^ tf;
            const int c = (2 * gh + gl) * 16 + qa;
            btr[hf][t] = lds0 + (wave * 2 + qtl) * 1024 + kv2_chunk_pos(c, qtl) * 16 + tf * 8;
            W16_OPAQUE(btr[hf][t]);
        }
    const int ib_lo = j0 / FA_BK;
    int ihi = j0 + KV3_KEYS - 1 + p.shift - 1;
    if (ihi > L - 1) ihi = L - 1;
    const int ib_hi = ihi / FA_BK;
    const int srow = wave * 4 + (lane >> 4);
    const int schunk = ((lane & 15) ^ swz_kv(srow)) << 3;
    // The query blocks are walked DOWNWARDS from the last one: the four key tiles of a (batch, head) -- co-resident on one XCD -- then start on
    // the same block at the same time and stay in step, so its Qu / dO tiles are fetched from HBM once instead of once per key tile (walking
    // up, tile kt starts at block 8 kt: no two tiles ever touch a block at the same time -- L2 hit rate 21 %, 3.25 GB per launch at 4.7 TB/s).
    const bf16_t* quptr = p.qu + ((int64_t)b * L + ib_hi * FA_BK + srow) * HD + h * FA_D + schunk;
    const bf16_t* doptr = p.dout + ((int64_t)b * L + ib_hi * FA_BK + srow) * HD + h * FA_D + schunk;
    // this wave copies the images (key block j0 / 32 + wave, query tiles 2 ib and 2 ib + 1): adjacent in memory
    const int64_t img = (((int64_t)b * H + h) * flash_pt_tiles(L) + flash_pt_index(j0 / FA_BK + wave, 2 * ib_hi, L / 16)) * 512;
    const bf16_t* pptr0 = p.pt + img + kv2_chunk_pos(lane, 0) * 8;
    const bf16_t* pptr1 = p.pt + img + 512 + kv2_chunk_pos(lane, 1) * 8;
    const int64_t q_step = (int64_t)FA_BK * HD;
    const float* fptr = lane < 32 ? p.fblk + (((int64_t)b * H + h) * (L / FA_BK) + j0 / FA_BK + wave) * L + ib_hi * FA_BK + lane
                                  : p.delta + ((int64_t)b * H + h) * L + ib_hi * FA_BK + lane - 32;
    unsigned fld[2];   // f of queries kk(t, g) .. + 3 (delta: + 128 bytes)
#pragma unroll
    for (int t = 0; t < 2; t++) { fld[t] = lds0 + KV3_OFF_F + wave * 256 + kk16(t, g) * 4; W16_OPAQUE(fld[t]); }
    bf16x8_t fv[2][4];    // V of this lane's two keys (B operand of dP = dO.V^T)
    {
        const bf16_t* vg = p.v + (int64_t)b * p.kv_bs + h * FA_D;
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) fv[hf][ks] = *reinterpret_cast<const bf16x8_t*>(vg + (int64_t)(kw + 16 * hf + a) * p.kv_rs + ks * 32 + g * 8);
    }
    constexpr int NP = 5;   // requests per wave and block
    auto stage = [&](int stg) __attribute__((always_inline)) {
        const unsigned dstf = __builtin_amdgcn_readfirstlane(lds0 + KV3_OFF_F + stg * (W16_WAVES * 256) + wave * 256);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(fptr), "s"(dstf) : "memory");
        fptr -= FA_BK;
        glds16(pptr0, lds0 + KV3_OFF_P + stg * 16384 + wave * 2048);
        glds16(pptr1, lds0 + KV3_OFF_P + stg * 16384 + wave * 2048 + 1024);
        glds16(quptr, lds0 + KV3_OFF_QU + stg * 8192 + wave * 1024);
        glds16(doptr, lds0 + KV3_OFF_DO + stg * 8192 + wave * 1024);
        pptr0 -= 1024; pptr1 -= 1024; quptr -= q_step; doptr -= q_step;   // the previous 32 queries: two images back
    };
    stage(0);
    if (ib_hi - 1 >= ib_lo) stage(1);
    if (ib_hi - 2 >= ib_lo) stage(2);
    f32x4 acc_dk[2][8], acc_dv[2][8];
#pragma unroll
    for (int hf = 0; hf < 2; hf++)
#pragma unroll
        for (int db = 0; db < 8; db++) { zero4(acc_dk[hf][db]); zero4(acc_dv[hf][db]); }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (the V fragments; as a builtin so that hipcc's own bookkeeping sees them retired)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto block = [&](auto STG, int ib) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value;
        const int i0q = ib * FA_BK;
        if (ib - 3 >= ib_lo) stage((stg + 3) % KV3_STAGES);
        if (!(i0q + 31 < kw || i0q >= kw + 31 + p.shift)) {  // some (i, j) of this block pair is visible
            bf16x8_t pb[2], sb[2];
            // dP[query][key] = dO.V^T: the dO rows of a 16-query tile (A operand) are read once for both key tiles
            f32x4 adp[2][2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                bf16x8_t dor[4];
#pragma unroll
                for (int ks = 0; ks < 4; ks++) dor[ks] = lds_ld128(ln.rowf[t][ks] + KV3_OFF_DO + stg * 8192);
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    zero4(adp[hf][t]);
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) adp[hf][t] = MFMA16(dor[ks], fv[hf][ks], adp[hf][t]);
                }
            }
            f32x4 f4[2], d4[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                f4[t] = *(lds_f32x4_ptr)(size_t)(fld[t] + stg * (W16_WAVES * 256));
                d4[t] = *(lds_f32x4_ptr)(size_t)(fld[t] + stg * (W16_WAVES * 256) + 128);
            }
            const bool edge = i0q < kw + 31 || i0q + 31 >= kw + p.shift;
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                const bf16x8_t pt = lds_tr_pair(btr[hf][0] + KV3_OFF_P + stg * 16384, btr[hf][1] + KV3_OFF_P + stg * 16384);
                float pf32[8], ds32[8];
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        pf32[t * 4 + r] = __uint_as_float((unsigned)(unsigned short)pt[t * 4 + r] << 16) * f4[t][r];
                        ds32[t * 4 + r] = pf32[t * 4 + r] * (adp[hf][t][r] - d4[t][r]) * p.scale;
                    }
                pb[hf] = pack8(pf32);
                sb[hf] = pack8(ds32);
                if (edge) {
                    // diagonal / window-edge pairs: a 16 x 32 tile without any visible pair was never written by the forward -- select, do
                    // not multiply (the bytes there are arbitrary)
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int i = i0q + kk16(t, g) + r, j = kw + 16 * hf + a;
                            const bool keep = (j <= i) && (j > i - p.shift);
                            pb[hf][t * 4 + r] = keep ? pb[hf][t * 4 + r] : (short)0;
                            sb[hf][t * 4 + r] = keep ? sb[hf][t * 4 + r] : (short)0;
                        }
                }
            }
#pragma unroll
            for (int db = 0; db < 8; db++) {
                const bf16x8_t dot = lds_tr_pair(ln.tr[0][db] + KV3_OFF_DO + stg * 8192, ln.tr[1][db] + KV3_OFF_DO + stg * 8192);
                const bf16x8_t qut = lds_tr_pair(ln.tr[0][db] + KV3_OFF_QU + stg * 8192, ln.tr[1][db] + KV3_OFF_QU + stg * 8192);
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    acc_dv[hf][db] = MFMA16(dot, pb[hf], acc_dv[hf][db]);   // dV^T[d][key] += dO^T . P
                    acc_dk[hf][db] = MFMA16(qut, sb[hf], acc_dk[hf][db]);   // dK^T[d][key] += Qu^T . dS
                }
            }
        }
        // block ib+1 must have landed: everything but the pieces of the (up to two) later blocks already requested
        if (ib - 3 >= ib_lo) w16_vmcnt<2 * NP>(); else if (ib - 2 >= ib_lo) w16_vmcnt<NP>(); else w16_vmcnt<0>();
        __syncthreads();
    };
    for (int ib = ib_hi; ib >= ib_lo; ib -= 4) {
        block(std::integral_constant<int, 0>{}, ib);
        if (ib - 1 >= ib_lo) block(std::integral_constant<int, 1>{}, ib - 1);
        if (ib - 2 >= ib_lo) block(std::integral_constant<int, 2>{}, ib - 2);
        if (ib - 3 >= ib_lo) block(std::integral_constant<int, 3>{}, ib - 3);
    }
    bf16_t* Ow = reinterpret_cast<bf16_t*>(smem + wave * W16_TW_BYTES);   // (every wave is past the final barrier: the stages are free)
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        store_acc_t16(acc_dk[hf], 1.f, Ow, p.dk + (int64_t)b * p.dq_bs + (int64_t)(kw + 16 * hf) * p.dq_rs + h * FA_D, p.dq_rs, lane);
        store_acc_t16(acc_dv[hf], 1.f, Ow, p.dv + (int64_t)b * p.dq_bs + (int64_t)(kw + 16 * hf) * p.dq_rs + h * FA_D, p.dq_rs, lane);
    }
}

// ======================================================================================= host side
extern "C" int db1_relattn_flash_fwd2_launch(const FlashArgs* a, void* stream);   // relattn_flash_fwd2.hip (8 waves x 16 rows)
extern "C" int db1_relattn_flash_fwd3_launch(const FlashArgs* a, void* stream);   // relattn_flash_fwd3.hip (4 waves x 32 rows: the default)
static thread_local int g_fwd2_on = 1;   // test hook (include/db1_hip_test.h): 0 compiled loop, 1 default (fwd3), 2 fwd2
extern "C" void db1_test_flash_fwd2(int on) { g_fwd2_on = on; }
extern "C" int db1_relattn_flash_supported(int B, int L, int H, int D, int dt) {
    return (dt == DB1_BF16 && D == FA_D && B > 0 && H > 0 && L >= FA_BQ && (L % FA_BQ) == 0 && B <= 65535 && H <= 65535) ? 1 : 0;
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
                                     int shift, float scale, void* probs, float* mblk, void* stream) {
    FlashArgs a = {};
    a.qu = (const bf16_t*)qu; a.qv = (const bf16_t*)qv; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.R = (const bf16_t*)R;
    a.o = (bf16_t*)out; a.lse_out = lse; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride;
    a.B = B; a.L = L; a.H = H; a.shift = shift; a.scale = scale;
    int st = flash_check(a, D, "relattn_flash_fwd");
    if (st) return st;
    if (!db1_aligned16(out)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_flash_fwd: out alignment");
    if ((probs == nullptr) != (mblk == nullptr) || !db1_aligned16(probs)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_flash_fwd: probs and mblk go together, probs 16-byte aligned");
    a.pt = (bf16_t*)probs; a.mblk = mblk;
    static Db1PerDeviceOnce attr_once;
    attr_once.run([] {
        hipFuncSetAttribute((const void*)relattn_flash_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W16_FWD_LDS);
        hipFuncSetAttribute((const void*)relattn_flash_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W16_FWD_LDS);
    });
    // plain causal window + kept probabilities (the training configuration): the hand-scheduled loop of relattn_flash_fwd2.hip
    const bool fwd2_off = db1_knob(DB1_KNOB_FLASH_FWD2, 1) == 0;   // A/B knob
    if (probs && shift >= L && !fwd2_off && g_fwd2_on) return g_fwd2_on == 2 ? db1_relattn_flash_fwd2_launch(&a, stream) : db1_relattn_flash_fwd3_launch(&a, stream);
    if (probs) relattn_flash_fwd_kernel<true><<<dim3(flash_grid(L / FA_BQ, H, B)), 512, W16_FWD_LDS, (hipStream_t)stream>>>(a);
    else relattn_flash_fwd_kernel<false><<<dim3(flash_grid(L / FA_BQ, H, B)), 512, W16_FWD_LDS, (hipStream_t)stream>>>(a);
    DB1_CHECK_LAUNCH("relattn_flash_fwd");
    return DB1_OK;
}

extern "C" int64_t db1_relattn_flash_probs_bytes(int B, int L, int H) {
    return (B > 0 && H > 0 && L > 0 && (L % FA_BK) == 0) ? (int64_t)B * H * flash_pt_tiles(L) * 512 * (int64_t)sizeof(bf16_t) : 0;
}
extern "C" int64_t db1_relattn_flash_bwd_workspace_bytes(int B, int L, int H, int have_probs) {
    const int64_t img = (int64_t)B * H * flash_pt_tiles(L) * 512 * (int64_t)sizeof(bf16_t);   // one set of fragment images, [B*H][triangle of (key block, query tile)][64][8] bf16
    if (have_probs) return ((int64_t)B * H * (L / FA_BK) * L + 64) * (int64_t)sizeof(float);   // the factors f, [B*H][L/32][L] floats (+ the over-read of the last row)
    return 2 * img;                                                                                  // P and dS
}

extern "C" int db1_relattn_flash_bwd(const void* qu, const void* qv, const void* k, const void* v, int64_t kv_row_stride,
                                     int64_t kv_batch_stride, const void* R, const void* out, const void* dout, const float* lse,
                                     float* delta, void* dq, void* dk, void* dv, int64_t dqkv_row_stride, int64_t dqkv_batch_stride,
                                     void* dT, int B, int L, int H, int D, int shift, float scale, const void* probs, const float* mblk,
                                     void* ws, int64_t ws_bytes, void* stream) {
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
    static Db1PerDeviceOnce attr_once;
    attr_once.run([] {
        hipFuncSetAttribute((const void*)relattn_flash_bwd_q_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W16_BQ_LDS);
        hipFuncSetAttribute((const void*)relattn_flash_bwd_q_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W16_BQ_LDS);
        hipFuncSetAttribute((const void*)relattn_flash_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KV16_LDS);
        hipFuncSetAttribute((const void*)relattn_flash_bwd_kv2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, KV2_LDS);
        hipFuncSetAttribute((const void*)relattn_flash_bwd_kv2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, KV2_LDS_F);
        hipFuncSetAttribute((const void*)relattn_flash_bwd_q2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Q2_LDS);
        hipFuncSetAttribute((const void*)relattn_flash_bwd_kv3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KV3_LDS);
    });
    const dim3 grid(flash_grid(L / FA_BQ, H, B));
    if ((probs == nullptr) != (mblk == nullptr) || !db1_aligned16(probs)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_flash_bwd: probs and mblk go together, probs 16-byte aligned");
    if (probs) {
        // the forward kept p~ and its maxima: nothing is recomputed on either side
        DB1_NEED_WS(ws, ws_bytes, db1_relattn_flash_bwd_workspace_bytes(B, L, H, 1), "relattn_flash_bwd (forward-stored probabilities)");
        a.pt = (bf16_t*)probs; a.mblk = const_cast<float*>(mblk);
        a.fblk = reinterpret_cast<float*>(ws);
        relattn_flash_bwd_q2_kernel<<<grid, 512, Q2_LDS, s>>>(a);
        DB1_CHECK_LAUNCH("relattn_flash_bwd_q2");
        // 32 keys per wave (256 per workgroup) where that still gives the chip two rounds of workgroups to balance the causal triangle with;
        // at the reference's micro-batch of 4 sequences it is 256 workgroups -- one per CU, the heaviest key tile alone at the end -- and the
        // 128-key kernel (512 workgroups) is 0.4 % of the step faster (r05: 132.2-132.5 k vs 131.6-132.0 k tok/s at 4 x GA 16)
        const int kv3_knob = db1_knob(DB1_KNOB_FLASH_KV3, -1);      // A/B knob: 0 / 1 force
        const bool kv3_on = kv3_knob >= 0 ? kv3_knob != 0 : (int64_t)B * H * (L / KV3_KEYS) >= 512;
        if (kv3_on && (L % KV3_KEYS) == 0) {   // a wave = 32 keys
            relattn_flash_bwd_kv3_kernel<<<dim3(flash_grid(L / KV3_KEYS, H, B)), 512, KV3_LDS, s>>>(a);
            DB1_CHECK_LAUNCH("relattn_flash_bwd_kv3");
            return DB1_OK;
        }
        relattn_flash_bwd_kv2_kernel<true><<<grid, 512, KV2_LDS_F, s>>>(a);
        DB1_CHECK_LAUNCH("relattn_flash_bwd_kv2");
        return DB1_OK;
    }
    const int64_t need = db1_relattn_flash_bwd_workspace_bytes(B, L, H, 0);
    if (ws && ws_bytes >= need && db1_aligned16(ws)) {
        // stored-probabilities backward: the query side leaves P and dS in the workspace, the key side is two contractions over them
        a.pbuf = (bf16_t*)ws;
        a.dsbuf = a.pbuf + (int64_t)B * H * flash_pt_tiles(L) * 512;
        relattn_flash_bwd_q_kernel<true><<<grid, 512, W16_BQ_LDS, s>>>(a);
        DB1_CHECK_LAUNCH("relattn_flash_bwd_q");
        relattn_flash_bwd_kv2_kernel<false><<<grid, 512, KV2_LDS, s>>>(a);
        DB1_CHECK_LAUNCH("relattn_flash_bwd_kv2");
        return DB1_OK;
    }
    relattn_flash_bwd_q_kernel<false><<<grid, 512, W16_BQ_LDS, s>>>(a);
    DB1_CHECK_LAUNCH("relattn_flash_bwd_q");
    relattn_flash_bwd_kv_kernel<<<grid, 512, KV16_LDS, s>>>(a);
    DB1_CHECK_LAUNCH("relattn_flash_bwd_kv");
    return DB1_OK;
}
