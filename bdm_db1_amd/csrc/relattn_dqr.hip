// dq_r = dT . R for the relative-position attention backward (transformer_xl.py:160-209): per (head, batch)
//     dq_r[b, i, h, :] = sum_{dist <= i} dT[h, b, i, dist] * R[dist, h, :]                (dT = dS re-indexed by distance, bf16)
// This contraction is HBM-bound on dT (1.07 GB per layer at 64 x 1024 tokens, a 128-wide output): as a batched tile GEMM it ran at
// 2.7 TB/s (441 us): 128 bytes per row and k-tile, and causal k-loops of 4..16 tiles that never filled the 3-stage pipeline
// (this kernel: 373 us on a cold dT, 302 us inside the step = 5.0 TB/s of its 1.51 GB; [128 rows][128 dist] tiles were slower: 485 us).
// Here dT is a STREAM and R is stationary:
//   * one workgroup per CU works for one head; wave w keeps R_h^T for its 16 output columns d in registers for all distances
//     (32 k-steps x 8 bf16 = 128 VGPRs, loaded once from a transposed copy of R);
//   * the (batch, 64-row tile) items of the head are walked as ONE continuous stream of [64 rows][128 dist] tiles (256 contiguous
//     bytes per row, only the tiles up to the causal diagonal) through a 4-stage LDS ring filled by the LDS-DMA three tiles ahead;
//   * per tile and wave: 16 row fragments (ds_read_b128, the B operand: lane = row i) x the stationary A fragments ->
//     out^T[d][i] accumulators, so a lane ends with 4 consecutive d of one row (8-byte stores, 256 B per row over the 8 waves).
#include "db1_common.h"

typedef __attribute__((ext_vector_type(8))) short dqr_bf16x8;
typedef __attribute__((ext_vector_type(4))) float dqr_f32x4;
#define DQR_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
typedef __attribute__((address_space(3))) const dqr_bf16x8* dqr_lds_rd;

#define DQR_ROWS 64
#define DQR_KT 128                 // distances per tile
#define DQR_TILE_BYTES (DQR_ROWS * DQR_KT * 2)
#define DQR_STAGES 4                // (8 stages / 7 tiles in flight were slower: 430 vs 373 us -- the per-tile barrier + issue work bounds it, not latency)
#define DQR_MAX_L 1024             // 32 k-steps of R^T per lane in registers

struct DqrArgs {
    const bf16_t* dT; const bf16_t* Rt; bf16_t* out;
    int B, L, H, wph;              // wph = workgroups per head
    int64_t o_rs, o_bs;            // row / batch strides of out (elements); head h at + h * 128
    // fused mode (part != nullptr): out holds dq_k = the (q+u).k branch on entry and dq = dq_k + dq_r on exit; the column sums of dq_k
    // and of dq_r over this workgroup's rows go to part[0][j][h*128 + c] / part[1][j][h*128 + c] (j = workgroup of the head; a reduce
    // kernel adds the rows of that table in order).  The 64 x 128 dq_k tile of an item travels through the SAME LDS-DMA stream as the
    // dT tiles -- one more tile per item -- so the counted vmcnt waits of the stream stay valid (plain global loads in the epilogue sit in
    // the same queue and broke them: measured and reverted before this version).
    float* part;
    // groups (ngroups > 1): the batch is ngroups blocks of B / ngroups sequences and every block has its OWN R (a gradient-accumulation window
    // run through one backward: each micro-step drew its own position-table dropout).  wph is a multiple of ngroups: workgroup j of a head
    // works for block j % ngroups only, so its stationary R^T never changes; Rt of block g at + g * rt_gs elements.
    int ngroups;
    int64_t rt_gs;
};

__device__ __forceinline__ int dqr_swz(int row) { return ((((row & 7) ^ ((row & 8) >> 1))) << 1) | ((row >> 3) & 1); }  // natural-order row fragments

__device__ __forceinline__ void dqr_glds(const void* src, unsigned dst_lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst_lds) : "memory");
}
template <int N> __device__ __forceinline__ void dqr_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512, 1) void relattn_dqr_kernel(DqrArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x / p.wph, j = blockIdx.x % p.wph;
    const int L = p.L, NT = L / DQR_ROWS;
    const int a = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)DQR_LDS_PTR(char, smem);

    // stationary operand: R_h^T rows d = 16 wave + a, k = 32 ks + 8 g .. +7
    dqr_bf16x8 rfr[DQR_MAX_L / 32];
    {
        const bf16_t* rt = p.Rt + (int64_t)(j % p.ngroups) * p.rt_gs + ((int64_t)h * 128 + 16 * wave + a) * L + 8 * g;
#pragma unroll
        for (int ks = 0; ks < DQR_MAX_L / 32; ks++)
            rfr[ks] = ks * 32 < L ? *reinterpret_cast<const dqr_bf16x8*>(rt + ks * 32) : (dqr_bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    // row-fragment read addresses inside a stage: row = 16 rt + a, 16-byte chunk 4 ks + g of the 256-byte row
    unsigned faddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) { faddr[ks] = lds0 + a * 256 + (((4 * ks + g) ^ dqr_swz(a)) << 4); asm volatile("" : "+v"(faddr[ks])); }
    // staging: a tile = 16 pieces of 4 rows; wave w moves pieces 2w, 2w+1 (rows 8w .. 8w+7)
    const int srow = 8 * wave + (lane >> 4);
    const int schunk0 = ((lane & 15) ^ dqr_swz(srow & 15)) << 3, schunk1 = ((lane & 15) ^ dqr_swz((srow + 4) & 15)) << 3;

    // the stream of tiles: items t = j, j + wph, ... (t -> batch b = t / NT, row tile (t + b) % NT: every workgroup sees every row tile
    // equally often), and inside an item the k-tiles 0 .. (i0 + 63) / 128
    const int grp = j % p.ngroups, jj = j / p.ngroups, wpg = p.wph / p.ngroups, bper = p.B / p.ngroups;   // (one group: grp = 0, jj = j, wpg = wph)
    const int n_items = bper * NT;
    const int fused = p.part != nullptr ? 1 : 0;
    struct Cur { int t, kt, nk, b, i0; };   // nk = tiles of the item in the stream: its dT k-tiles (+ the dq_k tile in fused mode, last)
    auto item_of = [&](int t, Cur& c) __attribute__((always_inline)) {
        c.t = t; c.kt = 0; c.b = grp * bper + t / NT; c.i0 = ((t + c.b) % NT) * DQR_ROWS; c.nk = (c.i0 + DQR_ROWS - 1) / DQR_KT + 1 + fused;
    };
    auto advance = [&](Cur& c) __attribute__((always_inline)) {  // next tile of the stream (c.t >= n_items: past the end)
        if (++c.kt >= c.nk) { const int t = c.t + wpg; if (t < n_items) item_of(t, c); else { c.t = t; c.kt = 0; c.nk = 1; } }
    };
    auto stage = [&](const Cur& c, int st) __attribute__((always_inline)) {
        const unsigned dst = lds0 + st * DQR_TILE_BYTES + wave * 2048;
        if (fused && c.kt == c.nk - 1) {   // the item's dq_k rows (128 columns of head h), same 64 x 256-byte image as a dT tile
            const bf16_t* src = p.out + (int64_t)c.b * p.o_bs + (int64_t)(c.i0 + srow) * p.o_rs + h * 128;
            dqr_glds(src + schunk0, dst);
            dqr_glds(src + (int64_t)4 * p.o_rs + schunk1, dst + 1024);
        } else {
            const bf16_t* src = p.dT + (((int64_t)h * p.B + c.b) * L + c.i0 + srow) * L + c.kt * DQR_KT;
            dqr_glds(src + schunk0, dst);
            dqr_glds(src + (int64_t)4 * L + schunk1, dst + 1024);
        }
    };
    Cur cs, cc;                      // staging cursor (DQR_STAGES - 1 tiles ahead), compute cursor
    if (jj >= n_items) {             // (more workgroups than items: never at the model's sizes) -- its partial rows must still be defined
        if (p.part && tid < 128) {
            p.part[(int64_t)j * p.H * 128 + h * 128 + tid] = 0.f;
            p.part[((int64_t)p.wph + j) * p.H * 128 + h * 128 + tid] = 0.f;
        }
        return;
    }
    item_of(jj, cs);
    cc = cs;
    int issued = 0;
#pragma unroll
    for (int s = 0; s < DQR_STAGES - 1; s++) {
        if (cs.t < n_items) { stage(cs, s); issued++; advance(cs); }
    }
    dqr_f32x4 acc[4];
#pragma unroll
    for (int rt = 0; rt < 4; rt++) acc[rt] = (dqr_f32x4){0.f, 0.f, 0.f, 0.f};
    float sum_k[4] = {0.f, 0.f, 0.f, 0.f}, sum_r[4] = {0.f, 0.f, 0.f, 0.f};   // fused mode: column sums over this lane's rows (all items)
    // fused mode: this lane's 4 dq_k values of row 16 rt + a inside a staged dq_k tile: columns 16 wave + 4 g .. +3 = chunk 2 wave + (g >> 1)
    unsigned qaddr = lds0 + a * 256 + (((2 * wave + (g >> 1)) ^ dqr_swz(a)) << 4) + (g & 1) * 8;
    asm volatile("" : "+v"(qaddr));
    int st = 0;  // stage of the compute cursor's tile
    for (; cc.t < n_items;) {
        // the tile of the compute cursor must have landed: all but the (issued - 1) pieces pairs issued after it
        switch (issued) {
            case 7: dqr_vmcnt<12>(); break; case 6: dqr_vmcnt<10>(); break; case 5: dqr_vmcnt<8>(); break; case 4: dqr_vmcnt<6>(); break;
            case 3: dqr_vmcnt<4>(); break; case 2: dqr_vmcnt<2>(); break; default: dqr_vmcnt<0>(); break;
        }
        __syncthreads();             // tile visible to all waves; the stage consumed last step is free
        if (cs.t < n_items) { stage(cs, (st + DQR_STAGES - 1) % DQR_STAGES); advance(cs); } else issued--;
        const unsigned sb = st * DQR_TILE_BYTES;
        // k-step kk of the tile is k-step cc.kt * 4 + kk of R^T: a runtime index into the register array -> a switch on cc.kt
#define DQR_TILE(KT)                                                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < 4; ks++) {                                                               \
            _Pragma("unroll") for (int rt = 0; rt < 4; rt++) {                                                           \
                const dqr_bf16x8 bf = *(dqr_lds_rd)(size_t)(faddr[ks] + sb + rt * 4096);                                 \
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rfr[(KT) * 4 + ks], bf, acc[rt], 0, 0, 0);             \
            }                                                                                                            \
        }
        const bool dq_tile = fused && cc.kt == cc.nk - 1;
        if (!dq_tile) {
            switch (cc.kt) {
                case 0: DQR_TILE(0) break; case 1: DQR_TILE(1) break; case 2: DQR_TILE(2) break; case 3: DQR_TILE(3) break;
                case 4: DQR_TILE(4) break; case 5: DQR_TILE(5) break; case 6: DQR_TILE(6) break; default: DQR_TILE(7) break;
            }
        }
        if (cc.kt == cc.nk - 1) {    // item done: out[b][i0 + 16 rt + a][h][16 wave + 4 g .. +3]
#pragma unroll
            for (int rt = 0; rt < 4; rt++) {
                float v[4] = {acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]};
                if (fused) {         // dq = dq_k + dq_r (one rounding), du += colsum(dq_k), dv_bias += colsum(dq_r)
                    typedef __attribute__((ext_vector_type(2))) unsigned dqr_u32x2;
                    const dqr_u32x2 qk = *(__attribute__((address_space(3))) const dqr_u32x2*)(size_t)(qaddr + sb + rt * 4096);
                    const float k0 = __uint_as_float(qk[0] << 16), k1 = __uint_as_float(qk[0] & 0xffff0000u);
                    const float k2 = __uint_as_float(qk[1] << 16), k3 = __uint_as_float(qk[1] & 0xffff0000u);
                    sum_k[0] += k0; sum_k[1] += k1; sum_k[2] += k2; sum_k[3] += k3;
                    sum_r[0] += v[0]; sum_r[1] += v[1]; sum_r[2] += v[2]; sum_r[3] += v[3];
                    v[0] += k0; v[1] += k1; v[2] += k2; v[3] += k3;
                }
                uint2 o;
                o.x = f2bf_pk(v[0], v[1]);
                o.y = f2bf_pk(v[2], v[3]);
                *reinterpret_cast<uint2*>(p.out + (int64_t)cc.b * p.o_bs + (int64_t)(cc.i0 + 16 * rt + a) * p.o_rs + h * 128 + 16 * wave + 4 * g) = o;
                acc[rt] = (dqr_f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        advance(cc);
        st = (st + 1) % DQR_STAGES;
    }
    if (fused) {   // this workgroup's column sums: over the 16 row lanes (fixed butterfly order), then one row of the partial table per sum
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { sum_k[c] += __shfl_xor(sum_k[c], o, 64); sum_r[c] += __shfl_xor(sum_r[c], o, 64); }
        }
        if (a == 0) {
            const int64_t HD = (int64_t)p.H * 128, col = (int64_t)h * 128 + 16 * wave + 4 * g;
            float* pk = p.part + (int64_t)j * HD + col;
            float* pr = p.part + ((int64_t)p.wph + j) * HD + col;
#pragma unroll
            for (int c = 0; c < 4; c++) { pk[c] = sum_k[c]; pr[c] = sum_r[c]; }
        }
    }
}

// R [nd][H * 128] -> Rt [H * 128][nd]: 64 x 64 tiles through LDS, 8-byte accesses on both sides (a lane reads 4 consecutive columns of a row
// and writes 4 consecutive rows' worth of one output row); nd and H * 128 are multiples of 64 here (db1_relattn_dqr_supported).
// (32 x 32 tiles with 2-byte accesses took 59 us for the 16 tables of a 4 x GA 16 window: 2.2 TB/s.)
__global__ __launch_bounds__(256) void relattn_dqr_transpose_kernel(const bf16_t* __restrict__ R, bf16_t* __restrict__ Rt, int nd, int HD, int64_t r_rs,
                                                                    int64_t r_gs, int64_t rt_gs) {
    __shared__ bf16_t tile[64][64 + 4];
    R += (int64_t)blockIdx.z * r_gs;
    Rt += (int64_t)blockIdx.z * rt_gs;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, t = threadIdx.x;
    const int lr = t >> 4, lc = (t & 15) * 4;      // 16 rows x 16 four-column groups per pass
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int r = lr + 16 * p;
        const uint2 v = *reinterpret_cast<const uint2*>(R + (int64_t)(r0 + r) * r_rs + c0 + lc);
        tile[r][lc] = (bf16_t)(v.x & 0xffffu); tile[r][lc + 1] = (bf16_t)(v.x >> 16);
        tile[r][lc + 2] = (bf16_t)(v.y & 0xffffu); tile[r][lc + 3] = (bf16_t)(v.y >> 16);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int oc = lr + 16 * p;                // output row = input column
        const unsigned e0 = tile[lc][oc], e1 = tile[lc + 1][oc], e2 = tile[lc + 2][oc], e3 = tile[lc + 3][oc];
        *reinterpret_cast<uint2*>(Rt + (int64_t)(c0 + oc) * nd + r0 + lc) = make_uint2(e0 | (e1 << 16), e2 | (e3 << 16));
    }
}

extern "C" int db1_relattn_dqr_supported(int B, int L, int H, int D, int dt) {
    return (dt == DB1_BF16 && D == 128 && B > 0 && H > 0 && H <= 256 && L >= 128 && (L % 128) == 0 && L <= DQR_MAX_L) ? 1 : 0;
}

static inline int dqr_wph(int H) { return 256 / H > 0 ? 256 / H : 1; }
static inline int64_t dqr_rt_bytes(int L, int H) { return (((int64_t)H * 128 * L * (int64_t)sizeof(bf16_t)) + 255) & ~(int64_t)255; }
// R^T, and (fused entry point) the per-workgroup column-sum partials [2][wph][H * 128]
extern "C" int64_t db1_relattn_dqr_workspace_bytes(int L, int H) { return dqr_rt_bytes(L, H) + 2 * (int64_t)dqr_wph(H) * H * 128 * (int64_t)sizeof(float); }

/* dq_r[b, i, h, :] = sum_dist dT[h, b, i, dist] * R[dist, h, :]; dT [H, B, L, L] bf16 (zero for dist > i), R [L, H, 128] with row stride r_rs,
 * out [B, L, H, 128] with row / batch strides (elements) */
__global__ __launch_bounds__(256) void dqr_part_reduce_kernel(const float* __restrict__ part, float* acc, int n, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float t = 0.f;
    for (int j = 0; j < n; j++) t += part[(int64_t)j * cols + c];   // workgroup order: deterministic
    acc[c] += t;
}

static int dqr_run(const void* dT, const void* R, int64_t r_row_stride, void* out, int64_t out_row_stride, int64_t out_batch_stride, float* du_acc,
                   float* dv_acc, int B, int L, int H, int D, void* ws, int64_t ws_bytes, void* stream, float* parts_out = nullptr, int ngroups = 1,
                   int64_t r_group_stride = 0);

extern "C" int db1_relattn_dqr(const void* dT, const void* R, int64_t r_row_stride, void* out, int64_t out_row_stride, int64_t out_batch_stride,
                               int B, int L, int H, int D, void* ws, int64_t ws_bytes, void* stream) {
    return dqr_run(dT, R, r_row_stride, out, out_row_stride, out_batch_stride, nullptr, nullptr, B, L, H, D, ws, ws_bytes, stream);
}
/* the same stream, finishing the query gradient: dq[b,i,h,:] = dq_k + dq_r in place over dq_k (one rounding), du_acc[h*128 + c] += sum_{b,i}
 * dq_k, dv_acc[h*128 + c] += sum_{b,i} dq_r (float32; per-workgroup partials added in a fixed order) */
extern "C" int db1_relattn_dqr_fused(const void* dT, const void* R, int64_t r_row_stride, void* dq, int64_t dq_row_stride, int64_t dq_batch_stride,
                                     float* du_acc, float* dv_acc, int B, int L, int H, int D, void* ws, int64_t ws_bytes, void* stream) {
    if (!du_acc || !dv_acc) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_dqr_fused: null accumulator");
    if ((dq_row_stride % 8) || (dq_batch_stride % 8)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_dqr_fused: dq strides must be multiples of 8 elements (16-byte LDS-DMA pieces)");
    return dqr_run(dT, R, r_row_stride, dq, dq_row_stride, dq_batch_stride, du_acc, dv_acc, B, L, H, D, ws, ws_bytes, stream);
}

/* the fused form WITHOUT its two reduces: the per-workgroup column-sum partials stay in `parts` = [2][db1_relattn_dqr_parts_rows(H)][H * 128]
 * float32 (first the dq_k sums -> du, then the dq_r sums -> dv) for the caller to add up later (db1_colsum_acc over each [rows, H * 128]
 * half): gradient accumulation reduces once per optimizer step instead of twice per layer and micro-step */
extern "C" int db1_relattn_dqr_parts_rows(int H) { return dqr_wph(H); }
extern "C" int db1_relattn_dqr_fused_parts(const void* dT, const void* R, int64_t r_row_stride, void* dq, int64_t dq_row_stride, int64_t dq_batch_stride,
                                           float* parts, int B, int L, int H, int D, void* ws, int64_t ws_bytes, void* stream) {
    if (!parts || !db1_aligned16(parts)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_dqr_fused_parts: parts");
    if ((dq_row_stride % 8) || (dq_batch_stride % 8)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_dqr_fused_parts: dq strides must be multiples of 8 elements");
    return dqr_run(dT, R, r_row_stride, dq, dq_row_stride, dq_batch_stride, parts, parts, B, L, H, D, ws, ws_bytes, stream, parts);
}

/* the fused form over a batch of `ngroups` blocks of B / ngroups sequences, block g with its own R at R + g * r_group_stride elements (every
 * micro-step of a gradient-accumulation window has its own position-table dropout, hence its own R; transformer_xl.py:138,575): ONE launch
 * for the whole window.  ngroups must divide B and the head's workgroup count (256 / H); ws >= db1_relattn_dqr_groups_workspace_bytes. */
extern "C" int64_t db1_relattn_dqr_groups_workspace_bytes(int L, int H, int ngroups) {
    return (int64_t)(ngroups > 0 ? ngroups : 1) * dqr_rt_bytes(L, H) + 2 * (int64_t)dqr_wph(H) * H * 128 * (int64_t)sizeof(float);
}
extern "C" int db1_relattn_dqr_groups_supported(int B, int L, int H, int D, int dt, int ngroups) {
    return (db1_relattn_dqr_supported(B, L, H, D, dt) && ngroups >= 1 && B % ngroups == 0 && dqr_wph(H) % ngroups == 0) ? 1 : 0;
}
extern "C" int db1_relattn_dqr_fused_groups(const void* dT, const void* R, int64_t r_row_stride, int64_t r_group_stride, int ngroups, void* dq,
                                            int64_t dq_row_stride, int64_t dq_batch_stride, float* du_acc, float* dv_acc, int B, int L, int H, int D,
                                            void* ws, int64_t ws_bytes, void* stream) {
    if (!du_acc || !dv_acc) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_dqr_fused_groups: null accumulator");
    if ((dq_row_stride % 8) || (dq_batch_stride % 8)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_dqr_fused_groups: dq strides must be multiples of 8 elements");
    if (!db1_relattn_dqr_groups_supported(B, L, H, D, DB1_BF16, ngroups)) DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_dqr_fused_groups: ngroups=%d must divide B=%d and %d", ngroups, B, dqr_wph(H));
    return dqr_run(dT, R, r_row_stride, dq, dq_row_stride, dq_batch_stride, du_acc, dv_acc, B, L, H, D, ws, ws_bytes, stream, nullptr, ngroups, r_group_stride);
}

static int dqr_run(const void* dT, const void* R, int64_t r_row_stride, void* out, int64_t out_row_stride, int64_t out_batch_stride, float* du_acc,
                   float* dv_acc, int B, int L, int H, int D, void* ws, int64_t ws_bytes, void* stream, float* parts_out, int ngroups, int64_t r_group_stride) {
    if (!db1_relattn_dqr_supported(B, L, H, D, DB1_BF16)) DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_dqr: needs bf16, d_head = 128, L %% 128 == 0, L <= 1024 (got L=%d D=%d)", L, D);
    if (!dT || !R || !out) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_dqr: null buffer");
    if (!db1_aligned16(dT) || !db1_aligned16(out) || (out_row_stride % 4) || (out_batch_stride % 4)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_dqr: alignment");
    hipStream_t st = (hipStream_t)stream;
    DB1_NEED_WS(ws, ws_bytes, db1_relattn_dqr_groups_workspace_bytes(L, H, ngroups), "relattn_dqr");
    bf16_t* Rt = (bf16_t*)ws;
    const int64_t rt_gs = dqr_rt_bytes(L, H) / (int64_t)sizeof(bf16_t);
    if ((r_row_stride % 4) || (r_group_stride % 4) || (((uintptr_t)R) & 7)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_dqr: R must be 8-byte aligned with strides that are multiples of 4 elements");
    relattn_dqr_transpose_kernel<<<dim3((unsigned)(H * 128 / 64), (unsigned)(L / 64), (unsigned)ngroups), 256, 0, st>>>((const bf16_t*)R, Rt, L, H * 128,
                                                                                                                                   r_row_stride, r_group_stride, rt_gs);
    DB1_CHECK_LAUNCH("relattn_dqr transpose");
    DqrArgs a;
    a.dT = (const bf16_t*)dT; a.Rt = Rt; a.out = (bf16_t*)out; a.B = B; a.L = L; a.H = H;
    a.wph = dqr_wph(H);
    a.o_rs = out_row_stride; a.o_bs = out_batch_stride;
    a.ngroups = ngroups; a.rt_gs = rt_gs;
    a.part = parts_out ? parts_out : (du_acc ? (float*)((char*)ws + (int64_t)ngroups * dqr_rt_bytes(L, H)) : nullptr);
    static Db1PerDeviceOnce attr_once;
    attr_once.run([] { hipFuncSetAttribute((const void*)relattn_dqr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DQR_STAGES * DQR_TILE_BYTES); });
    relattn_dqr_kernel<<<dim3((unsigned)(H * a.wph)), 512, DQR_STAGES * DQR_TILE_BYTES, st>>>(a);
    DB1_CHECK_LAUNCH("relattn_dqr");
    if (du_acc && !parts_out) {
        const int cols = H * 128;
        dqr_part_reduce_kernel<<<(cols + 255) / 256, 256, 0, st>>>(a.part, du_acc, a.wph, cols);
        dqr_part_reduce_kernel<<<(cols + 255) / 256, 256, 0, st>>>(a.part + (int64_t)a.wph * cols, dv_acc, a.wph, cols);
        DB1_CHECK_LAUNCH("relattn_dqr reduce");
    }
    return DB1_OK;
}
