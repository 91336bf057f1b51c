// Relative-position flash attention for gfx950 (bf16, d_head = 128): Transformer-XL scores
//     s[i,j] = ((q_i+u).k_j + (q_i+v).R[i-j]) / sqrt(d),   visible iff  i - shift < j <= i
// (closed form of AC + _rel_shift(BD) + mask, transformer_xl.py:98-110,160-209,551-567) with online softmax
// and P.V fused, never materialising an (L x L) tensor in HBM.
//
// Forward, per workgroup: 128 queries (4 waves x 32) of one (batch, head); loop over 32-key blocks.
//   S^T = K.Qu^T              "swapped" MFMA (v_mfma_f32_32x32x16_bf16): lane = query column, the 16 accumulator
//                             registers = keys -> row statistics are lane-local (+1 half-swap), P^T feeds P.V as the
//                             B operand straight from registers (no P round trip);
//   T   = Qv.Rband^T          non-swapped, 64 distances per 32x32 block (band i-j of the block); written to a per-wave
//                             LDS scratch [32 q][64 dist] and read back SKEWED (element (a, a-b+31)): both the write
//                             (lanes = consecutive distances) and the read (lane stride 65 words) are bank-conflict free;
//   O^T += V^T.P^T            V^T fragments by ds_read_b64_tr_b16 from the row-major V tile.
// K/V tiles and a 256-row ring of R rows (the band slides by 32 distances per key block) are staged with
// global_load_lds (16 B/lane); 16-B chunks are XOR-swizzled on the SOURCE side (K, R: chunk ^ (row & 15);
// V: chunk ^ ((row & 3) << 2)) so ds_read_b128 / tr reads are conflict-free.
// Inputs qu = q+u and qv = q+v_bias are materialised once per layer by db1_relattn_add_head_bias.
#include "db1_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

#define FA_D 128
#define FA_BQ 128
#define FA_BK 32
#define FA_RING 256
#define FA_TW_BYTES 8704                                  // per-wave scratch: T [32][64] f32 (8192) / O staging [32][136] bf16 (8704)
#define FA_OFF_K 0
#define FA_OFF_V 8192
#define FA_OFF_R 16384
#define FA_OFF_T (16384 + FA_RING * 256)
#define FA_LDS_BYTES (FA_OFF_T + 4 * FA_TW_BYTES)

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

// one 1 KiB global_load_lds piece = 4 rows of 256 B; lane -> (row = lane >> 4, chunk position = lane & 15)
__device__ __forceinline__ void glds_rows4(const bf16_t* row_ptr, int swz, char* lds_piece, int lane) {
    const int c = (lane & 15) ^ swz;
    __builtin_amdgcn_global_load_lds(row_ptr + c * 8, LDS_PTR(void, lds_piece), 16, 0, 0);
}

__device__ __forceinline__ bf16x8_t pack8(const float* p) {
    bf16x8_t o;
#pragma unroll
    for (int t = 0; t < 8; t++) o[t] = (short)f2bf(p[t]);
    return o;
}

// V^T (or any row-major [key][d] tile read as [d][key]) A-fragment for 32x32x16: 16 keys starting at key0, d-block db.
// slot t of lane (d = lane & 31, hb) <-> key key0 + (t & 3) + 8 * (t >> 2) + 4 * hb, matching the C-layout rows.
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int key0, int db, int lane, int swz_shift) {
    const int g4 = lane >> 4, t = lane & 15, hb = g4 >> 1;
    const int gran = ((32 * db + 16 * (g4 & 1)) >> 2) + (t & 3);  // 8-byte granule inside the 256-B row
    bf16x8_t out;
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
        const int row = key0 + 4 * hb + 8 * h2 + (t >> 2);
        const int chunk = (gran >> 1) ^ ((row & 3) << swz_shift);
        const int off = row * 256 + chunk * 16 + (gran & 1) * 8;
        bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4_t, const_cast<char*>(tile) + off));
        out[h2 * 4 + 0] = v[0]; out[h2 * 4 + 1] = v[1]; out[h2 * 4 + 2] = v[2]; out[h2 * 4 + 3] = v[3];
    }
    return out;
}

// row-major [row][128] tile with chunk ^ (row & 15) swizzle: fragment "row (lane & 31), k = ks*16 + (lane>>5)*8 .. +8"
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int row, int ks, int lane) {
    const int chunk = (ks * 2 + (lane >> 5)) ^ (row & 15);
    return *reinterpret_cast<const bf16x8_t*>(tile + row * 256 + chunk * 16);
}

__global__ __launch_bounds__(256, 1) void relattn_flash_fwd_kernel(FlashArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqt = p.L / FA_BQ;
    const int qt = nqt - 1 - (int)blockIdx.x;  // heavy (late) query tiles first
    const int h = blockIdx.y, b = blockIdx.z;
    const int H = p.H, L = p.L, HD = H * FA_D;
    const int i0 = qt * FA_BQ, iw = i0 + 32 * wave;
    const int a = lane & 31, hb = lane >> 5;
    char* Ks = smem + FA_OFF_K;
    char* Vs = smem + FA_OFF_V;
    char* Rr = smem + FA_OFF_R;
    float* Tw = reinterpret_cast<float*>(smem + FA_OFF_T + wave * FA_TW_BYTES);

    const bf16_t* qu = p.qu + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* qv = p.qv + ((int64_t)b * L) * HD + h * FA_D;
    const bf16_t* kg = p.k + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* vg = p.v + (int64_t)b * p.kv_bs + h * FA_D;
    const bf16_t* Rg = p.R + h * FA_D;

    // Q fragments (A and B operands share the register image): row iw + a, k = ks*16 + hb*8
    bf16x8_t fqu[8], fqv[8];
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
        fqu[ks] = *reinterpret_cast<const bf16x8_t*>(qu + (int64_t)(iw + a) * HD + ks * 16 + hb * 8);
        fqv[ks] = *reinterpret_cast<const bf16x8_t*>(qv + (int64_t)(iw + a) * HD + ks * 16 + hb * 8);
    }
    int jlo = i0 - p.shift + 1;
    if (jlo < 0) jlo = 0;
    const int jb_lo = jlo / FA_BK, jb_hi = (i0 + FA_BQ - 1) / FA_BK;
    // prologue: ring rows for distances [i0 - j0lo, i0 - j0lo + 128)
    {
        const int dbase = i0 - jb_lo * FA_BK;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int piece = wave * 8 + it;                 // 32 pieces x 4 rows
            const int slot0 = (dbase + piece * 4) & (FA_RING - 1);   // wave-uniform, multiple of 4
            const int dist = dbase + piece * 4 + (lane >> 4);
            const int slot = slot0 + (lane >> 4);
            int gr = dist < 0 ? 0 : (dist > L - 1 ? L - 1 : dist);
            glds_rows4(Rg + (int64_t)gr * HD, slot & 15, Rr + slot0 * 256, lane);
        }
    }
    f32x16 acc_o[4];
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc_o[db][r] = 0.f;
    float m_i = -1.0e30f, l_i = 0.f;

    for (int jb = jb_lo; jb <= jb_hi; jb++) {
        const int j0 = jb * FA_BK;
        __syncthreads();  // every wave is done with the previous K/V tiles and the ring rows about to be replaced
        {
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int piece = wave * 2 + it;             // 8 pieces x 4 rows = 32 rows
                const int r = piece * 4 + (lane >> 4);
                glds_rows4(kg + (int64_t)(j0 + r) * p.kv_rs, r & 15, Ks + piece * 1024, lane);
                glds_rows4(vg + (int64_t)(j0 + r) * p.kv_rs, (r & 3) << 2, Vs + piece * 1024, lane);
                const int slot0 = (i0 - j0 - 32 + piece * 4) & (FA_RING - 1);
                const int dist = i0 - j0 - 32 + r;
                const int slot = slot0 + (lane >> 4);
                int gr = dist < 0 ? 0 : (dist > L - 1 ? L - 1 : dist);
                glds_rows4(Rg + (int64_t)gr * HD, slot & 15, Rr + slot0 * 256, lane);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // wave-uniform skip of blocks that are entirely outside this wave's visibility window
        if (j0 > iw + 31 || j0 + 31 <= iw - p.shift) continue;

        f32x16 acc_s;
#pragma unroll
        for (int r = 0; r < 16; r++) acc_s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ks++) acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks, a, ks, lane), fqu[ks], acc_s, 0, 0, 0);
        // T = Qv . Rband^T for distances iw - j0 - 31 + [0, 64)
#pragma unroll
        for (int blk = 0; blk < 2; blk++) {
            f32x16 acc_t;
#pragma unroll
            for (int r = 0; r < 16; r++) acc_t[r] = 0.f;
            const int slot = (iw - j0 - 31 + 32 * blk + a) & (FA_RING - 1);
#pragma unroll
            for (int ks = 0; ks < 8; ks++) acc_t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fqv[ks], row_frag(Rr, slot, ks, lane), acc_t, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) Tw[crow(r, hb) * 64 + 32 * blk + a] = acc_t[r];
        }
        // skewed read-back + scale + mask (this lane: query iw + a; register r: key j0 + crow(r, hb))
        float s[16];
        float mblk = -1.0e30f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int bk = crow(r, hb);
            const float bd = Tw[a * 64 + a - bk + 31];
            const int i = iw + a, j = j0 + bk;
            const bool vis = (j <= i) && (j > i - p.shift);
            s[r] = vis ? (acc_s[r] + bd) * p.scale : -1.0e30f;
            mblk = fmaxf(mblk, s[r]);
        }
        mblk = fmaxf(mblk, __shfl_xor(mblk, 32, 64));
        const float m_new = fmaxf(m_i, mblk);
        const float alpha = __expf(m_i - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) { s[r] = __expf(s[r] - m_new); rs += s[r]; }
        rs += __shfl_xor(rs, 32, 64);
        l_i = l_i * alpha + rs;
        m_i = m_new;
#pragma unroll
        for (int db = 0; db < 4; db++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc_o[db][r] *= alpha;
        const bf16x8_t pb0 = pack8(s), pb1 = pack8(s + 8);
#pragma unroll
        for (int db = 0; db < 4; db++) {
            acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vs, 0, db, lane, 2), pb0, acc_o[db], 0, 0, 0);
            acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vs, 16, db, lane, 2), pb1, acc_o[db], 0, 0, 0);
        }
    }
    // epilogue: O[q][d] = acc_o^T / l, staged through the wave's scratch so that global stores are whole 256-B rows
    const float inv = 1.f / l_i;
    bf16_t* Ow = reinterpret_cast<bf16_t*>(Tw);  // [32][136] bf16 (row stride 272 B)
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
            uint2 o;
            o.x = (unsigned)f2bf(acc_o[db][rq * 4 + 0] * inv) | ((unsigned)f2bf(acc_o[db][rq * 4 + 1] * inv) << 16);
            o.y = (unsigned)f2bf(acc_o[db][rq * 4 + 2] * inv) | ((unsigned)f2bf(acc_o[db][rq * 4 + 3] * inv) << 16);
            *reinterpret_cast<uint2*>(Ow + a * 136 + 32 * db + 8 * rq + 4 * hb) = o;
        }
    bf16_t* og = p.o + ((int64_t)b * L + iw) * HD + h * FA_D;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int row = it * 4 + (lane >> 4), ch = lane & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(Ow + row * 136 + ch * 8);
        *reinterpret_cast<uint4*>(og + (int64_t)row * HD + ch * 8) = v;
    }
    if (hb == 0) p.lse_out[((int64_t)b * H + h) * L + iw + a] = m_i + logf(l_i);
}

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
                                     int shift, float scale, void* stream) {
    FlashArgs a = {};
    a.qu = (const bf16_t*)qu; a.qv = (const bf16_t*)qv; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.R = (const bf16_t*)R;
    a.o = (bf16_t*)out; a.lse_out = lse; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride;
    a.B = B; a.L = L; a.H = H; a.shift = shift; a.scale = scale;
    int st = flash_check(a, D, "relattn_flash_fwd");
    if (st) return st;
    if (!db1_aligned16(out)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_flash_fwd: out alignment");
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)relattn_flash_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS_BYTES); attr = true; }
    dim3 grid((unsigned)(L / FA_BQ), (unsigned)H, (unsigned)B);
    relattn_flash_fwd_kernel<<<grid, 256, FA_LDS_BYTES, (hipStream_t)stream>>>(a);
    DB1_CHECK_LAUNCH("relattn_flash_fwd");
    return DB1_OK;
}

extern "C" int db1_relattn_flash_bwd(const void* qu, const void* qv, const void* k, const void* v, int64_t kv_row_stride,
                                     int64_t kv_batch_stride, const void* R, const void* out, const void* dout, const float* lse,
                                     float* delta, void* dq, void* dk, void* dv, int64_t dqkv_row_stride, int64_t dqkv_batch_stride,
                                     void* dT, int B, int L, int H, int D, int shift, float scale, void* stream) {
    (void)qu; (void)qv; (void)k; (void)v; (void)kv_row_stride; (void)kv_batch_stride; (void)R; (void)out; (void)dout; (void)lse; (void)delta;
    (void)dq; (void)dk; (void)dv; (void)dqkv_row_stride; (void)dqkv_batch_stride; (void)dT; (void)B; (void)L; (void)H; (void)D; (void)shift;
    (void)scale; (void)stream;
    DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_flash_bwd: not built yet");
}
