// bf16 MFMA tile GEMM for gfx950 + the GEMM dispatcher of the C ABI.
//
//   C[M,N] = alpha * A.B + beta * C + bias[n],  fp32 accumulate on v_mfma_f32_16x16x32_bf16.
//
// Operand tiles come in two storage forms (this is what makes NT / NN / TN one kernel):
//   K-major : memory is [row][k] (k contiguous)  -> LDS image [128 rows][64 k], fragments by ds_read_b128
//   M-major : memory is [k][row] (row contiguous) -> LDS image [64 k][128 rows], fragments by
//             ds_read_b64_tr_b16 (hardware transpose read; lane map verified in profiles/r01_probe_*.txt)
//     NT (y = x W^T)   : A K-major, B K-major
//     NN (dx = dy W)   : A K-major, B M-major
//     TN (dW = dy^T x) : A M-major, B M-major
// Staging is global_load_lds (16 B per lane, LDS destination lane-linear), double-buffered over BK = 64;
// bank conflicts are removed by swizzling the per-lane SOURCE chunk and applying the same XOR on the read
// (cdna_hip_programming.md rule 21).  The MFMA is issued with the operands swapped (mfma(Bfrag, Afrag))
// so that each lane ends up with 4 CONSECUTIVE n of one row m: the epilogue stores 8-byte (bf16) or
// 16-byte (fp32) vectors.  Workgroup = 256 threads = 2x2 waves, wave tile 64x64 = 4x4 fragments.
// Workgroup ids are remapped so that each XCD (private L2) owns a contiguous band of output tiles.
#include "gemm_tile.h"
#include "gemm_args.h"
#include <stdlib.h>

int db1_gemm_strided_generic(const GemmStridedArgs& a, int dtA, int dtB, int dtC, int dtBias, int batch, hipStream_t st);

template <bool A_KMAJOR, bool B_KMAJOR, typename TC, typename TBIAS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_tile_kernel(GemmTileArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware remap: consecutive hardware block ids round-robin over 8 XCDs; give each XCD a contiguous span
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // column-major walk over tiles inside a band of 8 tile-rows keeps the B panel L2-resident
    const int band = 8;
    const int tiles_per_band = band * p.tiles_n;
    const int b0 = bid / tiles_per_band, rem = bid % tiles_per_band;
    const int band_rows = (p.tiles_m - b0 * band) < band ? (p.tiles_m - b0 * band) : band;
    const int tm = b0 * band + rem % band_rows, tn = rem / band_rows;
    const int z = blockIdx.y, z0 = z / p.batch1, z1 = z % p.batch1;
    const bf16_t* A = p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const bf16_t* B = p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    const int m0 = tm * TBM, n0 = tn * TBN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / TBK / p.ksplit;          // k-tiles of this split
    const int kt0 = (int)blockIdx.z * nt;          // first k-tile
    const int mv = p.M - m0 < TBM ? p.M - m0 : TBM, nv = p.N - n0 < TBN ? p.N - n0 : TBN;  // valid rows / columns of this tile
    stage_tile<A_KMAJOR, 4>(A, p.lda, m0, kt0 * TBK, smem, wave, lane, mv);
    stage_tile<B_KMAJOR, 4>(B, p.ldb, n0, kt0 * TBK, smem + TILE_BYTES, wave, lane, nv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nt; t++) {
        char* sa = smem + cur * 2 * TILE_BYTES;
        char* sb = sa + TILE_BYTES;
        if (t + 1 < nt) {
            char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile<A_KMAJOR, 4>(A, p.lda, m0, (kt0 + t + 1) * TBK, na, wave, lane, mv);
            stage_tile<B_KMAJOR, 4>(B, p.ldb, n0, (kt0 + t + 1) * TBK, na + TILE_BYTES, wave, lane, nv);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) af[i] = load_frag<A_KMAJOR>(sa, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; j++) bfr[j] = load_frag<B_KMAJOR>(sb, wn * 64 + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // swapped: D[n][m]
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    // epilogue.  swapped-operand layout: lane holds m = lane & 15, n = (lane >> 4) * 4 + r
    TC* C = (TC*)p.C + z0 * p.c_bs0 + z1 * p.c_bs1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (m >= p.M || n >= p.N) continue;  // tail tile
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = p.alpha * acc[i][j][r];
            if (p.bias && blockIdx.z == 0 && p.c_zs == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] += ldf((const TBIAS*)p.bias + n + r);
            }
            TC* c = C + (int64_t)m * p.ldc + n;
            if (sizeof(TC) == 4 && p.ksplit > 1 && p.c_zs != 0) {  // split-K through the workspace: this slice's partial sum, plain store
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(c) + (int64_t)blockIdx.z * p.c_zs) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (sizeof(TC) == 4) {   // (a split without a workspace does not exist: the dispatcher returns DB1_ERR_WORKSPACE_TOO_SMALL)
                if (p.beta != 0.f) {
                    float4 o = *reinterpret_cast<const float4*>(c);
                    v[0] += p.beta * o.x; v[1] += p.beta * o.y; v[2] += p.beta * o.z; v[3] += p.beta * o.w;
                }
                *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                if (p.beta != 0.f) {
                    uint2 o = *reinterpret_cast<const uint2*>(c);
                    v[0] += p.beta * __uint_as_float(o.x << 16); v[1] += p.beta * __uint_as_float(o.x & 0xffff0000u);
                    v[2] += p.beta * __uint_as_float(o.y << 16); v[3] += p.beta * __uint_as_float(o.y & 0xffff0000u);
                }
                uint2 o;
                o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(c) = o;
            }
        }
    }
}

// ---- deterministic split-K for outputs too small to fill the chip (o_net / ff2 weight gradients, the per-head dR of the
// relative-position attention): the contraction is cut into S slices that run as an extra batch dimension of the SAME tile
// kernels (operand base pointers advance by K/S per slice, no kernel change), each writing an fp32 partial into a workspace;
// splitk_reduce_kernel then adds the partials in a fixed order, so results do not depend on scheduling (no atomics).
template <typename TC, typename TBIAS>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, TC* __restrict__ C, const TBIAS* __restrict__ bias, int M, int N,
                                                            int S, int64_t ldc, int64_t c_bs0, float beta) {
    const int nq = N >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * nq) return;
    const int m = (int)(idx / nq), n = (int)(idx % nq) * 4;
    const int b0 = blockIdx.y;
    const float* w = ws + ((int64_t)b0 * S * M + m) * N + n;
    float4 acc = *reinterpret_cast<const float4*>(w);
    for (int z = 1; z < S; z++) {
        const float4 v = *reinterpret_cast<const float4*>(w + (int64_t)z * M * N);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (bias) { acc.x += ldf(bias + n); acc.y += ldf(bias + n + 1); acc.z += ldf(bias + n + 2); acc.w += ldf(bias + n + 3); }
    TC* c = C + (int64_t)b0 * c_bs0 + (int64_t)m * ldc + n;
    if (sizeof(TC) == 4) {
        if (beta != 0.f) {
            const float4 o = *reinterpret_cast<const float4*>(c);
            acc.x += beta * o.x; acc.y += beta * o.y; acc.z += beta * o.z; acc.w += beta * o.w;
        }
        *reinterpret_cast<float4*>(c) = acc;
    } else {
        if (beta != 0.f) {
            const uint2 o = *reinterpret_cast<const uint2*>(c);
            acc.x += beta * __uint_as_float(o.x << 16); acc.y += beta * __uint_as_float(o.x & 0xffff0000u);
            acc.z += beta * __uint_as_float(o.y << 16); acc.w += beta * __uint_as_float(o.y & 0xffff0000u);
        }
        uint2 o;
        o.x = (unsigned)f2bf(acc.x) | ((unsigned)f2bf(acc.y) << 16);
        o.y = (unsigned)f2bf(acc.z) | ((unsigned)f2bf(acc.w) << 16);
        *reinterpret_cast<uint2*>(c) = o;
    }
}
// storage form of an operand from its strides: returns 0 = K-major, 1 = M-major, -1 = neither
static int operand_form(int64_t row_stride, int64_t k_stride, int64_t* ld) {
    if (k_stride == 1 && row_stride >= 1) { *ld = row_stride; return 0; }
    if (row_stride == 1 && k_stride >= 1) { *ld = k_stride; return 1; }
    return -1;
}

static bool fast_ok(int M, int N, int K, int dtA, int dtB, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int64_t c_rs,
                    int64_t c_cs, int* fa, int* fb, int64_t* lda, int64_t* ldb) {
    if (dtA != DB1_BF16 || dtB != DB1_BF16) return false;
    if (K % TBK || M < 8 || N < 8 || (N % 8) || K <= 0) return false;  // M / N tails are handled by clamped loads + masked stores
    if (c_cs != 1 || (c_rs % 4)) return false;
    *fa = operand_form(a_rs, a_cs, lda);  // A: rows = m, k stride = a_cs
    *fb = operand_form(b_cs, b_rs, ldb);  // B: rows = n (stride b_cs), k stride = b_rs
    if (*fa < 0 || *fb < 0) return false;
    if (*fa == 1 && (M % 8)) return false;  // an M-major A tile is loaded in 8-row chunks
    if (*fa == 1 && *fb == 0) return false;  // "TT" never occurs on the path
    if ((*lda % 8) || (*ldb % 8)) return false;
    return true;
}

extern "C" int db1_gemm_would_use_fast(int M, int N, int K, int dtA, int dtB, int dtC, int64_t a_rs, int64_t a_cs, int64_t b_rs,
                                       int64_t b_cs, int64_t c_rs, int64_t c_cs) {
    int fa, fb;
    int64_t lda, ldb;
    (void)dtC;
    return fast_ok(M, N, K, dtA, dtB, a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, &fa, &fb, &lda, &ldb) ? 1 : 0;
}

template <bool AK, bool BK_>
static void launch_tile(const GemmTileArgs& t, int dtC, int dtBias, dim3 grid, hipStream_t st) {
    const size_t sm = 4 * TILE_BYTES;
    if (dtC == DB1_F32) {
        if (dtBias == DB1_BF16) gemm_bf16_tile_kernel<AK, BK_, float, bf16_t><<<grid, 256, sm, st>>>(t);
        else gemm_bf16_tile_kernel<AK, BK_, float, float><<<grid, 256, sm, st>>>(t);
    } else {
        if (dtBias == DB1_BF16) gemm_bf16_tile_kernel<AK, BK_, bf16_t, bf16_t><<<grid, 256, sm, st>>>(t);
        else gemm_bf16_tile_kernel<AK, BK_, bf16_t, float><<<grid, 256, sm, st>>>(t);
    }
}

static thread_local int g_tri_mode = 0, g_tri_period = 0;  // structural-zero hint of the current call (db1_gemm_strided_tri)
// test-only steering (include/db1_hip_test.h), thread-local: never set by product code
static thread_local int g_force_generic = 0;
static thread_local int g_tile_pref = 0;      // 0: measured heuristics; 128 / 256 / 512 / 1024: pin that tile kernel
extern "C" void db1_test_gemm_force_generic(int on) { g_force_generic = on; }
extern "C" void db1_test_gemm_tile_override(int tile) { g_tile_pref = tile; }
// A/B knobs (thread-local, test header): pin one tile kernel / disable the workspace split-K
struct GemmEnv { int tile, splitk; };
static GemmEnv gemm_env() { return GemmEnv{db1_knob(DB1_KNOB_GEMM_TILE, 0), db1_knob(DB1_KNOB_GEMM_SPLITK, 1)}; }

// ---- the dispatcher's decision, separated from the launch so that the workspace query, the kernel-choice query and the call agree
enum { GK_GENERIC = 0, GK_TILE128 = 1, GK_TILE256 = 2, GK_PP = 3, GK_PP32 = 4, GK_W4 = 5, GK_SKINNY = 6, GK_W4N = 7, GK_SPLITK = 16, GK_TAIL = 32 };
struct GemmPlan {
    int kind = GK_GENERIC;   // GK_* of the kernel that runs the contraction (for GK_TAIL: of the main part)
    int fa = 0, fb = 0;
    int64_t lda = 0, ldb = 0;
    int S = 0;               // GK_SPLITK: slices of k
    int m_tail = 0;          // GK_TAIL: rows of the second call
    int ksplit = 1;          // GK_TILE128 with atomic split-K (conv weight gradients)
};
struct GemmShape {
    int M, N, K, dtA, dtB, dtC, batch0, batch1;
    int64_t a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;
    float beta;
};
static GemmTileArgs tile_args(const GemmShape& g, const GemmPlan& pl) {
    GemmTileArgs t;
    t.A = nullptr; t.B = nullptr; t.C = nullptr; t.bias = nullptr;
    t.M = g.M; t.N = g.N; t.K = g.K; t.lda = pl.lda; t.ldb = pl.ldb; t.ldc = g.c_rs;
    t.batch1 = g.batch1; t.a_bs0 = g.a_bs0; t.a_bs1 = g.a_bs1; t.b_bs0 = g.b_bs0; t.b_bs1 = g.b_bs1; t.c_bs0 = g.c_bs0; t.c_bs1 = g.c_bs1;
    t.alpha = 1.f; t.beta = g.beta; t.tiles_m = (g.M + TBM - 1) / TBM; t.tiles_n = (g.N + TBN - 1) / TBN; t.ksplit = 1;
    t.tri_mode = g_tri_mode; t.tri_period = g_tri_period; t.tri_walk = db1_knob(DB1_KNOB_TRI_SPLIT, 1) != 0;
    t.split_n = 0; t.Cu = nullptr; t.Cv = nullptr; t.bias_u = nullptr; t.bias_v = nullptr; t.ld_uv = 0;
    if (t.tri_mode == 2 && (t.tri_period <= 0 || (t.tri_period % TBK) || (g.K % t.tri_period))) t.tri_mode = 0;
    return t;
}
static int64_t splitk_bytes(const GemmShape& g, int S, int M) { return (int64_t)g.batch0 * g.batch1 * S * M * g.N * (int64_t)sizeof(float); }

// `aligned`: the operand pointers are 16-byte aligned (the queries assume so); `ws_bytes`: workspace the caller offers (< 0: "whatever
// the plan needs", used by the queries).  A split-K plan is only made when its partial sums fit the offered workspace.
static GemmPlan gemm_plan(const GemmShape& g, bool aligned, int64_t ws_bytes) {
    GemmPlan pl;
    const int M = g.M, N = g.N, K = g.K;
    const int64_t batch = (int64_t)g.batch0 * g.batch1;
    aligned = aligned && !(g.a_bs0 % 8) && !(g.a_bs1 % 8) && !(g.b_bs0 % 8) && !(g.b_bs1 % 8) && !(g.c_bs0 % 4) && !(g.c_bs1 % 4);
    const int tile_pref = g_tile_pref ? g_tile_pref : gemm_env().tile;
    if (g_force_generic || !aligned) return pl;
    // few rows (inference with memory: 1 .. ~50 new tokens): stream W once, see gemm_skinny.hip
    if (batch == 1 && g.dtA == DB1_BF16 && g.dtB == DB1_BF16 && M <= 64 && g.a_cs == 1 && g.b_rs == 1 && g.c_cs == 1 && (N % 16) == 0 &&
        (K % 64) == 0 && (g.a_rs % 8) == 0 && (g.b_cs % 8) == 0 && (g.c_rs % 4) == 0 && tile_pref <= 0) {
        pl.kind = GK_SKINNY;
        return pl;
    }
    if (batch > 65535 || !fast_ok(M, N, K, g.dtA, g.dtB, g.a_rs, g.a_cs, g.b_rs, g.b_cs, g.c_rs, g.c_cs, &pl.fa, &pl.fb, &pl.lda, &pl.ldb)) return pl;
    const int fa = pl.fa, fb = pl.fb;
    const bool splitk_on = gemm_env().splitk != 0;
    // measured on MI355X at the DB1-1.3B shapes (tools/bench_kernels.py gemm; table in DESIGN.md): the 256x256 kernels win by 15-35 %
    // wherever they have >= ~160 output tiles to spread over the 256 CUs; below that the 3-stage 256x128 kernel wins for the
    // transposed-operand forms and the 2-stage 128x128 kernel for NT / small outputs.
    const bool pp_shape = (M % 256) == 0 && (N % 256) == 0, t256_shape = (M % 256) == 0 && (N % TBN) == 0;
    const GemmTileArgs t = tile_args(g, pl);
    auto big = [&](const GemmTileArgs& u, int nbatch, bool k32) {   // which 256x256 kernel db1_gemm_pp(32)_launch ends up running
        return db1_gemm_w4_supported(u, fa, fb, u.C == (void*)1 ? DB1_F32 : g.dtC, nbatch) ? GK_W4 : (k32 ? GK_PP32 : GK_PP);
    };
    // a short last wave of 256x256 tiles (head dW: 1040 tiles = 4 waves + 16 tiles, i.e. a fifth wave on 6 % of the CUs): the tile
    // rows of that remainder become a second call (9.3 -> 7.9 ms at T = 65 536)
    if (tile_pref == 0 && splitk_on && pp_shape && batch == 1 && g.c_cs == 1 && g_tri_mode == 0) {
        const int64_t tn_ = N / 256, wg_ = (int64_t)(M / 256) * tn_, rem = wg_ % 256;
        if (wg_ > 256 && rem > 0 && rem <= 48 && rem % tn_ == 0 && (K / TBK) >= 256) {
            GemmShape mainp = g;                      // (the second call plans itself: split-K or the 256x128 kernel, as its size says)
            mainp.M = M - (int)(rem / tn_) * 256;
            pl = gemm_plan(mainp, true, ws_bytes);
            pl.kind |= GK_TAIL;
            pl.m_tail = (int)(rem / tn_) * 256;
            return pl;
        }
    }
    // 256 x 128 tiles of the 4-wave kernel (gemm_w4.hip, NJ = 4): outputs whose 256 x 256 tiling fills at most half a wave of workgroups while
    // the 256 x 128 tiling fills a whole one -- micro-batches of 4 sequences: o_net, ff2, the data gradients (knob "w4n": 0 off; 1 before
    // the half-wave split-K; 2 after it; 3 = 1 + the last-wave rule below: the default)
    const int w4n_mode = db1_knob(DB1_KNOB_W4N, 3);
    const bool w4n_shape = w4n_mode && tile_pref == 0 && (M % 256) == 0 && (N % 128) == 0 && g.c_cs == 1 &&
                           db1_gemm_w4n_supported(t, fa, fb, g.dtC, (int)batch);   // (incl. its structural-zero rules)
    const int64_t wg256 = (int64_t)(M / 256) * ((N + 255) / 256) * batch, wg128 = (int64_t)(M / 256) * (N / 128) * batch;
    // (a weight gradient over K = 65 536 rows keeps the two-slice split-K of the 256 x 256 kernel: ff2 dW 831 us against 918 us here; at
    //  K = 16 384 the 256 x 128 kernel wins, 244 against 260 us)
    const bool w4n_half = w4n_shape && wg256 > 96 && wg256 <= 128 && wg128 >= 192 && !(fa == 1 && (K / TBK) >= 512);
    if (w4n_half && w4n_mode != 2) { pl.kind = GK_W4N; return pl; }
    // deterministic split-K (see splitk_reduce_kernel): only when the big-tile kernels would leave most CUs idle
    if (tile_pref == 0 && splitk_on && g.batch1 == 1 && (pp_shape || (t256_shape && fb == 1)) && g.c_cs == 1) {
        const int64_t wg = pp_shape ? (int64_t)(M / 256) * (N / 256) * batch : (int64_t)(M / 256) * (N / TBN) * batch;
        int S = 0;
        // structural-zero hint 2 on one column of tiles (the per-head dR, 64 tiles): the tile rows do 16 : 12 : 8 : 4 of the contraction, so
        // one slice set that fills the chip (4 slices: 256 workgroups) waits for its heaviest row; with 8 slices and the heavy-first walk of the
        // 4-wave kernel the light rows run behind the heavy ones (knob "tri_split": 0 = the plain rule)
        const bool tri_shape = !pp_shape && (int64_t)(N / TBN) == 1 && db1_knob(DB1_KNOB_TRI_SPLIT, 1) == 1;   // (2: the walk alone)
        const bool tri_rows = tri_shape && g_tri_mode == 2 && g_tri_period > 0;
        // (the workspace query does not know the hint: it prices the shape as if it came with one -- the larger slice count)
        const int64_t wg_cap = (tri_rows || (tri_shape && ws_bytes < 0)) ? 512 : 288;
        for (int cand = 8; cand >= 2; cand >>= 1)
            if (wg * cand <= wg_cap && (K / TBK) % cand == 0 && (K / TBK) / cand >= 16 && (!tri_rows || (K / cand) % g_tri_period == 0) &&
                (ws_bytes < 0 || splitk_bytes(g, cand, M) <= ws_bytes)) { S = cand; break; }   // (16: micro-batches of 4 sequences, K = 4096)
        // half a wave of 256x256 tiles (128 of them) in two slices.  Round 1 measured the weight gradients of 64-sequence batches (ff2 dW: nothing
        // at K = 16 384, 1184 -> 978 us at K = 65 536).  Round 4, micro-batches of 4 sequences (T = 4096, tools/bench_kernels.py gemm 4, 4-wave
        // kernels): the data gradients with long contractions gain most -- dqkv NN K = 6144: 112 -> 92 us, dff1 NN K = 8192: 163 -> 119 us --, ff2 dW
        // TN K = 4096: 91 -> 80 us, ff2 NT K = 4096: equal, and K = 2048 (o_net NT / NN) LOSES 15-20 % (16 k-tiles per slice do not pay for the
        // second prologue + the reduce).  Rule: >= 32 k-tiles per slice when B is M-major, >= 48 when both operands are K-major.
        const int hw_min = db1_knob(DB1_KNOB_GEMM_HALFWAVE, 0);   // A/B knob: > 0 = this many k-tiles per slice for every layout
        const int hw_need = hw_min > 0 ? hw_min : (fb == 1 ? 32 : 48);
        const bool half_wave = pp_shape && wg > 96 && wg <= 128 && S == 2 && (K / TBK) / S >= hw_need;
        // three quarters of a wave (qkv dW: 192 tiles) in four slices = three whole waves: 1310-1324 -> 1196-1221 us at K = 65 536 with
        // the 4-wave kernel (with the 8-wave kernels this gained 2 %)
        const bool three_quarters = pp_shape && fb == 1 && wg == 192 && (K / TBK) % 4 == 0 && (K / TBK) / 4 >= 128;
        if (three_quarters) S = 4;
        if (S && (wg <= 96 || half_wave || three_quarters) && (!pp_shape || wg * S >= 160) && (ws_bytes < 0 || splitk_bytes(g, S, M) <= ws_bytes)) {
            GemmTileArgs u = t;
            u.K = K / S; u.ldc = N; u.C = (void*)1;   // (marker: fp32 partials)
            if (u.tri_mode == 1 || (u.tri_mode == 2 && (u.K % u.tri_period))) u.tri_mode = 0;
            u.batch1 = S;
            pl.S = S;
            // (!pp_shape: N is a multiple of 128 only -- the per-head dR contraction: the 256 x 128 form of the 4-wave kernel where it applies)
            const bool n128 = !pp_shape && w4n_mode && tile_pref == 0 && db1_gemm_w4n_supported(u, fa, fb, DB1_F32, (int)batch * S);
            pl.kind = (pp_shape ? big(u, (int)batch * S, fb == 1) : (n128 ? GK_W4N : GK_TILE256)) | GK_SPLITK;
            return pl;
        }
    }
    if (w4n_half) { pl.kind = GK_W4N; return pl; }
    // ... and outputs whose LAST wave of 256 x 256 workgroups is mostly empty (qkv at 4 sequences: 384 tiles = 1.5 waves): by the fill of the
    // last wave, with the 256 x 128 kernel priced at 0.85 of the 256 x 256 one per FLOP (knob "w4n" >= 3; measured below)
    if (w4n_shape && w4n_mode >= 3 && wg256 >= 160 && wg256 < 1024) {
        const double e256 = (double)wg256 / (256.0 * ((wg256 + 255) / 256)), e128 = (double)wg128 / (256.0 * ((wg128 + 255) / 256));
        if (0.85 * e128 > e256) { pl.kind = GK_W4N; return pl; }
    }
    if (pp_shape && tile_pref == 1024) { pl.kind = big(t, (int)batch, true); return pl; }
    if (pp_shape && (tile_pref == 512 || (tile_pref == 0 && (int64_t)(M / 256) * (N / 256) * batch >= 160))) {
        // measured: the 4-stage k32 ring is 3-9 % faster when B is M-major (NN, TN); with both operands K-major
        // (NT) its 64-byte rows fetch half cache lines and the 2-stage k64 kernel is 4-9 % faster
        pl.kind = big(t, (int)batch, tile_pref == 0 && fb == 1);
        return pl;
    }
    // (measured: routing the under-filled transposed-operand cases -- o_net dW on 128 workgroups, the per-head dR on 64 --
    // to the 128x128 kernel for more workgroups made them 15-20 % slower, so form alone decides)
    // N a multiple of 128 but not of 256 with at least three quarters of a wave of 256 x 128 tiles: the 4-wave kernel's 256 x 128 form instead of the
    // 3-stage tile kernel (the per-head dR of an accumulation window, two batch levels, 1024 tiles: 404-409 -> 377-380 us with the heavy-first walk, profiles/r06q_dr_window_w4n.txt)
    // (not under hint 1: only the tile kernel skips those k-tiles -- half the product -- so that form stays where it is)
    if (w4n_shape && w4n_mode >= 3 && !pp_shape && wg128 >= 192 && g_tri_mode != 1) { pl.kind = GK_W4N; return pl; }
    const bool want256 = tile_pref == 256 || (tile_pref == 0 && fb == 1);
    if (want256 && t256_shape) { pl.kind = GK_TILE256; return pl; }
    // split-K for small outputs with a very long contraction (weight gradients of the 64-channel patch convolutions: 64 x 576 outputs
    // over 1.9 M rows would otherwise occupy 5 of 256 CUs): partial sums in the caller's workspace, added in slice order -- never
    // float atomics, so only shapes the workspace reduce handles (one batch, unit column stride, N % 4 == 0) are split
    pl.kind = GK_TILE128;
    const int ntiles = t.tiles_m * t.tiles_n * (int)batch;
    if (g.dtC == DB1_F32 && g.beta == 1.0f && ntiles < 128 && K >= 64 * TBK && batch == 1 && (N % 4) == 0 && g.c_cs == 1) {
        int ks = 512 / ntiles;
        while (ks > 1 && ((K / TBK) % ks || (K / TBK) / ks < 8)) ks--;
        pl.ksplit = ks;
    }
    return pl;
}

static GemmShape gemm_shape(int M, int N, int K, int dtA, int dtB, int dtC, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int64_t c_rs,
                            int64_t c_cs, int batch0, int batch1, int64_t a_bs0, int64_t a_bs1, int64_t b_bs0, int64_t b_bs1, int64_t c_bs0,
                            int64_t c_bs1, float beta) {
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.dtA = dtA; g.dtB = dtB; g.dtC = dtC; g.batch0 = batch0; g.batch1 = batch1;
    g.a_rs = a_rs; g.a_cs = a_cs; g.b_rs = b_rs; g.b_cs = b_cs; g.c_rs = c_rs; g.c_cs = c_cs;
    g.a_bs0 = a_bs0; g.a_bs1 = a_bs1; g.b_bs0 = b_bs0; g.b_bs1 = b_bs1; g.c_bs0 = c_bs0; g.c_bs1 = c_bs1; g.beta = beta;
    return g;
}

extern "C" int64_t db1_gemm_workspace_bytes(int M, int N, int K, int dtA, int dtB, int dtC, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs,
                                            int64_t c_rs, int64_t c_cs, int batch0, int batch1) {
    if (M <= 0 || N <= 0 || K <= 0 || batch0 <= 0 || batch1 <= 0) return 0;
    // (batch strides only enter through their alignment, which the model's operands satisfy; beta only matters for the 128-tile split of
    // small fp32 accumulators, which exists for beta = 1: asked for here, so that an accumulating call finds its partial-sum space)
    GemmShape g = gemm_shape(M, N, K, dtA, dtB, dtC, a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, batch0, batch1, 0, 0, 0, 0, 0, 0, 1.f);
    const GemmPlan pl = gemm_plan(g, true, -1);
    if (pl.kind & GK_TAIL) {
        GemmShape tail = g;
        tail.M = pl.m_tail;
        const GemmPlan tp = gemm_plan(tail, true, -1);
        return (tp.kind & GK_SPLITK) ? splitk_bytes(tail, tp.S, tail.M) : 0;
    }
    if (pl.ksplit > 1) return (int64_t)pl.ksplit * M * N * (int64_t)sizeof(float);   // 128-tile split-K: partial sums instead of atomics
    return (pl.kind & GK_SPLITK) ? splitk_bytes(g, pl.S, M) : 0;
}

extern "C" int db1_gemm_kernel_choice(int M, int N, int K, int dtA, int dtB, int dtC, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs,
                                      int64_t c_rs, int64_t c_cs, int batch0, int batch1, float beta, int64_t ws_bytes) {
    if (M <= 0 || N <= 0 || K <= 0 || batch0 <= 0 || batch1 <= 0) return -1;
    return gemm_plan(gemm_shape(M, N, K, dtA, dtB, dtC, a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, batch0, batch1, 0, 0, 0, 0, 0, 0, beta), true, ws_bytes).kind;
}

extern "C" int db1_gemm_strided(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int dtA, int dtB,
                                int dtC, int dtBias, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int64_t c_rs,
                                int64_t c_cs, int batch0, int batch1, int64_t a_bs0, int64_t a_bs1, int64_t b_bs0,
                                int64_t b_bs1, int64_t c_bs0, int64_t c_bs1, float alpha, float beta, void* ws, int64_t ws_bytes, void* stream) {
    if (!db1_dt_ok(dtA) || !db1_dt_ok(dtB) || !db1_dt_ok(dtC) || (bias && !db1_dt_ok(dtBias)))
        DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "gemm: dtype codes %d %d %d", dtA, dtB, dtC);
    if (M <= 0 || N <= 0 || K <= 0 || batch0 <= 0 || batch1 <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "gemm: M=%d N=%d K=%d batch=%dx%d", M, N, K, batch0, batch1);
    if (!A || !B || !C) DB1_FAIL(DB1_ERR_BAD_SHAPE, "gemm: null operand");
    hipStream_t st = (hipStream_t)stream;
    const int64_t batch = (int64_t)batch0 * batch1;
    const GemmShape g = gemm_shape(M, N, K, dtA, dtB, dtC, a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, batch0, batch1, a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1, beta);
    if (ws && (((uintptr_t)ws) & 15)) ws_bytes = 0;   // an unusable workspace is no workspace: the plan then avoids the split-K paths
    const GemmPlan pl = gemm_plan(g, db1_aligned16(A) && db1_aligned16(B) && db1_aligned16(C), ws ? ws_bytes : 0);
    const int base = pl.kind & 15, fa = pl.fa, fb = pl.fb;
    if (base == GK_SKINNY)
        return db1_gemm_skinny_launch((const bf16_t*)A, (const bf16_t*)B, C, bias, M, N, K, a_rs, b_cs, c_rs, alpha, beta, dtC, dtBias, st);
    if (base == GK_GENERIC) {
        GemmStridedArgs a;
        a.A = A; a.B = B; a.C = C; a.bias = bias; a.M = M; a.N = N; a.K = K;
        a.a_rs = a_rs; a.a_cs = a_cs; a.b_rs = b_rs; a.b_cs = b_cs; a.c_rs = c_rs; a.c_cs = c_cs;
        a.batch1 = batch1; a.a_bs0 = a_bs0; a.a_bs1 = a_bs1; a.b_bs0 = b_bs0; a.b_bs1 = b_bs1; a.c_bs0 = c_bs0; a.c_bs1 = c_bs1;
        a.alpha = alpha; a.beta = beta;
        return db1_gemm_strided_generic(a, dtA, dtB, dtC, dtBias, (int)batch, st);
    }
    if (pl.kind & GK_TAIL) {   // the last, short wave of tile rows as a second call (it takes the split-K path)
        const int m_main = M - pl.m_tail;
        const size_t esA = dtA == DB1_F32 ? 4 : 2, esC = dtC == DB1_F32 ? 4 : 2;
        int rc = db1_gemm_strided(A, B, C, bias, m_main, N, K, dtA, dtB, dtC, dtBias, a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, 1, 1, 0, 0, 0, 0, 0,
                                  0, alpha, beta, ws, ws_bytes, stream);
        if (rc) return rc;
        return db1_gemm_strided((const char*)A + (size_t)m_main * a_rs * esA, B, (char*)C + (size_t)m_main * c_rs * esC, bias, pl.m_tail, N, K, dtA,
                                dtB, dtC, dtBias, a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, 1, 1, 0, 0, 0, 0, 0, 0, alpha, beta, ws, ws_bytes, stream);
    }
    GemmTileArgs t = tile_args(g, pl);
    t.A = (const bf16_t*)A; t.B = (const bf16_t*)B; t.C = C; t.bias = bias; t.alpha = alpha;
    if (pl.kind & GK_SPLITK) {
        const int S = pl.S;
        float* wsf = (float*)ws;
        GemmTileArgs u = t;
        const int64_t kc = K / S;
        u.K = (int)kc; u.C = wsf; u.bias = nullptr; u.beta = 0.f; u.ldc = N;
        if (u.tri_mode == 1 || (u.tri_mode == 2 && (kc % u.tri_period))) u.tri_mode = 0;  // a slice of k no longer starts at k = 0 / on a period
        u.batch1 = S; u.a_bs1 = fa == 0 ? kc : kc * pl.lda; u.b_bs1 = fb == 0 ? kc : kc * pl.ldb;
        u.c_bs0 = (int64_t)S * M * N; u.c_bs1 = (int64_t)M * N;
        const bool pp_shape = (M % 256) == 0 && (N % 256) == 0;
        int rc = pp_shape ? (fb == 1 ? db1_gemm_pp32_launch(u, fa, fb, DB1_F32, DB1_F32, (int)batch * S, st)
                                     : db1_gemm_pp_launch(u, fa, fb, DB1_F32, DB1_F32, (int)batch * S, st))
                          : (base == GK_W4N ? db1_gemm_w4n_launch(u, fa, fb, DB1_F32, DB1_F32, (int)batch * S, st)
                                            : db1_gemm_tile256_launch(u, fa, fb, DB1_F32, DB1_F32, (int)batch * S, st));
        if (rc) return rc;
        dim3 rg((unsigned)(((int64_t)M * (N / 4) + 255) / 256), (unsigned)batch);
#define RED(TC, TB) splitk_reduce_kernel<TC, TB><<<rg, 256, 0, st>>>(wsf, (TC*)C, (const TB*)bias, M, N, S, c_rs, c_bs0, beta)
        if (dtC == DB1_F32) { if (dtBias == DB1_BF16) RED(float, bf16_t); else RED(float, float); }
        else { if (dtBias == DB1_BF16) RED(bf16_t, bf16_t); else RED(bf16_t, float); }
#undef RED
        DB1_CHECK_LAUNCH("splitk_reduce");
        return DB1_OK;
    }
    if (base == GK_W4 || base == GK_PP || base == GK_PP32) {
        const int tile_pref = g_tile_pref ? g_tile_pref : gemm_env().tile;
        const bool k32 = tile_pref == 1024 || (tile_pref == 0 && fb == 1);
        return k32 ? db1_gemm_pp32_launch(t, fa, fb, dtC, dtBias, (int)batch, st) : db1_gemm_pp_launch(t, fa, fb, dtC, dtBias, (int)batch, st);
    }
    if (base == GK_W4N) return db1_gemm_w4n_launch(t, fa, fb, dtC, dtBias, (int)batch, st);
    if (base == GK_TILE256) return db1_gemm_tile256_launch(t, fa, fb, dtC, dtBias, (int)batch, st);
    // long-contraction split of a small fp32 output (weight gradients of the patch convolutions): the slices' partial sums are stored in
    // the caller's workspace and added in a fixed order (bit-reproducible).  Without the workspace the product runs unsplit on the same
    // kernel (the header's contract for the GEMMs: slower, equally valid) -- never with float atomics.
    const bool ws_split = pl.ksplit > 1 && ws && db1_aligned16(ws) && ws_bytes >= (int64_t)pl.ksplit * M * N * (int64_t)sizeof(float);
    t.ksplit = ws_split ? pl.ksplit : 1;
    dim3 grid((unsigned)(t.tiles_m * t.tiles_n), (unsigned)batch, (unsigned)t.ksplit);
    void* C_final = C;
    if (ws_split) { t.C = ws; t.ldc = N; t.c_zs = (int64_t)M * N; t.bias = nullptr; t.beta = 0.f; }
    static Db1PerDeviceOnce attr_once;   // 64 KiB of dynamic LDS needs the opt-in attribute: once per device, every instantiation
    attr_once.run([] {
#define SET_ATTR(AK, BK_, TC, TB) hipFuncSetAttribute((const void*)gemm_bf16_tile_kernel<AK, BK_, TC, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES)
#define SET_ALL(AK, BK_) SET_ATTR(AK, BK_, float, float); SET_ATTR(AK, BK_, float, bf16_t); SET_ATTR(AK, BK_, bf16_t, float); SET_ATTR(AK, BK_, bf16_t, bf16_t)
        SET_ALL(true, true); SET_ALL(true, false); SET_ALL(false, false);
#undef SET_ALL
#undef SET_ATTR
    });
    if (fa == 0 && fb == 0) launch_tile<true, true>(t, dtC, dtBias, grid, st);
    else if (fa == 0 && fb == 1) launch_tile<true, false>(t, dtC, dtBias, grid, st);
    else launch_tile<false, false>(t, dtC, dtBias, grid, st);
    DB1_CHECK_LAUNCH("gemm_bf16_tile");
    if (ws_split) {
        dim3 rg((unsigned)(((int64_t)M * (N / 4) + 255) / 256), 1u);
        if (dtBias == DB1_BF16) splitk_reduce_kernel<float, bf16_t><<<rg, 256, 0, st>>>((const float*)ws, (float*)C_final, (const bf16_t*)bias, M, N, pl.ksplit, c_rs, 0, beta);
        else splitk_reduce_kernel<float, float><<<rg, 256, 0, st>>>((const float*)ws, (float*)C_final, (const float*)bias, M, N, pl.ksplit, c_rs, 0, beta);
        DB1_CHECK_LAUNCH("splitk_reduce (128-tile)");
    }
    return DB1_OK;
}

extern "C" int db1_gemm_strided_tri(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int dtA, int dtB,
                                    int dtC, int dtBias, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int64_t c_rs,
                                    int64_t c_cs, int batch0, int batch1, int64_t a_bs0, int64_t a_bs1, int64_t b_bs0,
                                    int64_t b_bs1, int64_t c_bs0, int64_t c_bs1, float alpha, float beta, int tri_mode, int tri_period,
                                    void* ws, int64_t ws_bytes, void* stream) {
    if (tri_mode < 0 || tri_mode > 2) DB1_FAIL(DB1_ERR_UNSUPPORTED, "gemm_strided_tri: mode %d", tri_mode);
    g_tri_mode = tri_mode;
    g_tri_period = tri_period;
    const int rc = db1_gemm_strided(A, B, C, bias, M, N, K, dtA, dtB, dtC, dtBias, a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, batch0, batch1, a_bs0,
                                    a_bs1, b_bs0, b_bs1, c_bs0, c_bs1, alpha, beta, ws, ws_bytes, stream);
    g_tri_mode = 0;
    g_tri_period = 0;
    return rc;
}

extern "C" int db1_gemm_nt(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                           int64_t ldc, int dtAB, int dtC, float alpha, float beta, void* ws, int64_t ws_bytes, void* stream) {
    return db1_gemm_strided(A, B, C, bias, M, N, K, dtAB, dtAB, dtC, dtAB, lda, 1, 1, ldb, ldc, 1, 1, 1, 0, 0, 0, 0, 0, 0, alpha, beta, ws, ws_bytes, stream);
}
// y = x W^T for the attention input projection with the two head biases folded into the epilogue (see GemmTileArgs::split_n)
extern "C" int db1_gemm_nt_headbias_supported(int M, int N, int K, int split_n) {
    return (M > 0 && (M % 256) == 0 && (N % 256) == 0 && (K % TBK) == 0 && split_n > 0 && (split_n % 256) == 0 && split_n < N &&
            (int64_t)(M / 256) * (N / 256) >= 160) ? 1 : 0;
}
extern "C" int db1_gemm_nt_headbias(const void* A, const void* W, void* C, void* Cu, void* Cv, const void* bias_u, const void* bias_v, int M, int N, int K,
                                    int split_n, int64_t lda, int64_t ldw, int64_t ldc, int64_t ld_uv, void* stream) {
    if (!db1_gemm_nt_headbias_supported(M, N, K, split_n)) DB1_FAIL(DB1_ERR_UNSUPPORTED, "gemm_nt_headbias: M=%d N=%d K=%d split=%d", M, N, K, split_n);
    if (!A || !W || !C || !Cu || !Cv || !bias_u || !bias_v) DB1_FAIL(DB1_ERR_BAD_SHAPE, "gemm_nt_headbias: null operand");
    if (!db1_aligned16(A) || !db1_aligned16(W) || !db1_aligned16(C) || !db1_aligned16(Cu) || !db1_aligned16(Cv) || (lda % 8) || (ldw % 8) || (ldc % 4) ||
        (ld_uv % 4) || lda < K || ldw < K || ldc < N || ld_uv < split_n)
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "gemm_nt_headbias: alignment / leading dimensions");
    GemmTileArgs t;
    t.A = (const bf16_t*)A; t.B = (const bf16_t*)W; t.C = C; t.bias = nullptr;
    t.M = M; t.N = N; t.K = K; t.lda = lda; t.ldb = ldw; t.ldc = ldc;
    t.batch1 = 1; t.a_bs0 = t.a_bs1 = t.b_bs0 = t.b_bs1 = t.c_bs0 = t.c_bs1 = 0;
    t.alpha = 1.f; t.beta = 0.f; t.tiles_m = M / 256; t.tiles_n = N / 256; t.ksplit = 1;
    t.tri_mode = 0; t.tri_period = 0;
    t.split_n = split_n; t.Cu = Cu; t.Cv = Cv; t.bias_u = bias_u; t.bias_v = bias_v; t.ld_uv = ld_uv;
    // the reference's micro-batch of 4 sequences: 384 tiles of 256 x 256 = 1.5 rounds of workgroups; 768 tiles of 256 x 128 fill three (the
    // dispatcher's rule for plain products, gemm_plan: the 256 x 128 form priced at 0.85 of the 256 x 256 one per FLOP)
    {
        const int64_t wg256 = (int64_t)(M / 256) * (N / 256), wg128 = (int64_t)(M / 256) * (N / 128);
        const double e256 = (double)wg256 / (256.0 * ((wg256 + 255) / 256)), e128 = (double)wg128 / (256.0 * ((wg128 + 255) / 256));
        const int w4n_mode = db1_knob(DB1_KNOB_W4N, 3);
        if (w4n_mode >= 3 && w4n_mode != 5 && wg256 < 1024 && 0.85 * e128 > e256 && db1_gemm_w4n_supported(t, 0, 0, DB1_BF16, 1)) {   // (knob value 5: everything but this routing, for the A/B)
            t.tiles_n = N / 128;
            return db1_gemm_w4n_launch(t, 0, 0, DB1_BF16, DB1_BF16, 1, (hipStream_t)stream);
        }
    }
    return db1_gemm_pp_launch(t, 0, 0, DB1_BF16, DB1_BF16, 1, (hipStream_t)stream);
}
extern "C" int db1_gemm_nn(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                           int64_t ldc, int dtAB, int dtC, float alpha, float beta, void* ws, int64_t ws_bytes, void* stream) {
    return db1_gemm_strided(A, B, C, bias, M, N, K, dtAB, dtAB, dtC, dtAB, lda, 1, ldb, 1, ldc, 1, 1, 1, 0, 0, 0, 0, 0, 0, alpha, beta, ws, ws_bytes, stream);
}
extern "C" int db1_gemm_tn(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                           int64_t ldc, int dtAB, int dtC, float alpha, float beta, void* ws, int64_t ws_bytes, void* stream) {
    return db1_gemm_strided(A, B, C, bias, M, N, K, dtAB, dtAB, dtC, dtAB, 1, lda, ldb, 1, ldc, 1, 1, 1, 0, 0, 0, 0, 0, 0, alpha, beta, ws, ws_bytes, stream);
}
