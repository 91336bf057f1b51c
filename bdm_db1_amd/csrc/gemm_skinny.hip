// Skinny bf16 GEMM for inference with Transformer-XL memory (evaluate_rl.py:157-266: 1 .. ~50 new tokens per call):
//     y[M, N] = alpha * x[M, K] . W[N, K]^T + beta * y + bias[n],     M <= 64, both operands K-major (the NT form of nn.Linear).
// With so few rows the product is a stream over W (HBM-bound: 2 N K bytes): the 128x128 tile kernels would run N / 128
// workgroups with an almost empty tile each, the generic strided kernel took 135-250 us per projection at M = 1.
// One workgroup = 16 output columns; its 4 waves split the contraction 4 ways (partials added in wave order through LDS:
// deterministic).  Lane (r = lane & 15, g = lane >> 4) reads 32 contiguous bytes of W row n0 + r per 64-wide k-step (the four
// lanes of a row cover one 128-byte line) and the same 32 bytes of each x row; the two 8-element halves are the two
// v_mfma_f32_16x16x32_bf16 k-steps (any k permutation is fine as long as x and W use the same one).  Operands are swapped in
// the MFMA like in the tile kernels, so a lane ends with 4 consecutive n of row m = lane & 15 and reuses store_frag.
#include "gemm_tile.h"
#include "ln_row.h"

struct GemmSkinnyArgs {
    const bf16_t* x; const bf16_t* w; void* y; const void* bias;
    int M, N, K;
    int64_t ldx, ldw, ldy;
    float alpha, beta;
};

template <typename TC, typename TBIAS, int MT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmSkinnyArgs p) {
    __shared__ f32x4 red[3][MT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int ksteps = p.K / 64, per = (ksteps + 3) / 4;
    const int ks0 = wave * per, ks1 = ks0 + per < ksteps ? ks0 + per : ksteps;
    const bf16_t* wrow = p.w + (int64_t)(n0 + r) * p.ldw + g * 16;
    const bf16_t* xrow[MT];
    bool xok[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        xok[t] = t * 16 + r < p.M;
        xrow[t] = p.x + (int64_t)(xok[t] ? t * 16 + r : 0) * p.ldx + g * 16;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
    // The kernel is a latency problem (a workgroup streams only 64-256 KB of W): the W pieces of EIGHT k-steps (16 loads of 16 bytes per
    // lane, 32 KB per workgroup) are requested before the first one is used, so a wave's share of K = 2048 is ONE round trip instead of
    // two (unroll 4) -- and the loads are non-temporal: every byte of W is read once per call, by one workgroup.
    constexpr int KB = MT == 1 ? 8 : (MT == 2 ? 4 : 2);
    // (every load is unconditional, at a clamped address: a load inside a divergent branch is waited for before the next one is issued.
    //  x rows beyond M repeat row 0 -- row m of x only reaches output row m, which is never stored; k-steps beyond the wave's share
    //  re-read its last one against a zero W piece)
    for (int kb = ks0; kb < ks1; kb += KB) {
        bf16x8_t b0[KB], b1[KB], a0[MT][KB], a1[MT][KB];
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const int ks = kb + u < ks1 ? kb + u : ks1 - 1;
            b0[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + ks * 64));
            b1[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + ks * 64 + 8));
        }
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const int ks = kb + u < ks1 ? kb + u : ks1 - 1;
#pragma unroll
            for (int t = 0; t < MT; t++) {
                a0[t][u] = *reinterpret_cast<const bf16x8_t*>(xrow[t] + ks * 64);
                a1[t][u] = *reinterpret_cast<const bf16x8_t*>(xrow[t] + ks * 64 + 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // all requests first (the scheduler otherwise pairs loads with their MFMAs: 4 in flight)
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const bool ok = kb + u < ks1;
            const bf16x8_t w0 = ok ? b0[u] : zero, w1 = ok ? b1[u] : zero;
#pragma unroll
            for (int t = 0; t < MT; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a0[t][u], acc[t], 0, 0, 0);  // swapped: D[n][m]
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a1[t][u], acc[t], 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; t++) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave == 0) {
        TC* Y = (TC*)p.y;
#pragma unroll
        for (int t = 0; t < MT; t++) {
            f32x4 s = acc[t];
#pragma unroll
            for (int w = 0; w < 3; w++) {
                const f32x4 v = red[w][t][lane];
                s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
            }
            const int m = t * 16 + r;
            if (m < p.M) store_frag<TC, TBIAS>(s, Y, p.ldy, m, n0 + g * 4, p.alpha, p.beta, p.bias);
        }
    }
}

template <int MT>
static void launch_skinny(const GemmSkinnyArgs& a, int dtC, int dtBias, hipStream_t st) {
    const dim3 grid((unsigned)(a.N / 16));
    if (dtC == DB1_F32) {
        if (dtBias == DB1_BF16) gemm_skinny_kernel<float, bf16_t, MT><<<grid, 256, 0, st>>>(a);
        else gemm_skinny_kernel<float, float, MT><<<grid, 256, 0, st>>>(a);
    } else {
        if (dtBias == DB1_BF16) gemm_skinny_kernel<bf16_t, bf16_t, MT><<<grid, 256, 0, st>>>(a);
        else gemm_skinny_kernel<bf16_t, float, MT><<<grid, 256, 0, st>>>(a);
    }
}

// caller has checked: bf16 operands, both K-major, M <= 64, N % 16 == 0, K % 64 == 0, 16-byte alignment, ld % 8 == 0
int db1_gemm_skinny_launch(const bf16_t* x, const bf16_t* w, void* y, const void* bias, int M, int N, int K, int64_t ldx, int64_t ldw,
                           int64_t ldy, float alpha, float beta, int dtC, int dtBias, hipStream_t st) {
    GemmSkinnyArgs a;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.alpha = alpha; a.beta = beta;
    if (M <= 16) launch_skinny<1>(a, dtC, dtBias, st);
    else if (M <= 32) launch_skinny<2>(a, dtC, dtBias, st);
    else if (M <= 48) launch_skinny<3>(a, dtC, dtBias, st);
    else launch_skinny<4>(a, dtC, dtBias, st);
    DB1_CHECK_LAUNCH("gemm_skinny");
    return DB1_OK;
}


// ======================================================================================= the inference layer's linear maps, fused
// One token per call leaves a layer as nine dependent launches of 4-10 us each, most of it launch + memory latency: the feed-forward
// activation and the two residual LayerNorms were separate launches over a few KB.  Here the linear map finishes the job itself:
//   * GEGLU: a workgroup owns 8 output columns and multiplies the 8 "value" rows n and the 8 "gate" rows N + n of W in ONE 16-row MFMA
//     tile; the epilogue rounds both halves to bf16 (z is a bf16 tensor in the reference), y = a * gelu(gate) (activations.py:19-32);
//   * split-K (gridDim.y = S): the S partial tiles of a column group go to the workspace, the LAST workgroup to arrive (ticket counter) adds
//     them in split order -- deterministic, whichever workgroup it is;
//   * LayerNorm tail: the last column group to finish (second ticket) normalises the M rows: ln_out = LN(alpha * res + y) * gamma + beta with
//     the row code of the LayerNorm kernel (ln_row.h) on the bf16 y it re-reads, so the result equals the unfused pair of launches.
// The ticket counters are zero on entry and are left zero (see last_arrival below for the hand-off protocol).
struct SkinnyFusedArgs {
    const bf16_t* x; const bf16_t* w; bf16_t* y; const void* bias;
    int M, N, K, S;
    int64_t ldx, ldw, ldy;
    float* part; unsigned* tickets;
    const bf16_t* res; const void* gamma; const void* beta; bf16_t* ln_out;
    int64_t ld_res, ld_out;
    float alpha, eps;
    // LayerNorm of the INPUT rows: x_eff = LN(pre_alpha * pre_res + x) * pre_gamma + pre_beta (also stored to pre_out by workgroup 0)
    const bf16_t* pre_res; const void* pre_gamma; const void* pre_beta; bf16_t* pre_out;
    int64_t ld_pre_res, ld_pre_out;
    float pre_alpha, pre_eps;
    // the input rows are the MERGE of the decode attention's per-chunk partial results (PRO == -1): part [B*H][nunit][64][D + 2], row m = b * q + i
    const float* att_part; int att_nunit, att_q, att_H;
};

// Hand-off between workgroups inside a launch (per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed): the data goes out
// as write-through stores and comes back through L1-bypassing loads (relaxed agent-scope atomics = the sc1 forms, 8 bytes each), every wave
// drains its stores (s_waitcnt vmcnt(0)) before the workgroup's ticket.  No cache-wide write-back / invalidate: __threadfence() in every
// workgroup made the layer 1.8x SLOWER than the separate launches it replaced.
__device__ __forceinline__ void st_agent(void* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint2 ld_agent(const void* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
__device__ __forceinline__ void st_agent_f4(f32x4* p, const f32x4& v) {
    st_agent(p, make_uint2(__float_as_uint(v[0]), __float_as_uint(v[1])));
    st_agent(reinterpret_cast<char*>(p) + 8, make_uint2(__float_as_uint(v[2]), __float_as_uint(v[3])));
}
__device__ __forceinline__ f32x4 ld_agent_f4(const f32x4* p) {
    const uint2 a = ld_agent(p), b = ld_agent(reinterpret_cast<const char*>(p) + 8);
    return (f32x4){__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(b.x), __uint_as_float(b.y)};
}
// every wave's stores are out, then ONE ticket per workgroup; returns true in the workgroup that drew the last one (and resets the counter)
__device__ __forceinline__ bool last_arrival(unsigned* counter, unsigned expected, unsigned* lds_slot) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *lds_slot = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = *lds_slot == expected - 1;
    if (last && threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return last;
}

template <typename TP, int NV>
__device__ __forceinline__ void skinny_ln_rows(const SkinnyFusedArgs& p, int wave, int lane) {
    constexpr int V = 8;
    for (int m = wave; m < p.M; m += 4) {
        Vec16<bf16_t> a[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) a[k].load(p.res + m * p.ld_res + (k * 64 + lane) * V);
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const bf16_t* yp = p.y + m * p.ldy + (k * 64 + lane) * V;   // written by other workgroups of this launch
            const uint2 lo = ld_agent(yp), hi = ld_agent(yp + 4);
            const unsigned w[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                a[k].v[2 * j] = p.alpha * a[k].v[2 * j] + __uint_as_float(w[j] << 16);
                a[k].v[2 * j + 1] = p.alpha * a[k].v[2 * j + 1] + __uint_as_float(w[j] & 0xffff0000u);
            }
        }
        float mu, rs;
        ln_row_stats<bf16_t, NV>(a, p.N, p.eps, mu, rs);
        ln_row_store<bf16_t, TP, NV>(a, mu, rs, (const TP*)p.gamma, (const TP*)p.beta, p.ln_out + m * p.ld_out, lane);
    }
}

template <typename TBIAS, typename TP, int MT, bool GEGLU, int PRO>
__global__ __launch_bounds__(256) void gemm_skinny_fused_kernel(SkinnyFusedArgs p) {
    __shared__ f32x4 red[3][MT][64];
    __shared__ unsigned ticket_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int nb = blockIdx.x, sp = blockIdx.y;
    const int wr = GEGLU ? (r < 8 ? nb * 8 + r : p.N + nb * 8 + r - 8) : nb * 16 + r;       // this lane's row of W
    const int kps = p.K / 64 / p.S, per = (kps + 3) / 4;
    const int ks0 = sp * kps + wave * per, ks1 = ks0 + per < (sp + 1) * kps ? ks0 + per : (sp + 1) * kps;
    const bf16_t* wrow = p.w + (int64_t)wr * p.ldw + g * 16;
    const bf16_t* xrow[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) xrow[t] = p.x + (int64_t)(t * 16 + r < p.M ? t * 16 + r : 0) * p.ldx + g * 16;
    // bias of this lane's four outputs, requested with the first operands (the epilogue would wait a round trip for it)
    // (unconditional, from a valid address when there is no bias: a load inside a branch is waited for on the spot -- a round trip up front)
    float bv[4];
    {
        const int bn = GEGLU ? (g < 2 ? nb * 8 + 4 * g : p.N + nb * 8 + 4 * (g - 2)) : nb * 16 + 4 * g;
        const TBIAS* bp = p.bias ? (const TBIAS*)p.bias + bn : (const TBIAS*)p.w;
#pragma unroll
        for (int j = 0; j < 4; j++) { const float t = ldf(bp + j); bv[j] = p.bias ? t : 0.f; }
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int KB = MT == 1 ? 8 : (MT == 2 ? 4 : 2);
    if constexpr (PRO < 0) {
        // The input rows (M <= 2) are the merge of the decode attention's chunk partials
        // (relattn_decode.hip: O, max, sum per 128-key chunk; rows m = b * q + i): thread t owns columns 8 t .. 8 t + 7 = head t / 16, and weights the <= 12 units
        // in chunk order -- the arithmetic of relattn_decode_merge2_kernel, in the shadow of the W stream instead of at the end of the
        // attention launch behind a ticket (drain + atomic + two dependent reads: ~3 us per layer).
        static_assert(MT == 1, "attention-merge prologue: one 16-row tile");
        constexpr int MR = -PRO, XLD = 2048 + 8, NU = 12, DD = 128;
        __shared__ __attribute__((aligned(16))) bf16_t xs[MR][XLD];
        const int tcol = (int)threadIdx.x * 8;
        const bool tin = tcol < p.K;
        const int hh = tin ? tcol / DD : 0, d0 = tin ? tcol % DD : 0;
        float4 olo[MR][NU], ohi[MR][NU];
        float2 ml[MR][NU];
#pragma unroll
        for (int m = 0; m < MR; m++) {
            const int bq = m < p.M ? m : 0, bb = bq / p.att_q, qi = bq % p.att_q;
            const float* src = p.att_part + ((((int64_t)bb * p.att_H + hh) * p.att_nunit) * 64 + qi) * (DD + 2);
#pragma unroll
            for (int c = 0; c < NU; c++) {
                const float* u = src + (int64_t)(c < p.att_nunit ? c : p.att_nunit - 1) * 64 * (DD + 2);
                ml[m][c] = *reinterpret_cast<const float2*>(u + DD);
                const float2 a0 = *reinterpret_cast<const float2*>(u + d0), a1 = *reinterpret_cast<const float2*>(u + d0 + 2);
                const float2 a2 = *reinterpret_cast<const float2*>(u + d0 + 4), a3 = *reinterpret_cast<const float2*>(u + d0 + 6);
                olo[m][c] = make_float4(a0.x, a0.y, a1.x, a1.y); ohi[m][c] = make_float4(a2.x, a2.y, a3.x, a3.y);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8_t b0[KB], b1[KB];
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const int ks = ks0 + u < ks1 ? ks0 + u : ks1 - 1;
            b0[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + ks * 64));
            b1[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + ks * 64 + 8));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MR; m++) {
            float mx = -1.0e30f;
#pragma unroll
            for (int c = 0; c < NU; c++) mx = fmaxf(mx, c < p.att_nunit ? ml[m][c].x : -1.0e30f);
            float l = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NU; c++) {
                const float lc = c < p.att_nunit ? ml[m][c].y : 0.f;
                const float wt = lc > 0.f ? __expf(ml[m][c].x - mx) : 0.f;
                l += lc * wt;
                o[0] += olo[m][c].x * wt; o[1] += olo[m][c].y * wt; o[2] += olo[m][c].z * wt; o[3] += olo[m][c].w * wt;
                o[4] += ohi[m][c].x * wt; o[5] += ohi[m][c].y * wt; o[6] += ohi[m][c].z * wt; o[7] += ohi[m][c].w * wt;
            }
            Vec16<bf16_t> ov;
#pragma unroll
            for (int j = 0; j < 8; j++) ov.v[j] = l > 0.f ? o[j] / l : 0.f;
            if (tin) ov.store(&xs[m][tcol]);
        }
        __syncthreads();
        const bf16_t* xl = &xs[r < p.M && r < MR ? r : 0][g * 16];
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const bool ok = ks0 + u < ks1;
            const int ks = ok ? ks0 + u : ks1 - 1;
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(xl + ks * 64), a1 = *reinterpret_cast<const bf16x8_t*>(xl + ks * 64 + 8);
            const bf16x8_t w0 = ok ? b0[u] : zero, w1 = ok ? b1[u] : zero;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a0, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a1, acc[0], 0, 0, 0);
        }
    } else
    if constexpr (PRO > 0) {
        // The input rows are normalised on the way in (<= 16 rows, S == 1, the wave's share of K is one batch of <= KB k-steps).  The
        // workgroup does it cooperatively -- thread t owns columns 8 t .. 8 t + 7 of every row -- and leaves the bf16 rows in LDS, from where
        // the lanes read their MFMA pieces: the arithmetic is M K / 256 elements per thread (normalising in the MFMA layout made every wave
        // redo all 16 rows' share: 3.5 us of VALU per wave, +5 us per launch).  Two passes like the LayerNorm kernel (mean, then centred
        // squares) over s = alpha * res + x rounded to bf16 like the reference's tensor; the row sums are added lanes -> waves in a fixed order.
        // Order of the requests: loads return in order, so the rows and parameters (L2) go first and the W pieces (HBM) last -- the
        // normalisation runs while W is in flight.
        static_assert(MT == 1, "the LayerNorm prologue handles up to 16 rows");
        constexpr int MR = PRO, XLD = 2048 + 8;
        __shared__ __attribute__((aligned(16))) bf16_t xs[MR][XLD];
        __shared__ float rowpart[2][4][MR];
        const int tcol = (int)threadIdx.x * 8;
        const bool tin = tcol < p.K;
        const int col = tin ? tcol : 0;
        Vec16<bf16_t> xv[MR], gv, bvv;
        {
            Vec16<bf16_t> rv[MR];
#pragma unroll
            for (int m = 0; m < MR; m++) {
                const int row = m < p.M ? m : 0;
                xv[m].load(p.x + (int64_t)row * p.ldx + col);
                rv[m].load(p.pre_res + (int64_t)row * p.ld_pre_res + col);
            }
            if (sizeof(TP) == 2) { gv.load((const bf16_t*)p.pre_gamma + col); bvv.load((const bf16_t*)p.pre_beta + col); }
            else {
#pragma unroll
                for (int j = 0; j < 8; j++) { gv.v[j] = ldf((const TP*)p.pre_gamma + col + j); bvv.v[j] = ldf((const TP*)p.pre_beta + col + j); }
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8_t b0[KB], b1[KB];
#pragma unroll
            for (int u = 0; u < KB; u++) {
                const int ks = ks0 + u < ks1 ? ks0 + u : ks1 - 1;
                b0[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + ks * 64));
                b1[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + ks * 64 + 8));
            }
            __builtin_amdgcn_sched_barrier(0);
            float part[MR];
#pragma unroll
            for (int m = 0; m < MR; m++) {
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    xv[m].v[j] = bf2f(f2bf(p.pre_alpha * rv[m].v[j] + xv[m].v[j]));
                    sum += xv[m].v[j];
                }
                part[m] = wave_sum(tin ? sum : 0.f);
            }
            if (lane == 0) {
#pragma unroll
                for (int m = 0; m < MR; m++) rowpart[0][wave][m] = part[m];
            }
            __syncthreads();
            float mu[MR];
#pragma unroll
            for (int m = 0; m < MR; m++) {
                mu[m] = (rowpart[0][0][m] + rowpart[0][1][m] + rowpart[0][2][m] + rowpart[0][3][m]) / (float)p.K;
                float sq = 0.f;
#pragma unroll
                for (int j = 0; j < 8; j++) { const float c = xv[m].v[j] - mu[m]; sq += c * c; }
                part[m] = wave_sum(tin ? sq : 0.f);
            }
            if (lane == 0) {
#pragma unroll
                for (int m = 0; m < MR; m++) rowpart[1][wave][m] = part[m];
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MR; m++) {
                const float rs = rsqrtf((rowpart[1][0][m] + rowpart[1][1][m] + rowpart[1][2][m] + rowpart[1][3][m]) / (float)p.K + p.pre_eps);
                Vec16<bf16_t> o;
#pragma unroll
                for (int j = 0; j < 8; j++) o.v[j] = (xv[m].v[j] - mu[m]) * rs * gv.v[j] + bvv.v[j];
                if (tin) {
                    o.store(&xs[m][col]);
                    if (nb == 0 && m < p.M) o.store(p.pre_out + (int64_t)m * p.ld_pre_out + col);
                }
            }
            __syncthreads();
            const bf16_t* xl = &xs[r < p.M && r < MR ? r : 0][g * 16];
#pragma unroll
            for (int u = 0; u < KB; u++) {
                const bool ok = ks0 + u < ks1;
                const int ks = ok ? ks0 + u : ks1 - 1;
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(xl + ks * 64), a1 = *reinterpret_cast<const bf16x8_t*>(xl + ks * 64 + 8);
                const bf16x8_t w0 = ok ? b0[u] : zero, w1 = ok ? b1[u] : zero;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a0, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a1, acc[0], 0, 0, 0);
            }
        }
    } else
    for (int kb = ks0; kb < ks1; kb += KB) {
        bf16x8_t b0[KB], b1[KB], a0[MT][KB], a1[MT][KB];
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const int ks = kb + u < ks1 ? kb + u : ks1 - 1;
            b0[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + ks * 64));
            b1[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wrow + ks * 64 + 8));
        }
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const int ks = kb + u < ks1 ? kb + u : ks1 - 1;
#pragma unroll
            for (int t = 0; t < MT; t++) {
                a0[t][u] = *reinterpret_cast<const bf16x8_t*>(xrow[t] + ks * 64);
                a1[t][u] = *reinterpret_cast<const bf16x8_t*>(xrow[t] + ks * 64 + 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KB; u++) {
            const bool ok = kb + u < ks1;
            const bf16x8_t w0 = ok ? b0[u] : zero, w1 = ok ? b1[u] : zero;
#pragma unroll
            for (int t = 0; t < MT; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a0[t][u], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a1[t][u], acc[t], 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; t++) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int w = 0; w < 3; w++) {
                const f32x4 v = red[w][t][lane];
                acc[t][0] += v[0]; acc[t][1] += v[1]; acc[t][2] += v[2]; acc[t][3] += v[3];
            }
    }
    if (p.S > 1) {   // (uniform over the launch)
        f32x4* mine = reinterpret_cast<f32x4*>(p.part) + ((int64_t)nb * p.S + sp) * MT * 64;
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < MT; t++) st_agent_f4(mine + t * 64 + lane, acc[t]);
        }
        if (!last_arrival(p.tickets + nb, (unsigned)p.S, &ticket_s)) return;
        if (wave == 0) {
            const f32x4* all = reinterpret_cast<const f32x4*>(p.part) + (int64_t)nb * p.S * MT * 64;
#pragma unroll
            for (int t = 0; t < MT; t++) {
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int q = 0; q < p.S; q++) {
                    const f32x4 v = ld_agent_f4(all + (q * MT + t) * 64 + lane);
                    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
                }
                acc[t] = s;
            }
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int m = t * 16 + r;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = acc[t][j] + bv[j];
            if (GEGLU) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float z = bf2f(f2bf(v[j]));                      // z is a bf16 tensor in the reference
                    const float gate = __shfl_xor(z, 32, 64);               // lanes g >= 2 hold the gate columns of lanes g - 2
                    o[j] = z * gelu_fwd_t<bf16_t>(gate);
                }
                if (g < 2 && m < p.M) {
                    uint2 w;
                    w.x = f2bf_pk(o[0], o[1]); w.y = f2bf_pk(o[2], o[3]);
                    bf16_t* dst = p.y + (int64_t)m * p.ldy + nb * 8 + 4 * g;
                    if (p.ln_out) st_agent(dst, w); else *reinterpret_cast<uint2*>(dst) = w;
                }
            } else if (m < p.M) {
                uint2 w;
                w.x = f2bf_pk(v[0], v[1]); w.y = f2bf_pk(v[2], v[3]);
                bf16_t* dst = p.y + (int64_t)m * p.ldy + nb * 16 + 4 * g;
                if (p.ln_out) st_agent(dst, w); else *reinterpret_cast<uint2*>(dst) = w;
            }
        }
    }
    if (!p.ln_out) return;
    if (!last_arrival(p.tickets + gridDim.x, gridDim.x, &ticket_s)) return;
    if (p.N == 2048) skinny_ln_rows<TP, 4>(p, wave, lane);
    else if (p.N == 1024) skinny_ln_rows<TP, 2>(p, wave, lane);
    else skinny_ln_rows<TP, 1>(p, wave, lane);
}

#define DB1_SKINNY_TICKETS 2048   // ticket words the caller provides (zero before the first launch; every launch leaves them zero)
extern "C" int64_t db1_linear_decode_tickets_bytes(void) { return (int64_t)DB1_SKINNY_TICKETS * 4; }
static int skinny_fused_split(int M, int N, int K, int geglu) {
    (void)M;
    const int mode = db1_knob(DB1_KNOB_LINEAR_DECODE_SPLITK, -1);   // A/B knob; 0 / 1: never split
    if (geglu || mode == 0 || mode == 1) return 1;
    const int groups = N / 16, ksteps = K / 64;
    int S = 1;
    while (groups * S < 256 && ksteps % (2 * S) == 0 && ksteps / (2 * S) >= 32) S *= 2;   // the partial-tile hand-off costs ~2 us: K >= 4096 only
    return S;
}
extern "C" int64_t db1_linear_decode_workspace_bytes(int M, int N, int K, int geglu) {
    const int S = skinny_fused_split(M, N, K, geglu);
    if (S == 1) return 0;
    return (int64_t)(N / 16) * S * ((M + 15) / 16) * 64 * (int64_t)sizeof(f32x4);
}
extern "C" int db1_linear_decode_supported(int M, int N, int K, int geglu, int ln) {   // ln: 0 none, 1 LayerNorm of the output rows, 2 of the input rows
    if (M < 1 || M > 64 || K % 64 || N % 16 || N < 16) return 0;
    if (((ln & 1) || (!(ln & 2) && skinny_fused_split(M, N, K, geglu) > 1)) && N / (geglu ? 8 : 16) + 1 > DB1_SKINNY_TICKETS) return 0;   // (tickets in use)
    if ((ln & 1) && N != 2048 && N != 1024 && N != 512) return 0;
    if ((ln & 2) && (M > 16 || K % 256 || K > 2048)) return 0;
    return 1;
}
extern "C" int db1_linear_decode(const void* x, int64_t ldx, const void* w, const void* bias, int dtBias, void* y, int64_t ldy, int M, int N, int K,
                                 int geglu, const void* pre_res, int64_t ld_pre_res, float pre_alpha, const void* pre_gamma, const void* pre_beta,
                                 float pre_eps, void* pre_out, int64_t ld_pre_out, const void* res, int64_t ld_res, float alpha, const void* gamma,
                                 const void* beta, float eps, void* ln_out, int64_t ld_out, int dtParam, void* tickets, void* ws, int64_t ws_bytes,
                                 void* stream) {
    const bool pro = pre_out != nullptr, tail = ln_out != nullptr;
    if (!db1_linear_decode_supported(M, N, K, geglu, (tail ? 1 : 0) | (pro ? 2 : 0)))
        DB1_FAIL(DB1_ERR_UNSUPPORTED, "linear_decode: M=%d N=%d K=%d geglu=%d ln_out=%d ln_in=%d (needs M <= 64, K %% 64 == 0, N %% 16 == 0; output LayerNorm: rows of 512/1024/2048; input LayerNorm: M <= 16, K %% 256 == 0, K <= 2048)", M, N, K, geglu, tail, pro);
    if (!x || !w || !y || !tickets || (tail && (!res || !gamma || !beta)) || (pro && (!pre_res || !pre_gamma || !pre_beta)))
        DB1_FAIL(DB1_ERR_BAD_SHAPE, "linear_decode: null operand");
    if (!db1_aligned16(x) || !db1_aligned16(w) || !db1_aligned16(y) || (ldx % 8) || (ldy % 8) ||
        (tail && (!db1_aligned16(res) || !db1_aligned16(ln_out) || (ld_res % 8) || (ld_out % 8))) ||
        (pro && (!db1_aligned16(pre_res) || !db1_aligned16(pre_out) || (ld_pre_res % 8) || (ld_pre_out % 8))))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "linear_decode: operands must be 16-byte aligned with leading dimensions in multiples of 8");
    if ((bias && !db1_dt_ok(dtBias)) || ((tail || pro) && !db1_dt_ok(dtParam))) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "linear_decode: dtype");
    SkinnyFusedArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.y = (bf16_t*)y; a.bias = bias; a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = K; a.ldy = ldy;
    a.S = pro ? 1 : skinny_fused_split(M, N, K, geglu);
    a.part = nullptr; a.tickets = (unsigned*)tickets;
    if (a.S > 1) {
        DB1_NEED_WS(ws, ws_bytes, db1_linear_decode_workspace_bytes(M, N, K, geglu), "linear_decode");
        a.part = (float*)ws;
    }
    a.res = (const bf16_t*)res; a.gamma = gamma; a.beta = beta; a.ln_out = (bf16_t*)ln_out; a.ld_res = ld_res; a.ld_out = ld_out; a.alpha = alpha; a.eps = eps;
    a.pre_res = (const bf16_t*)pre_res; a.pre_gamma = pre_gamma; a.pre_beta = pre_beta; a.pre_out = (bf16_t*)pre_out; a.ld_pre_res = ld_pre_res;
    a.ld_pre_out = ld_pre_out; a.pre_alpha = pre_alpha; a.pre_eps = pre_eps;
    const dim3 grid((unsigned)(geglu ? N / 8 : N / 16), (unsigned)a.S);
    hipStream_t st = (hipStream_t)stream;
    const int mt = (M + 15) / 16;
    const bool bb = bias && dtBias == DB1_BF16, pb = (tail || pro) && dtParam == DB1_BF16;
#define SKF(TB, TPp, MTv, G, PR) gemm_skinny_fused_kernel<TB, TPp, MTv, G, PR><<<grid, 256, 0, st>>>(a)
#define SKF_MT(TB, TPp, G) do { if (pro && M == 1) SKF(TB, TPp, 1, G, 1); else if (pro && M <= 4) SKF(TB, TPp, 1, G, 4); else if (pro) SKF(TB, TPp, 1, G, 16); \
                                else if (mt == 1) SKF(TB, TPp, 1, G, 0); else if (mt == 2) SKF(TB, TPp, 2, G, 0); \
                                else if (mt == 3) SKF(TB, TPp, 3, G, 0); else SKF(TB, TPp, 4, G, 0); } while (0)
#define SKF_G(TB, TPp) do { if (geglu) SKF_MT(TB, TPp, true); else SKF_MT(TB, TPp, false); } while (0)
    if (bb && pb) SKF_G(bf16_t, bf16_t);
    else if (bb) SKF_G(bf16_t, float);
    else if (pb) SKF_G(float, bf16_t);
    else SKF_G(float, float);
#undef SKF_G
#undef SKF_MT
#undef SKF
    DB1_CHECK_LAUNCH("linear_decode");
    return DB1_OK;
}


// y[M, N] = merge(attention partials)[M, H * 128] . W^T: the output projection of the inference layer fed directly by the per-chunk partial
// results of db1_relattn_decode_ring_fwd (called with out == NULL; q new tokens per sequence, B * q <= 2 rows, <= 12 chunks:
// 120 registers of partials per row and thread)
extern "C" int db1_linear_decode_attn_supported(int B, int q, int H, int D, int nunit, int N) {
    return (B >= 1 && q >= 1 && B * q <= 2 && D == 128 && H * D <= 2048 && (H * D) % 256 == 0 && nunit >= 1 && nunit <= 12 && N % 16 == 0 && N >= 16 &&
            N / 16 + 1 <= DB1_SKINNY_TICKETS) ? 1 : 0;
}
extern "C" int db1_linear_decode_attn(const float* attn_part, int nunit, int B, int q, int H, int D, const void* w, void* y, int64_t ldy, int N, void* stream) {
    if (!db1_linear_decode_attn_supported(B, q, H, D, nunit, N))
        DB1_FAIL(DB1_ERR_UNSUPPORTED, "linear_decode_attn: B=%d q=%d H=%d D=%d chunks=%d N=%d (needs B q <= 2, d_head = 128, H D <= 2048, <= 12 chunks)", B, q, H, D, nunit, N);
    if (!attn_part || !w || !y) DB1_FAIL(DB1_ERR_BAD_SHAPE, "linear_decode_attn: null operand");
    if (!db1_aligned16(attn_part) || !db1_aligned16(w) || !db1_aligned16(y) || (ldy % 8)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "linear_decode_attn: alignment");
    SkinnyFusedArgs a = {};
    a.x = (const bf16_t*)w;   // (never read: the rows come from attn_part)
    a.w = (const bf16_t*)w; a.y = (bf16_t*)y; a.bias = nullptr; a.M = B * q; a.N = N; a.K = H * D; a.ldx = 0; a.ldw = a.K; a.ldy = ldy; a.S = 1;
    a.att_part = attn_part; a.att_nunit = nunit; a.att_q = q; a.att_H = H;
    const dim3 grid((unsigned)(N / 16), 1u);
    hipStream_t st = (hipStream_t)stream;
    if (a.M == 1) gemm_skinny_fused_kernel<float, float, 1, false, -1><<<grid, 256, 0, st>>>(a);
    else gemm_skinny_fused_kernel<float, float, 1, false, -2><<<grid, 256, 0, st>>>(a);
    DB1_CHECK_LAUNCH("linear_decode_attn");
    return DB1_OK;
}
