// One new token through the linear maps of an inference layer as ONE launch (evaluate_rl.py:157-266: batch 1, 1-token calls with a full
// Transformer-XL memory; transformer_xl.py:227-243 o_net + post-LN, :246-292 PositionwiseFF, :136 the next layer's qkv_net).
//
// The 1-token call is a chain of W-streams of 8-34 MB, each a latency chain of its own when it is a launch (rows -> W round trip -> dot
// products -> store: 6-10 us for 2-7 us of HBM time).  Here the four streams between two attention launches
//     o_net (input: the merge of the attention's chunk partials)  ->  LayerNorm  ->  ff1 + GEGLU  ->  ff2  ->  LayerNorm  ->  next layer's qkv
// run inside one kernel of 256 workgroups (one per CU, 11 waves):
//   * eight WORKER waves per workgroup hold the W rows of o_net, ff1 and ff2 in registers; each publishes the rows it computed itself;
//   * one SERVICE wave per workgroup does everything that depends on other workgroups: it waits for the whole stage vector, applies the
//     residual LayerNorm and leaves the stage's input in LDS for the workers.  It never has W requests outstanding, so its waits cost
//     nothing (vmcnt retires in order: a worker that polled memory between its prefetches would wait for all of them);
//   * two DMA waves bring the rows of the next layer's qkv projection into LDS (96 KB, global_load_lds: no registers; in waves of their
//     own, because a compiler that sees LDS-DMA requests in flight drains vmcnt before every LDS read it cannot tell from their target).
// Order of the memory requests -- what the CU's memory pipeline allows (tools/exp/stream_first.hip, dbg_chain_inmodel.py):
//   a CU accepts ~160 KB of outstanding requests, a wave that asks for more stalls IN THE ISSUE until earlier data has returned, and the
//   CU serves its requests in the order they were made -- a store or poll issued behind a burst of bulk requests waits until the burst
//   has drained (24 KB / us per CU).  So every stage k runs
//       dot products -> each worker PUBLISHES its rows -> barrier B_k -> the service wave issues its first poll round -> barrier C_k
//       -> the workers request the weights of stage k + 1 (just those: they have to land before the next dot product anyway)
//       -> poll complete -> LayerNorm / activation row into LDS -> barrier A_k+1
//   and nothing is requested further ahead: W0 + half of W1 before the first stage, the rest of W1 at C0, W2 at C1, W3 (DMA) at A2.
// Hand-off between workgroups without a counter: every value is stored as a 32-bit word {bf16 value, 16-bit tag of this launch}; the service
// waves poll the ROW ITSELF (L1 / L2-bypassing loads) until all its words carry the launch's tag.  One memory round trip after the last
// producer's store instead of store -> wait -> counter add -> counter poll -> load: 2.05 us instead of 4.55 us per hand-off between 256
// workgroups (tools/exp/sync_bench.hip).  The tag only has to differ from the tag of the PREVIOUS launch on the same scratch (every launch
// rewrites every word of its three rows): tag = slot + 1, and consecutive launches must use different slots (the layer index, n_layer >= 2).
// M = 1 row: the products are VALU dot products (fp32 FMA, 8 bf16 of W per 16-byte load), the matrix pipe has nothing to gain at one row.
// Arithmetic per element as the launches this replaces (db1_linear_decode_attn / db1_linear_decode): the merged attention row, the LayerNorm
// inputs and outputs, z and act are rounded to bf16 where those store bf16; fp32 accumulation; results agree to fp32 summation order.
#include "ln_row.h"

#define DC_D 2048
#define DC_DFF 4096
#define DC_WG 256
#define DC_WORKERS 8
#define DC_DMA_WAVES 2
#define DC_THREADS ((DC_WORKERS + 1 + DC_DMA_WAVES) * 64)
#define DC_SPIN_LIMIT (1 << 17)
#ifndef DC_W1_EARLY
#define DC_W1_EARLY 2   /* ff1 rows (of 4 per wave) requested before the first stage's dot product */
#endif

typedef __attribute__((ext_vector_type(4))) unsigned dc_u32x4;
// workgroup barrier WITHOUT the vmcnt(0) a __syncthreads() carries (its fence would drain the workers' W prefetches at every stage): LDS
// traffic of this wave is complete (lgkmcnt), global loads stay in flight
#define DC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define DC_PIN() __builtin_amdgcn_sched_barrier(0)   /* keep the request batches where they are written (the scheduler would sink them to their uses) */

struct DecodeChainArgs {
    const float* att_part; int att_nunit, att_H;        // attention partials [H][nunit][64][128 + 2], query row 0
    const bf16_t* x_res;                                 // the layer's input row [d] (residual of the first LayerNorm)
    const bf16_t* w_o; const bf16_t* w1; const bf16_t* w2; const bf16_t* w_qkv;   // [d, d], [2 dff, d], [d, dff], [3 d, d] or null (last layer)
    const bf16_t* b1; const bf16_t* b2;                  // [2 dff], [d]
    const bf16_t* w_o_next;                              // the NEXT launch's w_o (or null): its rows are touched here so that they wait in L2 / the memory-side cache
    const bf16_t* g1; const bf16_t* be1; const bf16_t* g2; const bf16_t* be2;    // LayerNorm parameters
    float alpha, eps;
    unsigned* y_o; unsigned* act; unsigned* f;           // hand-off rows between the stages (global scratch), tagged words: [d], [dff], [d]
    unsigned tag;                                        // this launch's tag (low 16 bits of every word)
    bf16_t* h1_out; bf16_t* f_out;                       // last layer only: LN1's output and ff2's output (the pending LayerNorm of the head), [d] each
    bf16_t* x_next; bf16_t* qkv_next;                    // [d]: LN2 output = the next layer's input; [3 d]: its projection (w_qkv != null)
    int* err;                                            // set to 1 if a poll ran into its limit (results are then invalid)
    unsigned long long* ts;                              // test hook (db1_test_decode_chain_timestamps): workgroup 0's stage times, 100 MHz ticks; null in production
};
#define DC_TSW(k) do { } while (0)   /* (worker-side stamps: their stores enter the wave's vmcnt order and move the waits they are meant to time) */
#define DC_TS(k) do { if (p.ts && lane == 0) { if (bid == 0) p.ts[k] = wall_clock64(); if ((k) < 4) p.ts[16 + bid * 4 + (k)] = wall_clock64(); } } while (0)

__device__ __forceinline__ void dc_st_agent(void* p, unsigned lo, unsigned hi) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float dc_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float dc_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned dc_word(float v, unsigned tag) { return (f2bf_pk(0.f, v) & 0xffff0000u) | tag; }   // {bf16(v), tag}

// sum over the 64 lanes, the same value in every lane: DPP adds inside the rows of 16 (4 x ~8 cycles) and one readlane per row, instead of
// six dependent ds_bpermute round trips (~0.3 us per sum: the LayerNorms and dot products here are latency chains of such sums)
template <int CTRL> __device__ __forceinline__ float dc_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float dc_wave_sum(float x) {
    x += dc_dpp<0xB1>(x);    // quad_perm [1, 0, 3, 2]
    x += dc_dpp<0x4E>(x);    // quad_perm [2, 3, 0, 1]
    x += dc_dpp<0x124>(x);   // row_ror 4
    x += dc_dpp<0x128>(x);   // row_ror 8: every lane holds its row's sum
    const int xi = __float_as_int(x);
    return (__int_as_float(__builtin_amdgcn_readlane(xi, 0)) + __int_as_float(__builtin_amdgcn_readlane(xi, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(xi, 32)) + __int_as_float(__builtin_amdgcn_readlane(xi, 48)));
}
// LayerNorm statistics of one row held by one wave (the arithmetic of ln_row_stats: the sum is a bf16 tensor in the reference, two passes)
__device__ __forceinline__ void dc_ln_stats(Vec16<bf16_t> (&a)[4], float eps, float& mu, float& rs) {
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 8; j++) { a[k].v[j] = bf2f(f2bf(a[k].v[j])); sum += a[k].v[j]; }
    mu = dc_wave_sum(sum) * (1.f / DC_D);
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 8; j++) { const float c = a[k].v[j] - mu; sq += c * c; }
    rs = rsqrtf(dc_wave_sum(sq) * (1.f / DC_D) + eps);
}

// element j (0 .. 7) of eight bf16 packed in four dwords
__device__ __forceinline__ float dc_el(const dc_u32x4& v, int j) { return (j & 1) ? dc_hi(v[j >> 1]) : dc_lo(v[j >> 1]); }
__device__ __forceinline__ dc_u32x4 dc_ld16(const bf16_t* p) { return *reinterpret_cast<const dc_u32x4*>(p); }

// NL 16-byte pieces of one W row for this lane: columns (j * 64 + lane) * 8 .. + 7
template <int NL> struct DcRow { dc_u32x4 w[NL]; };
template <int NL> __device__ __forceinline__ void dc_load_row(DcRow<NL>& r, const bf16_t* row, int lane) {
#pragma unroll
    for (int j = 0; j < NL; j++) r.w[j] = __builtin_nontemporal_load(reinterpret_cast<const dc_u32x4*>(row + (j * 64 + lane) * 8));
}
__device__ __forceinline__ float dc_fma8(const dc_u32x4 w, const dc_u32x4 x, float acc) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        acc = fmaf(dc_lo(w[e]), dc_lo(x[e]), acc);
        acc = fmaf(dc_hi(w[e]), dc_hi(x[e]), acc);
    }
    return acc;
}
template <int NL> __device__ __forceinline__ float dc_dot(const DcRow<NL>& r, const bf16_t* xs, int lane) {   // xs: the input vector in LDS
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NL; j++) acc = dc_fma8(r.w[j], *reinterpret_cast<const dc_u32x4*>(xs + (j * 64 + lane) * 8), acc);
    return dc_wave_sum(acc);
}

// service wave: the words (k * 64 + lane) * 8 .. + 7 of a tagged row for two k, read past L1 / L2 (sc1)
__device__ __forceinline__ void dc_read_pair(const unsigned* a, dc_u32x4& w0, dc_u32x4& w1, dc_u32x4& w2, dc_u32x4& w3) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:2064 sc1"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(a) : "memory");
}
// one poll round of a tagged row: w[2 k + h] = words (k * 64 + lane) * 8 + 4 h .. + 3, k < 2 NP.  issue: the requests only; land: wait for them
// (the registers are operands of the wait, so nothing reads them before it); check: every word carries the tag?
template <int NP> __device__ __forceinline__ void dc_poll_issue(const unsigned* row, int lane, dc_u32x4 (&w)[4 * NP]) {
#pragma unroll
    for (int q = 0; q < NP; q++) dc_read_pair(row + q * 1024 + lane * 8, w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
__device__ __forceinline__ void dc_poll_land(dc_u32x4 (&w)[8]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) :: "memory");
}
__device__ __forceinline__ void dc_poll_land(dc_u32x4 (&w)[16]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]),
                 "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]) :: "memory");
}
template <int N> __device__ __forceinline__ bool dc_poll_check(const dc_u32x4 (&w)[N], unsigned tag) {
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) bad |= (w[i][e] ^ tag) & 0xffffu;
    return __all(bad == 0);
}
// the rounds after the first (which the caller issued ahead of the workers' next bulk requests): false if the poll limit ran out
template <int NP> __device__ __forceinline__ bool dc_poll_rest(const unsigned* row, unsigned tag, int lane, dc_u32x4 (&w)[4 * NP]) {
    dc_poll_land(w);
    for (int spins = 0; spins < DC_SPIN_LIMIT; spins++) {
        if (dc_poll_check(w, tag)) return true;
        __builtin_amdgcn_s_sleep(2);
        dc_poll_issue<NP>(row, lane, w);
        dc_poll_land(w);
    }
    return false;
}

template <int NU>   // chunks of the attention partials the merge is unrolled for (>= att_nunit)
__global__ __launch_bounds__(DC_THREADS, 1) void decode_chain_kernel(DecodeChainArgs p) {
    // (separate LDS objects: the compiler then knows that reads of xs / res never touch the rows the LDS-DMA requests are still writing,
    //  and does not drain vmcnt before them)
    __shared__ __attribute__((aligned(16))) bf16_t xs[DC_DFF];
    __shared__ __attribute__((aligned(16))) char w3s[24 * DC_D * 2];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), bid = blockIdx.x;
    const bool last_layer = p.w_qkv == nullptr;

    if (wave < DC_WORKERS) {
        // ------------------------------------------------------------------------------------------------ worker waves
        const unsigned tag = p.tag;
        float bias1[4], bias2;          // of this wave's ff1 rows ([value, gate] x 2 pairs) and its ff2 row
#pragma unroll
        for (int r = 0; r < 4; r++) bias1[r] = bf2f(p.b1[(r & 1 ? DC_DFF : 0) + bid * 16 + wave * 2 + (r >> 1)]);
        bias2 = bf2f(p.b2[bid * 8 + wave]);
        DcRow<4> w0;                    // o_net: row bid * 8 + wave
        dc_load_row(w0, p.w_o + (int64_t)(bid * 8 + wave) * DC_D, lane);
        DC_PIN();
        DC_TSW(0);
        // stage 0 input: the merge of the attention's chunk partials (worker thread t owns columns 4 t .. 4 t + 3 = head t / 32), the arithmetic
        // of relattn_decode_merge2_kernel / db1_linear_decode_attn.  Its inputs are requested, then W1, and only then are they waited for:
        //   ff1 rows of this wave: pairs pr = bid * 16 + wave * 2 + {0, 1}: value row pr, gate row dff + pr          (16 pieces)
        //   (ff2's row follows after B0: a CU takes about 160 KB of requests before the requesting wave stalls in the issue -- with W2 up
        //    here the workers reached A0 at 5-7 us instead of 1.5 us; the next layer's qkv rows go to LDS, requested by the DMA waves after A1)
        constexpr int DD = 128;
        const int t = threadIdx.x, hh = (t * 4) / DD, d0 = (t * 4) % DD;
        float4 ov[NU];
        float2 ml[NU];
        {
            const float* src = p.att_part + ((int64_t)hh * p.att_nunit * 64) * (DD + 2);
#pragma unroll
            for (int c = 0; c < NU; c++) {     // (row 0 of a unit starts 16-byte aligned: 64 * 130 floats per unit)
                if (c < p.att_nunit) {
                    const float* u = src + (int64_t)c * 64 * (DD + 2);
                    ml[c] = *reinterpret_cast<const float2*>(u + DD);
                    ov[c] = *reinterpret_cast<const float4*>(u + d0);
                } else {
                    ml[c] = make_float2(-1.0e30f, 0.f); ov[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        DC_PIN();
        DC_TSW(1);
        DcRow<4> w1[4];                 // [0] value row pr0, [1] gate row pr0, [2] value row pr1, [3] gate row pr1
        auto w1_row = [&](int r) { return p.w1 + (int64_t)((r & 1 ? DC_DFF : 0) + bid * 16 + wave * 2 + (r >> 1)) * DC_D; };
#pragma unroll
        for (int r = 0; r < DC_W1_EARLY; r++) dc_load_row(w1[r], w1_row(r), lane);
        DC_PIN();
        DC_TSW(2);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * DC_W1_EARLY) : "memory");   // W0 and the merge inputs have landed; the early W1 rows may be in flight
        {
            float mx = -1.0e30f;
#pragma unroll
            for (int c = 0; c < NU; c++) mx = fmaxf(mx, ml[c].x);
            float l = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
            for (int c = 0; c < NU; c++) {
                const float wt = ml[c].y > 0.f ? __expf(ml[c].x - mx) : 0.f;
                l += ml[c].y * wt;
                o0 += ov[c].x * wt; o1 += ov[c].y * wt; o2 += ov[c].z * wt; o3 += ov[c].w * wt;
            }
            uint2 w;
            w.x = f2bf_pk(l > 0.f ? o0 / l : 0.f, l > 0.f ? o1 / l : 0.f); w.y = f2bf_pk(l > 0.f ? o2 / l : 0.f, l > 0.f ? o3 / l : 0.f);
            *reinterpret_cast<uint2*>(&xs[t * 4]) = w;
        }
        DC_TSW(3);
        DC_BARRIER();                                   // A0: the merged row is in LDS (worker waves wrote it)
        {   // y_o row bid * 8 + wave, published by this wave itself BEFORE it requests anything else: a CU's memory requests are served in
            // the order they were made, and bulk requests take microseconds to drain
            const float s = dc_dot(w0, xs, lane);
            if (lane == 0) __hip_atomic_store(p.y_o + bid * 8 + wave, dc_word(s, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        DC_BARRIER();                                   // B0: the stage's input row is free again
        DC_BARRIER();                                   // C0: the service wave's first poll round is queued; now the next stage's weights
        // ---- stage 1: ff1 + GEGLU.  W2 requested now: a CU takes about 160 KB of requests before the requesting wave stalls in the issue
        // (W0 + W1 = 160 KB; with W2 on top the workers reached A0 at 5-7 us instead of 1.5 us)
#pragma unroll
        for (int r = DC_W1_EARLY; r < 4; r++) dc_load_row(w1[r], w1_row(r), lane);
        DC_PIN();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DC_BARRIER();                                   // A1: LN1 row in LDS
        {   // z = bf16(sum + bias) for both halves, act = bf16(z_v * gelu(z_g)): pairs bid * 16 + wave * 2 + {0, 1}
            float sv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) sv[j] = dc_dot(w1[j], xs, lane);
            const float a0 = bf2f(f2bf(sv[0] + bias1[0])) * gelu_fwd_t<bf16_t>(bf2f(f2bf(sv[1] + bias1[1])));
            const float a1 = bf2f(f2bf(sv[2] + bias1[2])) * gelu_fwd_t<bf16_t>(bf2f(f2bf(sv[3] + bias1[3])));
            if (lane == 0) dc_st_agent(p.act + bid * 16 + wave * 2, dc_word(a0, tag), dc_word(a1, tag));
        }
        DC_BARRIER();                                   // B1
        DC_BARRIER();                                   // C1
        DcRow<8> w2;                    // ff2 row bid * 8 + wave (K = dff), requested behind the service wave's poll of the act row
        dc_load_row(w2, p.w2 + (int64_t)(bid * 8 + wave) * DC_DFF, lane);
        DC_PIN();
        // ---- stage 2: ff2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DC_BARRIER();                                   // A2: act row (dff) in LDS
        {
            const float v = dc_dot(w2, xs, lane) + bias2;
            if (lane == 0) {
                __hip_atomic_store(p.f + bid * 8 + wave, dc_word(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (p.f_out) p.f_out[bid * 8 + wave] = f2bf(v);   // (last layer: the head reads a plain bf16 row)
            }
        }
        DC_BARRIER();                                   // B2
        if (last_layer) return;
        DC_BARRIER();                                   // C2
        // ---- stage 3: the next layer's qkv projection of LN2's row; W from LDS (the DMA waves waited for their requests before this barrier)
        DC_BARRIER();                                   // A3: LN2 row in LDS, W3 rows in LDS
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++)
                acc = dc_fma8(*reinterpret_cast<const dc_u32x4*>(w3s + ((wave * 3 + j) * 4 + i) * 1024 + lane * 16),
                              *reinterpret_cast<const dc_u32x4*>(xs + (i * 64 + lane) * 8), acc);
            const float s = dc_wave_sum(acc);
            if (lane == 0) p.qkv_next[bid * 24 + wave * 3 + j] = f2bf(s);   // (no bias; read by the NEXT launch: a plain store is enough)
        }
        return;
    }

    if (wave > DC_WORKERS) {
        // ---------------------------------------------------------------------------------------------- LDS-DMA waves
        // the next layer's qkv rows bid * 24 .. + 23 (96 KB) -> LDS, 12 rows per wave, requested once the workers' requests are queued (after A0):
        // they arrive last, as they are needed last, and no register holds them.  (In a wave of their own: a compiler that sees LDS-DMA requests
        // in flight drains vmcnt before every LDS read it cannot tell apart from their destination -- in a worker that is every stage's input.)
        const int v = wave - DC_WORKERS - 1;
        DC_BARRIER();                                   // A0
        DC_BARRIER();                                   // B0
        DC_BARRIER();                                   // C0
        DC_BARRIER();                                   // A1
        // the next launch opens with a cold read of ITS o_net rows: ask for one dword of each of their 128-byte lines now -- same workgroup
        // index = same XCD = same L2 next time; the values are dropped
        unsigned touch0 = 0, touch1 = 0;
        if (p.w_o_next) {
            const char* nx = reinterpret_cast<const char*>(p.w_o_next + (int64_t)(bid * 8 + v * 4) * DC_D) + lane * 128;   // this wave: 4 rows = 16 KB = 128 lines
            touch0 = *reinterpret_cast<const unsigned*>(nx); touch1 = *reinterpret_cast<const unsigned*>(nx + 8192);
        }
        DC_BARRIER();                                   // B1
        DC_BARRIER();                                   // C1
        DC_BARRIER();                                   // A2: W0, W1 and W2 have landed in every wave of the workgroup -- the CU's request queue is empty
        if (!last_layer) {
#pragma unroll
            for (int j = 0; j < 12; j++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    __builtin_amdgcn_global_load_lds(p.w_qkv + (int64_t)(bid * 24 + v * 12 + j) * DC_D + (i * 64 + lane) * 8,
                                                     (__attribute__((address_space(3))) void*)(w3s + ((v * 12 + j) * 4 + i) * 1024), 16, 0, 0);
        }
        DC_BARRIER();                                   // B2
        asm volatile("" :: "v"(touch0), "v"(touch1));      // (the only use of the touched lines)
        if (last_layer) return;
        DC_BARRIER();                                   // C2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows are in LDS
        DC_BARRIER();                                   // A3
        return;
    }

    // ---------------------------------------------------------------------------------------------------- service wave
    // LayerNorm parameters and the residual row of LN1: requested up front (plain loads: written before this launch)
    const unsigned tag = p.tag;
    dc_u32x4 xr[4], gg[4], bb[4];          // (eight bf16 per 16 bytes: kept packed, the wave holds a whole polled row next to them)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        xr[k] = dc_ld16(p.x_res + (k * 64 + lane) * 8);
        gg[k] = dc_ld16(p.g1 + (k * 64 + lane) * 8);
        bb[k] = dc_ld16(p.be1 + (k * 64 + lane) * 8);
    }
    DC_TS(0);
    DC_BARRIER();                                       // A0 (the workers prepared the stage-0 input themselves)
    DC_TS(1);
    DC_BARRIER();                                       // B0: every worker of this workgroup has published its y_o row
    DC_TS(2);
    bool live = true;
    // ---- LN1: h1 = LN(alpha * x_res + y_o) * g1 + be1
    dc_u32x4 h1[4];
    {
        dc_u32x4 w[8];
        dc_poll_issue<2>(p.y_o, lane, w);
        DC_BARRIER();                                   // C0
        live = dc_poll_rest<2>(p.y_o, tag, lane, w) && live;
        DC_TS(3);
        Vec16<bf16_t> a[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int j = 0; j < 8; j++) a[k].v[j] = p.alpha * dc_el(xr[k], j) + dc_hi(w[2 * k + (j >> 2)][j & 3]);
        float mu, rs;
        dc_ln_stats(a, p.eps, mu, rs);
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int e = 0; e < 4; e++)                     // h1 is a bf16 row
                h1[k][e] = f2bf_pk((a[k].v[2 * e] - mu) * rs * dc_el(gg[k], 2 * e) + dc_el(bb[k], 2 * e),
                                   (a[k].v[2 * e + 1] - mu) * rs * dc_el(gg[k], 2 * e + 1) + dc_el(bb[k], 2 * e + 1));
            *reinterpret_cast<dc_u32x4*>(&xs[(k * 64 + lane) * 8]) = h1[k];
            if (bid == 0 && p.h1_out) *reinterpret_cast<dc_u32x4*>(p.h1_out + (k * 64 + lane) * 8) = h1[k];
        }
    }
    DC_TS(4);
    DC_BARRIER();                                       // A1
    DC_TS(5);
    DC_BARRIER();                                       // B1
    DC_TS(6);
    {   // act row (dff values) -> LDS as bf16
        dc_u32x4 w[16];
        dc_poll_issue<4>(p.act, lane, w);
        DC_BARRIER();                                   // C1
        live = dc_poll_rest<4>(p.act, tag, lane, w) && live;
        DC_TS(7);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            dc_u32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (w[2 * k + (e >> 1)][2 * (e & 1)] >> 16) | (w[2 * k + (e >> 1)][2 * (e & 1) + 1] & 0xffff0000u);
            *reinterpret_cast<dc_u32x4*>(xs + (k * 64 + lane) * 8) = v;
        }
    }
    // (the parameters of LN2: requested now that the act row's registers are free, used after the next hand-off)
#pragma unroll
    for (int k = 0; k < 4; k++) { gg[k] = dc_ld16(p.g2 + (k * 64 + lane) * 8); bb[k] = dc_ld16(p.be2 + (k * 64 + lane) * 8); }
    DC_TS(8);
    DC_BARRIER();                                       // A2
    DC_TS(9);
    DC_BARRIER();                                       // B2
    DC_TS(10);
    if (last_layer) {
        if (!live && lane == 0) *p.err = 1;
        return;                                            // (the head's projection normalises LN2's input rows itself: h1_out and f_out)
    }
    {   // LN2: x_next = LN(alpha * h1 + f) * g2 + be2
        dc_u32x4 w[8];
        dc_poll_issue<2>(p.f, lane, w);
        DC_BARRIER();                                   // C2
        live = dc_poll_rest<2>(p.f, tag, lane, w) && live;
        DC_TS(11);
        Vec16<bf16_t> a[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int j = 0; j < 8; j++) a[k].v[j] = p.alpha * dc_el(h1[k], j) + dc_hi(w[2 * k + (j >> 2)][j & 3]);
        float mu, rs;
        dc_ln_stats(a, p.eps, mu, rs);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            Vec16<bf16_t> o;
#pragma unroll
            for (int j = 0; j < 8; j++) o.v[j] = (a[k].v[j] - mu) * rs * dc_el(gg[k], j) + dc_el(bb[k], j);
            o.store(&xs[(k * 64 + lane) * 8]);
            if (bid == 0) o.store(p.x_next + (k * 64 + lane) * 8);
        }
    }
    if (!live && lane == 0) *p.err = 1;
    DC_TS(12);
    DC_BARRIER();                                       // A3
    DC_TS(13);
}

static unsigned long long* g_dc_ts = nullptr;
static int g_dc_ts_slot = -1;
extern "C" void db1_test_decode_chain_timestamps(void* buf, int slot) { g_dc_ts = (unsigned long long*)buf; g_dc_ts_slot = slot; }

// the 256 workgroups wait for each other: every one of them needs a CU of its own (a partitioned or smaller device would run into the poll
// bound instead of finishing) -- the current device's CU count, looked up once per device
static int dc_device_fits() {
    static int cus[64];
    static Db1PerDeviceOnce once;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    once.run([dev] {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus[dev & 63] = n;
    });
    return cus[dev & 63] >= DC_WG;
}
extern "C" int db1_decode_chain_supported(int d, int dff, int H, int D, int nunit) {
    return (d == DC_D && dff == DC_DFF && D == 128 && H * D == d && nunit >= 1 && nunit <= 12 && dc_device_fits()) ? 1 : 0;
}
// scratch: the tagged rows y_o | act | f (32-bit words, shared by all layers: a layer's launch ends before the next one starts), then the error
// flag.  The caller zeroes it ONCE, when it allocates it (tag 0 is never a launch's tag), and never again.
#define DC_SLOTS 65535
extern "C" int64_t db1_decode_chain_error_offset(void) { return (int64_t)(2 * DC_D + DC_DFF) * 4; }
extern "C" int64_t db1_decode_chain_scratch_bytes(void) { return db1_decode_chain_error_offset() + 64; }

extern "C" int db1_decode_chain(const float* att_part, int nunit, int H, const void* x_res, const void* w_o, const void* w1, const void* b1, const void* w2,
                                const void* b2, const void* w_qkv_next, const void* w_o_next, const void* g1, const void* be1, const void* g2, const void* be2, float alpha, float eps,
                                void* h1_out, void* f_out, void* x_next, void* qkv_next, void* scratch, int slot, int d, int dff, void* stream) {
    if (!db1_decode_chain_supported(d, dff, H, 128, nunit)) DB1_FAIL(DB1_ERR_UNSUPPORTED, "decode_chain: d=%d dff=%d H=%d chunks=%d (built for d 2048, dff 4096, d_head 128, <= 12 chunks)", d, dff, H, nunit);
    if (slot < 0 || slot >= DC_SLOTS) DB1_FAIL(DB1_ERR_BAD_SHAPE, "decode_chain: slot %d", slot);
    if (!att_part || !x_res || !w_o || !w1 || !b1 || !w2 || !b2 || !g1 || !be1 || !g2 || !be2 || !scratch || (w_qkv_next && (!x_next || !qkv_next)) || (!w_qkv_next && (!h1_out || !f_out)))
        DB1_FAIL(DB1_ERR_BAD_SHAPE, "decode_chain: null operand");
    if (!db1_aligned16(x_res) || !db1_aligned16(w_o) || !db1_aligned16(w1) || !db1_aligned16(w2) || (w_qkv_next && !db1_aligned16(w_qkv_next)) || !db1_aligned16(scratch) ||
        !db1_aligned16(g1) || !db1_aligned16(be1) || !db1_aligned16(g2) || !db1_aligned16(be2) || (x_next && !db1_aligned16(x_next)) || (h1_out && !db1_aligned16(h1_out)) ||
        (qkv_next && ((uintptr_t)qkv_next & 7)) || (f_out && ((uintptr_t)f_out & 3)))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "decode_chain: operands must be 16-byte aligned");
    DecodeChainArgs a;
    a.att_part = att_part; a.att_nunit = nunit; a.att_H = H;
    a.x_res = (const bf16_t*)x_res; a.w_o = (const bf16_t*)w_o; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w_qkv = (const bf16_t*)w_qkv_next; a.w_o_next = (const bf16_t*)w_o_next;
    a.b1 = (const bf16_t*)b1; a.b2 = (const bf16_t*)b2; a.g1 = (const bf16_t*)g1; a.be1 = (const bf16_t*)be1; a.g2 = (const bf16_t*)g2; a.be2 = (const bf16_t*)be2;
    a.alpha = alpha; a.eps = eps;
    unsigned* s = (unsigned*)scratch;
    a.y_o = s; a.act = s + DC_D; a.f = s + DC_D + DC_DFF;
    a.err = (int*)(s + 2 * DC_D + DC_DFF);
    a.tag = (unsigned)slot + 1u;
    a.ts = (g_dc_ts_slot < 0 || g_dc_ts_slot == slot) ? g_dc_ts : nullptr;
    a.h1_out = (bf16_t*)h1_out; a.f_out = (bf16_t*)f_out; a.x_next = (bf16_t*)x_next; a.qkv_next = (bf16_t*)qkv_next;
    if (nunit <= 9) decode_chain_kernel<9><<<DC_WG, DC_THREADS, 0, (hipStream_t)stream>>>(a);   // (mem_len 1024 + 1 token = 9 chunks)
    else decode_chain_kernel<12><<<DC_WG, DC_THREADS, 0, (hipStream_t)stream>>>(a);
    DB1_CHECK_LAUNCH("decode_chain");
    return DB1_OK;
}
