// Relative-position attention, materialised path (fp32 parity gate / any head size).
// The contractions (AC = (q+u)K^T, T = (q+v)R^T, PV, and all backward products) run on the strided GEMM;
// this file holds the three pieces that are not contractions:
//   * q + u / q + v_bias                                      (transformer_xl.py:161,167)
//   * the closed form of _rel_shift + mask + softmax           (:98-110, 171-209, 551-567)
//   * its backward, which also re-indexes dS by distance (dT) for the dq_r / dR products.
// Score buffers are float32 in [H][B][Lq][*] layout so that, per head, the (batch, query) rows
// are contiguous (the dR contraction then runs over B*Lq with a single stride).
#include "db1_common.h"

template <typename T, typename TP>
__global__ __launch_bounds__(256) void add_head_bias_kernel(const T* __restrict__ qkv, const TP* __restrict__ u, const TP* __restrict__ vb,
                                                            T* __restrict__ qu, T* __restrict__ qv, int64_t n_tok_q, int Lq, int Lk, int HD) {
    // qkv rows are the Lk key positions per batch; queries are the LAST Lq of them (transformer_xl.py:133)
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n_tok_q * HD; idx += (int64_t)gridDim.x * 256) {
        const int64_t t = idx / HD;
        const int c = (int)(idx % HD);
        const int64_t b = t / Lq, i = t % Lq;
        const float q = ldf(qkv + ((b * Lk + (Lk - Lq) + i) * 3) * HD + c);
        stf(qu + idx, q + ldf(u + c));
        stf(qv + idx, q + ldf(vb + c));
    }
}

// 16-byte vectors (HD % V == 0): thread = V consecutive channels of one query token
template <typename T, typename TP>
__global__ __launch_bounds__(256) void add_head_bias_vec_kernel(const T* __restrict__ qkv, const TP* __restrict__ u, const TP* __restrict__ vb,
                                                                T* __restrict__ qu, T* __restrict__ qv, int64_t n_tok_q, int Lq, int Lk, int HD) {
    constexpr int V = Vec16<T>::N;
    const int vpr = HD / V;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n_tok_q * vpr; idx += (int64_t)gridDim.x * 256) {
        const int64_t t = idx / vpr;
        const int c = (int)(idx - t * vpr) * V;
        const int64_t b = t / Lq, i = t - b * Lq;
        Vec16<T> q, a, o;
        q.load(qkv + ((b * Lk + (Lk - Lq) + i) * 3) * HD + c);
#pragma unroll
        for (int j = 0; j < V; j++) { a.v[j] = q.v[j] + ldf(u + c + j); o.v[j] = q.v[j] + ldf(vb + c + j); }
        a.store(qu + t * HD + c);
        o.store(qv + t * HD + c);
    }
}

extern "C" int db1_relattn_add_head_bias(const void* qkv, const void* u, const void* vb, void* qu, void* qv, int B, int Lq, int Lk,
                                         int H, int D, int dt, int dtParam, void* stream) {
    if (!db1_dt_ok(dt) || !db1_dt_ok(dtParam)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "add_head_bias: dtype");
    if (B <= 0 || Lq <= 0 || Lk < Lq || H <= 0 || D <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "add_head_bias: shape");
    const int64_t n = (int64_t)B * Lq;
    int64_t blocks = (n * H * D + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    const int Vv = dt == DB1_F32 ? 4 : 8;
    if ((H * D) % Vv == 0 && db1_aligned16(qkv) && db1_aligned16(qu) && db1_aligned16(qv)) {
        int64_t vb_ = (n * (H * D / Vv) + 255) / 256;
        if (vb_ > 256 * 32) vb_ = 256 * 32;
#define LV_(T, TP) add_head_bias_vec_kernel<T, TP><<<(unsigned)vb_, 256, 0, st>>>((const T*)qkv, (const TP*)u, (const TP*)vb, (T*)qu, (T*)qv, n, Lq, Lk, H * D)
        if (dt == DB1_F32 && dtParam == DB1_F32) LV_(float, float);
        else if (dt == DB1_BF16 && dtParam == DB1_BF16) LV_(bf16_t, bf16_t);
        else if (dt == DB1_BF16) LV_(bf16_t, float);
        else LV_(float, bf16_t);
#undef LV_
        DB1_CHECK_LAUNCH("add_head_bias (vec)");
        return DB1_OK;
    }
#define L_(T, TP) add_head_bias_kernel<T, TP><<<(unsigned)blocks, 256, 0, st>>>((const T*)qkv, (const TP*)u, (const TP*)vb, (T*)qu, (T*)qv, n, Lq, Lk, H * D)
    if (dt == DB1_F32 && dtParam == DB1_F32) L_(float, float);
    else if (dt == DB1_BF16 && dtParam == DB1_BF16) L_(bf16_t, bf16_t);
    else if (dt == DB1_BF16) L_(bf16_t, float);
    else L_(float, bf16_t);
#undef L_
    DB1_CHECK_LAUNCH("add_head_bias");
    return DB1_OK;
}

// one 256-thread block per (h, b, i) row
__global__ __launch_bounds__(256) void relattn_softmax_fwd_kernel(float* AC, const float* __restrict__ T, float* lse, int Lq, int Lk, int nd,
                                                                  int mlen, int shift, float scale) {
    __shared__ float sm[4];
    const int64_t row = blockIdx.x;  // (h*B + b)*Lq + i
    const int i = (int)(row % Lq);
    float* s = AC + row * Lk;
    const float* t = T + row * nd;
    const int jlo = i - shift + 1 > 0 ? i - shift + 1 : 0;         // visible: i - shift < j <= i + mlen
    const int jhi = i + mlen < Lk - 1 ? i + mlen : Lk - 1;
    if (jlo > jhi) {
        // no visible key: the reference fills EVERY score of the row with -1e30 (masked_fill, transformer_xl.py:181-204) and the softmax of
        // a constant row is uniform over all Lk keys (e.g. same_length with the default mem_len = 0).  masked_fill passes no gradient,
        // which is what the backward kernel produces for such a row (dS = dT = 0).
        const float u = 1.f / (float)Lk;
        for (int j = threadIdx.x; j < Lk; j += 256) s[j] = u;
        if (lse && threadIdx.x == 0) lse[row] = -1e30f + logf((float)Lk);
        return;
    }
    float m = -3.0e38f;
    for (int j = jlo + threadIdx.x; j <= jhi; j += 256) {
        const float v = (s[j] + t[mlen + i - j]) * scale;
        s[j] = v;
        m = fmaxf(m, v);
    }
    m = block_max256(m, sm);
    float sum = 0.f;
    for (int j = jlo + threadIdx.x; j <= jhi; j += 256) {
        const float e = __expf(s[j] - m);
        s[j] = e;
        sum += e;
    }
    sum = block_sum256(sum, sm);
    const float inv = 1.f / sum;
    for (int j = threadIdx.x; j < Lk; j += 256) s[j] = (j >= jlo && j <= jhi) ? s[j] * inv : 0.f;
    if (lse && threadIdx.x == 0) lse[row] = m + logf(sum);
}

extern "C" int db1_relattn_softmax_fwd(float* AC, const float* T, float* lse, int H, int B, int Lq, int Lk, int nd, int mlen, int shift,
                                       float scale, void* stream) {
    if (H <= 0 || B <= 0 || Lq <= 0 || Lk <= 0 || nd < mlen + Lq) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_softmax_fwd: shape (nd=%d must be >= mlen+Lq)", nd);
    const int64_t rows = (int64_t)H * B * Lq;
    relattn_softmax_fwd_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(AC, T, lse, Lq, Lk, nd, mlen, shift, scale);
    DB1_CHECK_LAUNCH("relattn_softmax_fwd");
    return DB1_OK;
}

__global__ __launch_bounds__(256) void relattn_softmax_bwd_kernel(const float* __restrict__ P, float* dP, float* dT, int Lq, int Lk, int nd,
                                                                  int mlen, int shift, float scale) {
    __shared__ float sm[4];
    const int64_t row = blockIdx.x;
    const int i = (int)(row % Lq);
    const float* p = P + row * Lk;
    float* dp = dP + row * Lk;
    float* dt = dT + row * nd;
    const int jlo = i - shift + 1 > 0 ? i - shift + 1 : 0;
    const int jhi = i + mlen < Lk - 1 ? i + mlen : Lk - 1;
    float dot = 0.f;
    for (int j = jlo + threadIdx.x; j <= jhi; j += 256) dot += p[j] * dp[j];
    dot = block_sum256(dot, sm);
    for (int r = threadIdx.x; r < nd; r += 256) dt[r] = 0.f;
    __syncthreads();
    for (int j = threadIdx.x; j < Lk; j += 256) {
        float ds = 0.f;
        if (j >= jlo && j <= jhi) {
            ds = p[j] * (dp[j] - dot) * scale;
            dt[mlen + i - j] = ds;
        }
        dp[j] = ds;
    }
}

extern "C" int db1_relattn_softmax_bwd(const float* P, float* dP, float* dT, int H, int B, int Lq, int Lk, int nd, int mlen, int shift,
                                       float scale, void* stream) {
    if (H <= 0 || B <= 0 || Lq <= 0 || Lk <= 0 || nd < mlen + Lq) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_softmax_bwd: shape");
    const int64_t rows = (int64_t)H * B * Lq;
    relattn_softmax_bwd_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(P, dP, dT, Lq, Lk, nd, mlen, shift, scale);
    DB1_CHECK_LAUNCH("relattn_softmax_bwd");
    return DB1_OK;
}
