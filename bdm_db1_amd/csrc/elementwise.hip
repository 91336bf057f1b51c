// HBM-bound kernels of the DB1 hot path: residual+LayerNorm, FF activation, column sums,
// embedding gather / scatter-add, RL sequence assembly, masked cross-entropy, fused Adam,
// global-norm, mu-law tokenizer.  All are one-pass (or L2-resident re-read) streaming kernels
// with 16-byte vector accesses and wave-shuffle reductions; roofline = HBM bytes.
#include "db1_common.h"
#include "ln_row.h"
#ifndef DB1_ACT_UNROLL
#define DB1_ACT_UNROLL 2   /* rows in flight per thread of the activation kernels (tools/exp/act_unroll.sh) */
#endif

// =====================================================================================
// residual + LayerNorm      (reference: transformer_xl.py:231-238, 288-290)
// one wave per row, 4 rows per 256-thread block; the row (<= 8 KB) is re-read from L1/L2
// =====================================================================================
template <typename T, typename TP>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ r, float alpha,
                                                     const TP* __restrict__ gamma, const TP* __restrict__ beta,
                                                     T* __restrict__ y, T* s_out, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows, int d, float eps, Db1Drop drp) {
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    constexpr int V = Vec16<T>::N;
    const T* xr = x + row * d;
    const T* rr = r ? r + row * d : nullptr;
    auto load_s = [&](int i, Vec16<T>& a) {
        a.load(xr + i);
        if (rr) {
            Vec16<T> b;
            b.load(rr + i);
            if (drp.thr) db1_drop_apply<V>(drp, row * d + i, b.v);   // dropout on the sub-layer output, before the residual sum
#pragma unroll
            for (int j = 0; j < V; j++) a.v[j] = alpha * a.v[j] + b.v[j];
        } else {
#pragma unroll
            for (int j = 0; j < V; j++) a.v[j] = alpha * a.v[j];
        }
        if (sizeof(T) == 2) {  // s is a tensor of dtype T in the reference: round before the statistics
#pragma unroll
            for (int j = 0; j < V; j++) a.v[j] = bf2f(f2bf(a.v[j]));
        }
    };
    float sum = 0.f;
    for (int i = lane * V; i < d; i += 64 * V) {
        Vec16<T> a;
        load_s(i, a);
#pragma unroll
        for (int j = 0; j < V; j++) sum += a.v[j];
    }
    const float mu = wave_sum(sum) / (float)d;
    float sq = 0.f;
    for (int i = lane * V; i < d; i += 64 * V) {
        Vec16<T> a;
        load_s(i, a);
#pragma unroll
        for (int j = 0; j < V; j++) { float c = a.v[j] - mu; sq += c * c; }
    }
    const float rs = rsqrtf(wave_sum(sq) / (float)d + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    T* yr = y + row * d;
    T* sr = s_out ? s_out + row * d : nullptr;
    for (int i = lane * V; i < d; i += 64 * V) {
        Vec16<T> a, o;
        load_s(i, a);
        if (sr) a.store(sr + i);
#pragma unroll
        for (int j = 0; j < V; j++) o.v[j] = (a.v[j] - mu) * rs * ldf(gamma + i + j) + ldf(beta + i + j);
        o.store(yr + i);
    }
}

template <typename T, typename TP>
__global__ __launch_bounds__(256) void ln_bwd_ds_kernel(const T* __restrict__ dy, const T* __restrict__ s,
                                                        const TP* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, T* __restrict__ ds,
                                                        int64_t rows, int d, T* __restrict__ dr_out, Db1Drop drp) {
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    constexpr int V = Vec16<T>::N;
    const T* dyr = dy + row * d;
    const T* sr = s + row * d;
    const float mu = mean[row], rs = rstd[row];
    float c1 = 0.f, c2 = 0.f;
    for (int i = lane * V; i < d; i += 64 * V) {
        Vec16<T> a, b;
        a.load(dyr + i);
        b.load(sr + i);
#pragma unroll
        for (int j = 0; j < V; j++) {
            float g = a.v[j] * ldf(gamma + i + j);
            c1 += g;
            c2 += g * (b.v[j] - mu) * rs;
        }
    }
    c1 = wave_sum(c1) / (float)d;
    c2 = wave_sum(c2) / (float)d;
    T* dsr = ds + row * d;
    Db1Drop drow = drp;
    const int64_t ebase = db1_drop_row_base(drow, row, d);
    for (int i = lane * V; i < d; i += 64 * V) {
        Vec16<T> a, b, o;
        a.load(dyr + i);
        b.load(sr + i);
#pragma unroll
        for (int j = 0; j < V; j++) {
            float g = a.v[j] * ldf(gamma + i + j);
            float xh = (b.v[j] - mu) * rs;
            o.v[j] = rs * (g - c1 - xh * c2);
        }
        o.store(dsr + i);
        if (dr_out) {   // gradient of the dropped sub-layer output: the same keep decisions as the forward
            if (drp.thr) db1_drop_apply<V>(drow, ebase + i, o.v);
            o.store(dr_out + row * d + i);
        }
    }
}

// dgamma[c] += sum_r dy*xhat ; dbeta[c] += sum_r dy.  thread per column; ONE block walks all rows of its columns (generic / fp32 parity
// path: every accumulator has a single contributor, so the result does not depend on arrival order -- no float atomics)
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const T* __restrict__ dy, const T* __restrict__ s,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* dgamma, float* dbeta, int64_t rows, int d, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float ag = 0.f, ab = 0.f;
    for (int64_t r = r0; r < r1; r++) {
        float g = ldf(dy + r * d + c);
        float xh = (ldf(s + r * d + c) - mean[r]) * rstd[r];
        ag += g * xh;
        ab += g;
    }
    dgamma[c] += ag;
    dbeta[c] += ab;
}

// ---- register-resident variants for d == 64 * V * NV (d = 2048 in bf16: NV = 4): the row lives in registers, so the forward
// is ONE pass over x / r (the kernels above re-read the row from L2 three times) and the backward produces ds and the
// dgamma / dbeta partial sums of its rows from the same registers (the separate parameter pass re-read dy and s from HBM).
// Partials go to a workspace [block][2][d] and are added in block order by ln_param_reduce_kernel: no atomics, deterministic.
template <typename T, typename TP, int NV>
__global__ __launch_bounds__(256) void ln_fwd_reg_kernel(const T* __restrict__ x, const T* __restrict__ r, float alpha,
                                                         const TP* __restrict__ gamma, const TP* __restrict__ beta,
                                                         T* __restrict__ y, T* s_out, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int64_t rows, int d, float eps, Db1Drop drp) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    constexpr int V = Vec16<T>::N;
    Vec16<T> a[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) a[k].load(x + row * d + (k * 64 + lane) * V);
    if (r) {
#pragma unroll
        for (int k = 0; k < NV; k++) {
            Vec16<T> b;
            b.load(r + row * d + (k * 64 + lane) * V);
            if (drp.thr) db1_drop_apply<V>(drp, row * d + (k * 64 + lane) * V, b.v);   // dropout on the sub-layer output (:229, :269)
#pragma unroll
            for (int j = 0; j < V; j++) a[k].v[j] = alpha * a[k].v[j] + b.v[j];
        }
    } else {
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int j = 0; j < V; j++) a[k].v[j] = alpha * a[k].v[j];
    }
    float mu, rs;
    ln_row_stats<T, NV>(a, d, eps, mu, rs);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    if (s_out) {
#pragma unroll
        for (int k = 0; k < NV; k++) a[k].store(s_out + row * d + (k * 64 + lane) * V);
    }
    ln_row_store<T, TP, NV>(a, mu, rs, gamma, beta, y + row * d, lane);
}

template <typename T, typename TP, int NV>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ s, const TP* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ ds,
                                                           float* __restrict__ part, int64_t rows, int d, int rows_per_block,
                                                           T* __restrict__ dr_out, Db1Drop drp) {
    constexpr int V = Vec16<T>::N;
    __shared__ float red[2 * 64 * V * NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gam[NV][V], ag[NV][V], ab[NV][V];
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int j = 0; j < V; j++) { gam[k][j] = ldf(gamma + (k * 64 + lane) * V + j); ag[k][j] = 0.f; ab[k][j] = 0.f; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (int64_t row = r0 + wave; row < r1; row += 4) {
        Vec16<T> a[NV], b[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) { a[k].load(dy + row * d + (k * 64 + lane) * V); b[k].load(s + row * d + (k * 64 + lane) * V); }
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int j = 0; j < V; j++) {
                const float xh = (b[k].v[j] - mu) * rs, g = a[k].v[j] * gam[k][j];
                b[k].v[j] = xh;
                c1 += g;
                c2 += g * xh;
                ag[k][j] += a[k].v[j] * xh;
                ab[k][j] += a[k].v[j];
            }
        c1 = wave_sum(c1) / (float)d;
        c2 = wave_sum(c2) / (float)d;
        Db1Drop drow = drp;
        const int64_t ebase = db1_drop_row_base(drow, row, d);
#pragma unroll
        for (int k = 0; k < NV; k++) {
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < V; j++) o.v[j] = rs * (a[k].v[j] * gam[k][j] - c1 - b[k].v[j] * c2);
            o.store(ds + row * d + (k * 64 + lane) * V);
            if (dr_out) {   // gradient of the dropped sub-layer output: ds under the forward's keep decisions (regenerated, not stored)
                if (drp.thr) db1_drop_apply<V>(drow, ebase + (k * 64 + lane) * V, o.v);
                o.store(dr_out + row * d + (k * 64 + lane) * V);
            }
        }
    }
    if (!part) return;
    // the four waves add their column partials in wave order (fixed order: deterministic), then one row of the workspace is written
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < NV; k++)
#pragma unroll
                for (int j = 0; j < V; j++) {
                    const int c = (k * 64 + lane) * V + j;
                    if (w == 0) { red[c] = ag[k][j]; red[d + c] = ab[k][j]; }
                    else { red[c] += ag[k][j]; red[d + c] += ab[k][j]; }
                }
        }
        __syncthreads();
    }
    float* dst = part + (int64_t)blockIdx.x * 2 * d;
    for (int c = threadIdx.x; c < 2 * d; c += 256) dst[c] = red[c];
}
// 64 columns per workgroup; wave w adds the partial rows w, w+16, ... in order, then the 16 wave sums are added in wave order
// (4 waves per workgroup left each wave a chain of nblocks / 4 dependent loads: 35 us per call, latency-bound)
#define LNR_WAVES 16
__global__ __launch_bounds__(64 * LNR_WAVES) void ln_param_reduce_kernel(const float* __restrict__ part, float* dgamma, float* dbeta, int nblocks, int d) {
    __shared__ float red[LNR_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;  // < 2 * d (d is a multiple of 64 here)
    float acc = 0.f;
#pragma unroll 8
    for (int b = wave; b < nblocks; b += LNR_WAVES) acc += part[(int64_t)b * 2 * d + c];
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        float t = red[0][lane];
#pragma unroll
        for (int w = 1; w < LNR_WAVES; w++) t += red[w][lane];
        if (c < d) dgamma[c] += t; else dbeta[c - d] += t;
    }
}
template <typename T> static int ln_reg_nv(int d) {  // NV such that d == 64 * V * NV, or 0
    const int V = Vec16<T>::N;
    for (int nv = 1; nv <= 4; nv <<= 1) if (d == 64 * V * nv) return nv;
    return 0;
}

extern "C" int db1_layernorm_residual_fwd(const void* x, const void* r, float alpha, const void* gamma, const void* beta,
                                          void* y, void* s_out, float* mean, float* rstd, int64_t rows, int d, float eps,
                                          float drop_p, uint64_t drop_seed, uint32_t drop_site, uint32_t drop_step, const uint32_t* drop_step_dev,
                                          int dt, int dtParam, void* stream) {
    if (drop_p < 0.f || drop_p >= 1.f) DB1_FAIL(DB1_ERR_BAD_SHAPE, "layernorm fwd: dropout p=%g", (double)drop_p);
    if (drop_p > 0.f && !r) DB1_FAIL(DB1_ERR_BAD_SHAPE, "layernorm fwd: dropout applies to the residual input r, which is NULL");
    const Db1Drop drp = db1_drop_make(drop_p, drop_seed, drop_site, drop_step, drop_step_dev);
    if (!db1_dt_ok(dt) || !db1_dt_ok(dtParam)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "layernorm fwd: dtype");
    const int V = dt == DB1_F32 ? 4 : 8;
    if (rows <= 0 || d <= 0 || d % V) DB1_FAIL(DB1_ERR_BAD_SHAPE, "layernorm fwd: d=%d must be a multiple of %d", d, V);
    if (!db1_aligned16(x) || !db1_aligned16(y) || (r && !db1_aligned16(r)) || (s_out && !db1_aligned16(s_out)))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "layernorm fwd: pointers must be 16-byte aligned");
    dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (dt == DB1_BF16 && ln_reg_nv<bf16_t>(d)) {  // register-resident one-pass kernel
        const int nv = ln_reg_nv<bf16_t>(d);
#define LN_FWD_REG(TP, NV) ln_fwd_reg_kernel<bf16_t, TP, NV><<<grid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)r, alpha, (const TP*)gamma, (const TP*)beta, (bf16_t*)y, (bf16_t*)s_out, mean, rstd, rows, d, eps, drp)
#define LN_FWD_NV(TP) do { if (nv == 1) LN_FWD_REG(TP, 1); else if (nv == 2) LN_FWD_REG(TP, 2); else LN_FWD_REG(TP, 4); } while (0)
        if (dtParam == DB1_BF16) LN_FWD_NV(bf16_t); else LN_FWD_NV(float);
#undef LN_FWD_NV
#undef LN_FWD_REG
        DB1_CHECK_LAUNCH("layernorm fwd (registers)");
        return DB1_OK;
    }
#define LN_FWD(T, TP) ln_fwd_kernel<T, TP><<<grid, 256, 0, st>>>((const T*)x, (const T*)r, alpha, (const TP*)gamma, (const TP*)beta, (T*)y, (T*)s_out, mean, rstd, rows, d, eps, drp)
    if (dt == DB1_F32 && dtParam == DB1_F32) LN_FWD(float, float);
    else if (dt == DB1_BF16 && dtParam == DB1_BF16) LN_FWD(bf16_t, bf16_t);
    else if (dt == DB1_BF16 && dtParam == DB1_F32) LN_FWD(bf16_t, float);
    else LN_FWD(float, bf16_t);
#undef LN_FWD
    DB1_CHECK_LAUNCH("layernorm fwd");
    return DB1_OK;
}

// rows per block of the fused backward = rows per partial-sum slot: 32 for the benchmark's 65 536 rows (2048 blocks); fewer for the
// reference's micro-batch of 4 sequences (4096 rows gave 128 blocks: half the CUs, 40 us for a 10 us stream) so that >= 512 blocks exist
static int ln_bwd_rpb(int64_t rows) {
    int rpb = 32;
    const int want = 512;     // (256 / 512 / 1024 measured at 4096 rows, profiles/r05e_*: 512)
    while (rpb > 4 && (rows + rpb - 1) / rpb < want) rpb >>= 1;
    return rpb;
}
extern "C" int64_t db1_layernorm_residual_bwd_workspace_bytes(int64_t rows, int d, int dt) {
    if (dt != DB1_BF16 || !ln_reg_nv<bf16_t>(d)) return 0;  // the generic kernels accumulate the parameter gradients directly
    const int rpb = ln_bwd_rpb(rows);
    return ((rows + rpb - 1) / rpb) * 2 * (int64_t)d * (int64_t)sizeof(float);
}
static int ln_bwd_impl(const void* dy, const void* s, const void* gamma, const float* mean, const float* rstd,
                       void* ds, void* dr_out, float* dgamma_acc, float* dbeta_acc, int64_t rows, int d,
                       float drop_p, uint64_t drop_seed, uint32_t drop_site, uint32_t drop_step, const uint32_t* drop_step_dev, int64_t drop_rows_per_step,
                       int dt, int dtParam, void* ws_, int64_t ws_bytes, void* stream, bool parts_only) {
    if (drop_p < 0.f || drop_p >= 1.f) DB1_FAIL(DB1_ERR_BAD_SHAPE, "layernorm bwd: dropout p=%g", (double)drop_p);
    if (dr_out && !db1_aligned16(dr_out)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "layernorm bwd: dr_out alignment");
    if (drop_rows_per_step < 0 || drop_rows_per_step > 0x7fffffff || rows > 0x7fffffff) DB1_FAIL(DB1_ERR_BAD_SHAPE, "layernorm bwd: rows / drop_rows_per_step");
    Db1Drop drp = db1_drop_make(drop_p, drop_seed, drop_site, drop_step, drop_step_dev);
    drp.rows_per_step = drop_rows_per_step;
    if (!db1_dt_ok(dt) || !db1_dt_ok(dtParam)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "layernorm bwd: dtype");
    const int V = dt == DB1_F32 ? 4 : 8;
    if (rows <= 0 || d <= 0 || d % V) DB1_FAIL(DB1_ERR_BAD_SHAPE, "layernorm bwd: d=%d must be a multiple of %d", d, V);
    if (!db1_aligned16(dy) || !db1_aligned16(s) || !db1_aligned16(ds)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "layernorm bwd: alignment");
    hipStream_t st = (hipStream_t)stream;
    if (dt == DB1_BF16 && ln_reg_nv<bf16_t>(d)) {  // fused one-pass backward (ds + parameter partials)
        const int nv = ln_reg_nv<bf16_t>(d), rpb = ln_bwd_rpb(rows);
        const int nblocks = (int)((rows + rpb - 1) / rpb);
        const bool params = parts_only || (dgamma_acc && dbeta_acc);
        float* ws = nullptr;
        if (params) {
            DB1_NEED_WS(ws_, ws_bytes, db1_layernorm_residual_bwd_workspace_bytes(rows, d, dt), "layernorm bwd");
            ws = (float*)ws_;
        }
#define LN_BWD_F(TP, NV) ln_bwd_fused_kernel<bf16_t, TP, NV><<<nblocks, 256, 0, st>>>((const bf16_t*)dy, (const bf16_t*)s, (const TP*)gamma, mean, rstd, (bf16_t*)ds, ws, rows, d, rpb, (bf16_t*)dr_out, drp)
#define LN_BWD_NV(TP) do { if (nv == 1) LN_BWD_F(TP, 1); else if (nv == 2) LN_BWD_F(TP, 2); else LN_BWD_F(TP, 4); } while (0)
        if (dtParam == DB1_BF16) LN_BWD_NV(bf16_t); else LN_BWD_NV(float);
#undef LN_BWD_NV
#undef LN_BWD_F
        DB1_CHECK_LAUNCH("layernorm bwd (fused)");
        if (params && !parts_only) {
            ln_param_reduce_kernel<<<2 * d / 64, 64 * LNR_WAVES, 0, st>>>(ws, dgamma_acc, dbeta_acc, nblocks, d);
            DB1_CHECK_LAUNCH("layernorm bwd param reduce");
        }
        return DB1_OK;
    }
    if (parts_only) DB1_FAIL(DB1_ERR_UNSUPPORTED, "layernorm bwd (partials only): the register-resident bf16 kernel only (db1_layernorm_residual_bwd_workspace_bytes > 0)");
    dim3 grid((unsigned)((rows + 3) / 4));
#define LN_BWD(T, TP) ln_bwd_ds_kernel<T, TP><<<grid, 256, 0, st>>>((const T*)dy, (const T*)s, (const TP*)gamma, mean, rstd, (T*)ds, rows, d, (T*)dr_out, drp)
    if (dt == DB1_F32 && dtParam == DB1_F32) LN_BWD(float, float);
    else if (dt == DB1_BF16 && dtParam == DB1_BF16) LN_BWD(bf16_t, bf16_t);
    else if (dt == DB1_BF16 && dtParam == DB1_F32) LN_BWD(bf16_t, float);
    else LN_BWD(float, bf16_t);
#undef LN_BWD
    DB1_CHECK_LAUNCH("layernorm bwd ds");
    if (dgamma_acc && dbeta_acc) {
        const int rpb = (int)rows;
        dim3 g2((unsigned)((d + 255) / 256), 1u);
        if (dt == DB1_F32) ln_bwd_param_kernel<float><<<g2, 256, 0, st>>>((const float*)dy, (const float*)s, mean, rstd, dgamma_acc, dbeta_acc, rows, d, rpb);
        else ln_bwd_param_kernel<bf16_t><<<g2, 256, 0, st>>>((const bf16_t*)dy, (const bf16_t*)s, mean, rstd, dgamma_acc, dbeta_acc, rows, d, rpb);
        DB1_CHECK_LAUNCH("layernorm bwd param");
    }
    return DB1_OK;
}

extern "C" int db1_layernorm_residual_bwd(const void* dy, const void* s, const void* gamma, const float* mean, const float* rstd,
                                          void* ds, void* dr_out, float* dgamma_acc, float* dbeta_acc, int64_t rows, int d,
                                          float drop_p, uint64_t drop_seed, uint32_t drop_site, uint32_t drop_step, const uint32_t* drop_step_dev,
                                          int64_t drop_rows_per_step, int dt, int dtParam, void* ws_, int64_t ws_bytes, void* stream) {
    return ln_bwd_impl(dy, s, gamma, mean, rstd, ds, dr_out, dgamma_acc, dbeta_acc, rows, d, drop_p, drop_seed, drop_site, drop_step, drop_step_dev, drop_rows_per_step,
                       dt, dtParam, ws_, ws_bytes, stream, false);
}
// the same launch WITHOUT the parameter reduce: the per-block partial sums [blocks][2][d] (float32, db1_layernorm_residual_bwd_workspace_bytes)
// stay in `parts` for the caller to add up later -- db1_colsum_acc over the [blocks, 2 d] matrix gives (dgamma | dbeta).  Gradient
// accumulation keeps the partials of every micro-step and reduces once per optimizer step (one launch instead of one per micro-step).
extern "C" int db1_layernorm_residual_bwd_parts(const void* dy, const void* s, const void* gamma, const float* mean, const float* rstd,
                                                void* ds, void* dr_out, float* parts, int64_t parts_bytes, int64_t rows, int d,
                                                float drop_p, uint64_t drop_seed, uint32_t drop_site, uint32_t drop_step, const uint32_t* drop_step_dev,
                                                int64_t drop_rows_per_step, int dt, int dtParam, void* stream) {
    return ln_bwd_impl(dy, s, gamma, mean, rstd, ds, dr_out, nullptr, nullptr, rows, d, drop_p, drop_seed, drop_site, drop_step, drop_step_dev, drop_rows_per_step,
                       dt, dtParam, parts, parts_bytes, stream, true);
}

// =====================================================================================
// feed-forward activation (GEGLU = a * gelu_erf(b), activations.py:19-32)
// =====================================================================================
// thread = one 16-byte column vector, block = 256 vectors x a chunk of rows (two rows in flight per thread; no index division)
template <typename T, int ACT>
__global__ __launch_bounds__(256) void act_fwd_kernel(const T* __restrict__ z, T* __restrict__ out, int64_t rows, int n, int rows_per_chunk) {
    constexpr int V = Vec16<T>::N;
    const int ld = ACT == DB1_ACT_GEGLU ? 2 * n : n;
    const int c = (blockIdx.x * 256 + threadIdx.x) * V;
    if (c >= n) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
#pragma unroll DB1_ACT_UNROLL
    for (int64_t r = r0; r < r1; r++) {
        Vec16<T> a, o;
        a.load(z + r * ld + c);
        if (ACT == DB1_ACT_GEGLU) {
            Vec16<T> b;
            b.load(z + r * ld + n + c);
#pragma unroll
            for (int j = 0; j < V; j++) o.v[j] = a.v[j] * gelu_fwd_t<T>(b.v[j]);
        } else if (ACT == DB1_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < V; j++) o.v[j] = gelu_fwd_t<T>(a.v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < V; j++) o.v[j] = fmaxf(a.v[j], 0.f);
        }
        o.store(out + r * n + c);
    }
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ z, const T* __restrict__ dout, T* __restrict__ dz,
                                                      int64_t rows, int n) {
    constexpr int V = Vec16<T>::N;
    const int64_t nv = (int64_t)rows * (n / V);
    const int ld = ACT == DB1_ACT_GEGLU ? 2 * n : n;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nv; idx += (int64_t)gridDim.x * 256) {
        int64_t r = idx / (n / V);
        int c = (int)(idx % (n / V)) * V;
        Vec16<T> a, g, o;
        a.load(z + r * ld + c);
        g.load(dout + r * n + c);
        if (ACT == DB1_ACT_GEGLU) {
            Vec16<T> b, o2;
            b.load(z + r * ld + n + c);
#pragma unroll
            for (int j = 0; j < V; j++) { float y, dy; gelu_both_t<T>(b.v[j], y, dy); o.v[j] = g.v[j] * y; o2.v[j] = g.v[j] * a.v[j] * dy; }
            o.store(dz + r * ld + c);
            o2.store(dz + r * ld + n + c);
        } else {
            if (ACT == DB1_ACT_GELU) {
#pragma unroll
                for (int j = 0; j < V; j++) { float y, dy; gelu_both_t<T>(a.v[j], y, dy); o.v[j] = g.v[j] * dy; }
            } else {
#pragma unroll
                for (int j = 0; j < V; j++) o.v[j] = a.v[j] > 0.f ? g.v[j] : 0.f;
            }
            o.store(dz + r * ld + c);
        }
    }
}

// Activation backward that also produces the column sums of dz (the gradient of the first feed-forward bias) from the values
// it stores: thread = one 16-byte column vector (both GEGLU halves), block = 256 vectors x a chunk of rows; per-chunk partial
// sums go to a workspace [chunk][ld] and are added in chunk order by colsum_part_reduce_kernel (deterministic, no atomics).
// This removes a separate full read of dz (the widest activation of the layer: 8192 columns at d = 2048).
template <typename T, int ACT>
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(const T* __restrict__ z, const T* __restrict__ dout, T* __restrict__ dz,
                                                           float* __restrict__ part, int64_t rows, int n, int rows_per_chunk) {
    constexpr int V = Vec16<T>::N;
    const int ld = ACT == DB1_ACT_GEGLU ? 2 * n : n;
    const int c = (blockIdx.x * 256 + threadIdx.x) * V;
    if (c >= n) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    float s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; j++) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll DB1_ACT_UNROLL
    for (int64_t r = r0; r < r1; r++) {
        Vec16<T> a, g, o;
        a.load(z + r * ld + c);
        g.load(dout + r * n + c);
        if (ACT == DB1_ACT_GEGLU) {
            Vec16<T> b, o2;
            b.load(z + r * ld + n + c);
#pragma unroll
            for (int j = 0; j < V; j++) { float y, dy; gelu_both_t<T>(b.v[j], y, dy); o.v[j] = g.v[j] * y; o2.v[j] = g.v[j] * a.v[j] * dy; }
            o.store(dz + r * ld + c);
            o2.store(dz + r * ld + n + c);
#pragma unroll
            for (int j = 0; j < V; j++) {  // sum what was stored (dz is a tensor of dtype T in the reference)
                s1[j] += sizeof(T) == 2 ? bf2f(f2bf(o.v[j])) : o.v[j];
                s2[j] += sizeof(T) == 2 ? bf2f(f2bf(o2.v[j])) : o2.v[j];
            }
        } else {
            if (ACT == DB1_ACT_GELU) {
#pragma unroll
                for (int j = 0; j < V; j++) { float y, dy; gelu_both_t<T>(a.v[j], y, dy); o.v[j] = g.v[j] * dy; }
            } else {
#pragma unroll
                for (int j = 0; j < V; j++) o.v[j] = a.v[j] > 0.f ? g.v[j] : 0.f;
            }
            o.store(dz + r * ld + c);
#pragma unroll
            for (int j = 0; j < V; j++) s1[j] += sizeof(T) == 2 ? bf2f(f2bf(o.v[j])) : o.v[j];
        }
    }
    float* dst = part + (int64_t)blockIdx.y * ld + c;
#pragma unroll
    for (int j = 0; j < V; j += 4) *reinterpret_cast<float4*>(dst + j) = make_float4(s1[j], s1[j + 1], s1[j + 2], s1[j + 3]);
    if (ACT == DB1_ACT_GEGLU) {
#pragma unroll
        for (int j = 0; j < V; j += 4) *reinterpret_cast<float4*>(dst + n + j) = make_float4(s2[j], s2[j + 1], s2[j + 2], s2[j + 3]);
    }
}
// out[c] += sum over chunks of part[chunk][c]; 64 columns per workgroup, 16 waves: wave w takes chunks w, w+16, ... in order and the 16
// partial sums are added in wave order (deterministic).  (With 4 waves per workgroup the 32 workgroups of a 2048-column sum walked
// 512 chunks each, one dependent load chain per wave: 31 us, latency-bound.)
#define CSR_WAVES 16
__device__ __forceinline__ void colsum_part_reduce_body(const float* __restrict__ part, float* out, int nchunks, int cols, int64_t pitch) {
    __shared__ float red[CSR_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (c < cols) {
#pragma unroll 8
        for (int b = wave; b < nchunks; b += CSR_WAVES) acc += part[(int64_t)b * pitch + c];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < cols) {
        float s = red[0][lane];
#pragma unroll
        for (int w = 1; w < CSR_WAVES; w++) s += red[w][lane];
        out[c] += s;
    }
}
__global__ __launch_bounds__(64 * CSR_WAVES) void colsum_part_reduce_kernel(const float* __restrict__ part, float* out, int nchunks, int cols) {
    colsum_part_reduce_body(part, out, nchunks, cols, cols);
}
// partial rows `pitch` floats apart (several sums interleaved per chunk)
__global__ __launch_bounds__(64 * CSR_WAVES) void colsum_part_reduce_strided_kernel(const float* __restrict__ part, float* out, int nchunks, int cols, int pitch) {
    colsum_part_reduce_body(part, out, nchunks, cols, pitch);
}

int db1_colsum_part_reduce_launch(const float* part, float* out, int nchunks, int cols, hipStream_t st) {   // (gemm_geglu.hip)
    colsum_part_reduce_kernel<<<(cols + 63) / 64, 64 * CSR_WAVES, 0, st>>>(part, out, nchunks, cols);
    DB1_CHECK_LAUNCH("colsum_part_reduce");
    return DB1_OK;
}

static inline unsigned grid_for(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;  // 16 blocks per CU, grid-stride beyond
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int db1_ffn_act_fwd(const void* z, void* out, int64_t rows, int n, int act, int dt, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "ffn_act_fwd: dtype");
    const int V = dt == DB1_F32 ? 4 : 8;
    if (rows <= 0 || n <= 0 || n % V) DB1_FAIL(DB1_ERR_BAD_SHAPE, "ffn_act_fwd: n=%d must be a multiple of %d", n, V);
    if (act < 0 || act > 2) DB1_FAIL(DB1_ERR_UNSUPPORTED, "ffn_act_fwd: act %d", act);
    hipStream_t st = (hipStream_t)stream;
    int rpc = rows >= 32768 ? 64 : (rows >= 4096 ? 32 : 8);
    while ((rows + rpc - 1) / rpc > 65535) rpc *= 2;
    dim3 g((unsigned)((n / V + 255) / 256), (unsigned)((rows + rpc - 1) / rpc));
#define L(T, A) act_fwd_kernel<T, A><<<g, 256, 0, st>>>((const T*)z, (T*)out, rows, n, rpc)
    DB1_DISPATCH_DT(dt, T, { if (act == 0) L(T, 0); else if (act == 1) L(T, 1); else L(T, 2); });
#undef L
    DB1_CHECK_LAUNCH("ffn_act_fwd");
    return DB1_OK;
}

extern "C" int db1_ffn_act_bwd(const void* z, const void* dout, void* dz, int64_t rows, int n, int act, int dt, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "ffn_act_bwd: dtype");
    const int V = dt == DB1_F32 ? 4 : 8;
    if (rows <= 0 || n <= 0 || n % V) DB1_FAIL(DB1_ERR_BAD_SHAPE, "ffn_act_bwd: n=%d must be a multiple of %d", n, V);
    if (act < 0 || act > 2) DB1_FAIL(DB1_ERR_UNSUPPORTED, "ffn_act_bwd: act %d", act);
    hipStream_t st = (hipStream_t)stream;
    unsigned g = grid_for(rows * (n / V));
#define L(T, A) act_bwd_kernel<T, A><<<g, 256, 0, st>>>((const T*)z, (const T*)dout, (T*)dz, rows, n)
    DB1_DISPATCH_DT(dt, T, { if (act == 0) L(T, 0); else if (act == 1) L(T, 1); else L(T, 2); });
#undef L
    DB1_CHECK_LAUNCH("ffn_act_bwd");
    return DB1_OK;
}

static inline int act_bias_rpc(int64_t rows) { return rows >= 32768 ? 64 : 32; }  // (8 workgroups per CU are enough to stream)
extern "C" int64_t db1_ffn_act_bwd_bias_workspace_bytes(int64_t rows, int n, int act) {
    const int rpc = act_bias_rpc(rows);
    return ((rows + rpc - 1) / rpc) * (int64_t)(act == DB1_ACT_GEGLU ? 2 * n : n) * (int64_t)sizeof(float);
}
extern "C" int db1_ffn_act_bwd_bias(const void* z, const void* dout, void* dz, float* dbias_acc, int64_t rows, int n, int act, int dt,
                                    void* ws_, int64_t ws_bytes, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "ffn_act_bwd_bias: dtype");
    const int V = dt == DB1_F32 ? 4 : 8;
    if (rows <= 0 || n <= 0 || n % V) DB1_FAIL(DB1_ERR_BAD_SHAPE, "ffn_act_bwd_bias: n=%d must be a multiple of %d", n, V);
    if (act < 0 || act > 2) DB1_FAIL(DB1_ERR_UNSUPPORTED, "ffn_act_bwd_bias: act %d", act);
    if (!dbias_acc) DB1_FAIL(DB1_ERR_BAD_SHAPE, "ffn_act_bwd_bias: null accumulator");
    hipStream_t st = (hipStream_t)stream;
    const int ld = act == DB1_ACT_GEGLU ? 2 * n : n;
    const int rpc = act_bias_rpc(rows), nchunks = (int)((rows + rpc - 1) / rpc);
    DB1_NEED_WS(ws_, ws_bytes, db1_ffn_act_bwd_bias_workspace_bytes(rows, n, act), "ffn_act_bwd_bias");
    float* ws = (float*)ws_;
    dim3 g((unsigned)((n / V + 255) / 256), (unsigned)nchunks);
#define L(T, A) act_bwd_bias_kernel<T, A><<<g, 256, 0, st>>>((const T*)z, (const T*)dout, (T*)dz, ws, rows, n, rpc)
    DB1_DISPATCH_DT(dt, T, { if (act == 0) L(T, 0); else if (act == 1) L(T, 1); else L(T, 2); });
#undef L
    DB1_CHECK_LAUNCH("ffn_act_bwd_bias");
    colsum_part_reduce_kernel<<<(ld + 63) / 64, 64 * CSR_WAVES, 0, st>>>(ws, dbias_acc, nchunks, ld);
    DB1_CHECK_LAUNCH("ffn_act_bwd_bias reduce");
    return DB1_OK;
}

// =====================================================================================
// column sums, add, cast
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* out, int64_t rows, int cols, int64_t ldx, int rpb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    int64_t r0 = (int64_t)blockIdx.y * rpb;
    int64_t r1 = r0 + rpb < rows ? r0 + rpb : rows;
    float a = 0.f;
    for (int64_t r = r0; r < r1; r++) a += ldf(x + r * ldx + c);
    out[c] += a;      // (launched with ONE row block: single contributor)
}

// vectorised form: a lane owns one 16-byte column group (4 f32 / 8 bf16), a wave reads 1 KiB of one row per load, the 4 waves of a
// block take alternating rows; partial sums are combined through LDS and added with one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ x, float* out, int64_t rows, int cols, int64_t ldx, int rpb) {
    constexpr int V = Vec16<T>::N;
    __shared__ float part[4][64 * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * V;
    const int64_t r0 = (int64_t)blockIdx.y * rpb;
    const int64_t r1 = r0 + rpb < rows ? r0 + rpb : rows;
    float a[V];
#pragma unroll
    for (int j = 0; j < V; j++) a[j] = 0.f;
    if (c0 < cols) {
        for (int64_t r = r0 + w; r < r1; r += 4) {
            Vec16<T> v;
            v.load(x + r * ldx + c0);
#pragma unroll
            for (int j = 0; j < V; j++) a[j] += v.v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < V; j++) part[w][lane * V + j] = a[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * V; i += 256) {
        const int c = blockIdx.x * 64 * V + i;
        if (c < cols) out[c] += part[0][i] + part[1][i] + part[2][i] + part[3][i];   // (launched with ONE row block per column group)
    }
}

// deterministic variant: per-chunk partial sums (4 waves added in wave order) -> workspace [chunk][cols] -> colsum_part_reduce_kernel
template <typename T>
__global__ __launch_bounds__(256) void colsum_chunk_kernel(const T* __restrict__ x, float* __restrict__ part, int64_t rows, int cols, int64_t ldx, int rpc) {
    constexpr int V = Vec16<T>::N;
    __shared__ float red[4][64 * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * V;
    const int64_t r0 = (int64_t)blockIdx.y * rpc;
    const int64_t r1 = r0 + rpc < rows ? r0 + rpc : rows;
    float a[V];
#pragma unroll
    for (int j = 0; j < V; j++) a[j] = 0.f;
    if (c0 < cols) {
#pragma unroll 8
        for (int64_t r = r0 + w; r < r1; r += 4) {
            Vec16<T> v;
            v.load(x + r * ldx + c0);
#pragma unroll
            for (int j = 0; j < V; j++) a[j] += v.v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < V; j++) red[w][lane * V + j] = a[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * V; i += 256) {
        const int c = blockIdx.x * 64 * V + i;
        if (c < cols) part[(int64_t)blockIdx.y * cols + c] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    }
}

static inline int colsum_rpc(int64_t rows) {  // the ordered reduce walks the chunks serially: keep them few (conv bias sums: 3.9 M rows)
    int rpc = 32;
    while (rows / rpc > 1024) rpc *= 2;
    return rpc;
}
extern "C" int64_t db1_colsum_acc_workspace_bytes(int64_t rows, int cols) {
    if (rows < 1024) return 0;  // short inputs add straight into the accumulator
    const int rpc = colsum_rpc(rows);
    return ((rows + rpc - 1) / rpc) * (int64_t)cols * (int64_t)sizeof(float);
}
extern "C" int db1_colsum_acc(const void* x, float* out_acc, int64_t rows, int cols, int64_t ldx, int dt, void* ws_, int64_t ws_bytes,
                              void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "colsum: dtype");
    if (rows <= 0 || cols <= 0 || ldx < cols) DB1_FAIL(DB1_ERR_BAD_SHAPE, "colsum: shape");
    const int V = dt == DB1_F32 ? 4 : 8;
    if (cols % V == 0 && ldx % V == 0 && db1_aligned16(x)) {
        if (rows >= 1024) {  // long inputs: chunk partials + ordered reduce (deterministic, ~16 resident waves per CU)
            const int rpc = colsum_rpc(rows);
            const int nchunks = (int)((rows + rpc - 1) / rpc);
            DB1_NEED_WS(ws_, ws_bytes, db1_colsum_acc_workspace_bytes(rows, cols), "colsum");
            float* ws = (float*)ws_;
            dim3 gc((unsigned)((cols / V + 63) / 64), (unsigned)nchunks);
            DB1_DISPATCH_DT(dt, T, (colsum_chunk_kernel<T><<<gc, 256, 0, (hipStream_t)stream>>>((const T*)x, ws, rows, cols, ldx, rpc)));
            DB1_CHECK_LAUNCH("colsum_chunk");
            colsum_part_reduce_kernel<<<(cols + 63) / 64, 64 * CSR_WAVES, 0, (hipStream_t)stream>>>(ws, out_acc, nchunks, cols);
            DB1_CHECK_LAUNCH("colsum reduce");
            return DB1_OK;
        }
        const int rpbv = (int)rows;   // short inputs (< 1024 rows): ONE workgroup per 64 column vectors walks all rows, so every accumulator has a single
        dim3 gv((unsigned)((cols / V + 63) / 64), 1u);   // contributor and the result does not depend on arrival order (it used to be 4 row blocks + atomics)
        DB1_DISPATCH_DT(dt, T, (colsum_vec_kernel<T><<<gv, 256, 0, (hipStream_t)stream>>>((const T*)x, out_acc, rows, cols, ldx, rpbv)));
        DB1_CHECK_LAUNCH("colsum_vec");
        return DB1_OK;
    }
    const int rpb = (int)rows;   // unaligned / odd shapes (fp32 parity path): one block per 256 columns walks all rows -- deterministic, not fast
    dim3 g((unsigned)((cols + 255) / 256), 1u);
    DB1_DISPATCH_DT(dt, T, (colsum_kernel<T><<<g, 256, 0, (hipStream_t)stream>>>((const T*)x, out_acc, rows, cols, ldx, rpb)));
    DB1_CHECK_LAUNCH("colsum");
    return DB1_OK;
}


template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) stf(y + i, ldf(a + i) + ldf(b + i));
}
// (16-byte vectors: the element-at-a-time form above moved the RL step's 246 MB tensors at 1.5 TB/s, 487 us per call; y may alias a or b)
template <typename T>
__global__ __launch_bounds__(256) void add_vec_kernel(const T* a, const T* b, T* y, int64_t nvec) {
    constexpr int V = Vec16<T>::N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        Vec16<T> x, z;
        x.load(a + i * V);
        z.load(b + i * V);
#pragma unroll
        for (int j = 0; j < V; j++) x.v[j] += z.v[j];
        x.store(y + i * V);
    }
}
// out[t, :] = out[t, :] + row_table[row_ids[t], :] + col_table[col_ids[t], :]  (the image-patch embedder's position term,
// vision_embedding.py:117-180, onto the patch embeddings in one pass: it was two gathers into a scratch tensor and two adds, 1.2 ms of
// element-wise traffic per RL step).  Sum order (emb + row) + col in fp32, ONE rounding to the output dtype.
template <typename TT, typename T>
__global__ __launch_bounds__(256) void vision_pos_add_kernel(T* __restrict__ out, const TT* __restrict__ row_table, const TT* __restrict__ col_table,
                                                             const int64_t* __restrict__ row_ids, const int64_t* __restrict__ col_ids, int64_t n, int d,
                                                             int64_t n_rows) {
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t ri = row_ids[t], ci = col_ids[t];
    const bool rok = ri >= 0 && ri < n_rows, cok = ci >= 0 && ci < n_rows;      // (ids outside the table add nothing, as the gather gave zeros)
    const TT* rr = row_table + (rok ? ri : 0) * d;
    const TT* cr = col_table + (cok ? ci : 0) * d;
    T* o = out + t * d;
    constexpr int V = Vec16<T>::N;
    if (sizeof(TT) == sizeof(T) && (d % V) == 0) {
        for (int i = lane * V; i < d; i += 64 * V) {
            Vec16<T> x, a, b;
            x.load(o + i);
            a.load(reinterpret_cast<const T*>(rr) + i);
            b.load(reinterpret_cast<const T*>(cr) + i);
#pragma unroll
            for (int j = 0; j < V; j++) x.v[j] = (x.v[j] + (rok ? a.v[j] : 0.f)) + (cok ? b.v[j] : 0.f);
            x.store(o + i);
        }
        return;
    }
    for (int i = lane; i < d; i += 64) stf(o + i, (ldf(o + i) + (rok ? ldf(rr + i) : 0.f)) + (cok ? ldf(cr + i) : 0.f));
}
extern "C" int db1_vision_pos_add(void* out, const void* row_table, const void* col_table, const int64_t* row_ids, const int64_t* col_ids, int64_t n,
                                  int d, int64_t n_table_rows, int dtTable, int dt, void* stream) {
    if (!db1_dt_ok(dt) || !db1_dt_ok(dtTable)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "vision_pos_add: dtype");
    if (n <= 0 || d <= 0 || n_table_rows <= 0 || !out || !row_table || !col_table || !row_ids || !col_ids) DB1_FAIL(DB1_ERR_BAD_SHAPE, "vision_pos_add: shape / null buffer");
    if (!db1_aligned16(out) || !db1_aligned16(row_table) || !db1_aligned16(col_table)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "vision_pos_add: alignment");
    hipStream_t st = (hipStream_t)stream;
    dim3 g((unsigned)((n + 3) / 4));
#define L(A, B) vision_pos_add_kernel<A, B><<<g, 256, 0, st>>>((B*)out, (const A*)row_table, (const A*)col_table, row_ids, col_ids, n, d, n_table_rows)
    if (dtTable == DB1_F32 && dt == DB1_F32) L(float, float);
    else if (dtTable == DB1_BF16 && dt == DB1_BF16) L(bf16_t, bf16_t);
    else if (dtTable == DB1_F32) L(float, bf16_t);
    else L(bf16_t, float);
#undef L
    DB1_CHECK_LAUNCH("vision_pos_add");
    return DB1_OK;
}
extern "C" int db1_add(const void* a, const void* b, void* y, int64_t n, int dt, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "add: dtype");
    if (n <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "add: n");
    const int V = dt == DB1_F32 ? 4 : 8;
    if ((n % V) == 0 && db1_aligned16(a) && db1_aligned16(b) && db1_aligned16(y)) {
        DB1_DISPATCH_DT(dt, T, (add_vec_kernel<T><<<grid_for(n / V), 256, 0, (hipStream_t)stream>>>((const T*)a, (const T*)b, (T*)y, n / V)));
        DB1_CHECK_LAUNCH("add");
        return DB1_OK;
    }
    DB1_DISPATCH_DT(dt, T, (add_kernel<T><<<grid_for(n), 256, 0, (hipStream_t)stream>>>((const T*)a, (const T*)b, (T*)y, n)));
    DB1_CHECK_LAUNCH("add");
    return DB1_OK;
}

template <typename TA, typename T>
__global__ __launch_bounds__(256) void add2d_kernel(const TA* __restrict__ a, int64_t lda, const T* b, int64_t ldb, T* y, int64_t ldy,
                                                    int64_t rows, int cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i % cols);
        stf(y + r * ldy + c, ldf(a + r * lda + c) + ldf(b + r * ldb + c));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void add2d_vec_kernel(const T* __restrict__ a, int64_t lda, const T* b, int64_t ldb, T* y, int64_t ldy,
                                                        int64_t rows, int cols) {
    constexpr int V = Vec16<T>::N;
    const int vpr = cols / V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * vpr; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / vpr;
        const int c = (int)(i - r * vpr) * V;
        Vec16<T> p, q, o;
        p.load(a + r * lda + c);
        q.load(b + r * ldb + c);
#pragma unroll
        for (int j = 0; j < V; j++) o.v[j] = p.v[j] + q.v[j];
        o.store(y + r * ldy + c);
    }
}
extern "C" int db1_add2d(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int64_t rows, int cols, int dtA,
                         int dt, void* stream) {
    if (!db1_dt_ok(dt) || !db1_dt_ok(dtA)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "add2d: dtype");
    if (rows <= 0 || cols <= 0 || lda < cols || ldb < cols || ldy < cols) DB1_FAIL(DB1_ERR_BAD_SHAPE, "add2d: shape");
    hipStream_t st = (hipStream_t)stream;
    const int Vv = dt == DB1_F32 ? 4 : 8;
    if (dtA == dt && cols % Vv == 0 && lda % Vv == 0 && ldb % Vv == 0 && ldy % Vv == 0 && db1_aligned16(a) && db1_aligned16(b) && db1_aligned16(y)) {
        unsigned gv = grid_for(rows * (cols / Vv));
        DB1_DISPATCH_DT(dt, T, (add2d_vec_kernel<T><<<gv, 256, 0, st>>>((const T*)a, lda, (const T*)b, ldb, (T*)y, ldy, rows, cols)));
        DB1_CHECK_LAUNCH("add2d (vec)");
        return DB1_OK;
    }
    unsigned g = grid_for(rows * cols);
#define L_(TA, T) add2d_kernel<TA, T><<<g, 256, 0, st>>>((const TA*)a, lda, (const T*)b, ldb, (T*)y, ldy, rows, cols)
    if (dtA == DB1_F32 && dt == DB1_F32) L_(float, float);
    else if (dtA == DB1_BF16 && dt == DB1_BF16) L_(bf16_t, bf16_t);
    else if (dtA == DB1_F32) L_(float, bf16_t);
    else L_(bf16_t, float);
#undef L_
    DB1_CHECK_LAUNCH("add2d");
    return DB1_OK;
}

// y = a + b together with the column sums of a and of b (the attention backward: dq = dq_k + dq_r, du = colsum(dq_k),
// dv_bias = colsum(dq_r) -- three passes over the same two matrices before).  thread = one 16-byte column vector, block = 64 vectors x 4
// waves over a chunk of rows; per-chunk partials [chunk][2][cols] are added in chunk order by colsum_part_reduce_kernel (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void add2d_colsums_kernel(const T* __restrict__ a, int64_t lda, const T* b, int64_t ldb, T* y, int64_t ldy,
                                                            float* __restrict__ part, int64_t rows, int cols, int rpc) {
    constexpr int V = Vec16<T>::N;
    __shared__ float red[4][2][64 * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * V;
    const int64_t r0 = (int64_t)blockIdx.y * rpc;
    const int64_t r1 = r0 + rpc < rows ? r0 + rpc : rows;
    float sa[V], sb[V];
#pragma unroll
    for (int j = 0; j < V; j++) { sa[j] = 0.f; sb[j] = 0.f; }
    if (c0 < cols) {
#pragma unroll 4
        for (int64_t r = r0 + w; r < r1; r += 4) {
            Vec16<T> va, vb, vo;
            va.load(a + r * lda + c0);
            vb.load(b + r * ldb + c0);
#pragma unroll
            for (int j = 0; j < V; j++) { sa[j] += va.v[j]; sb[j] += vb.v[j]; vo.v[j] = va.v[j] + vb.v[j]; }
            vo.store(y + r * ldy + c0);
        }
    }
#pragma unroll
    for (int j = 0; j < V; j++) { red[w][0][lane * V + j] = sa[j]; red[w][1][lane * V + j] = sb[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 64 * V; i += 256) {
        const int which = i / (64 * V), ii = i % (64 * V);
        const int c = blockIdx.x * 64 * V + ii;
        if (c < cols) part[((int64_t)blockIdx.y * 2 + which) * cols + c] = ((red[0][which][ii] + red[1][which][ii]) + red[2][which][ii]) + red[3][which][ii];
    }
}
// y may alias b.  sum_a_acc[c] += sum_r a[r, c], sum_b_acc[c] += sum_r b[r, c] (float32, fixed summation order).
extern "C" int64_t db1_add2d_colsums_workspace_bytes(int64_t rows, int cols) {
    const int rpc = colsum_rpc(rows);
    return ((rows + rpc - 1) / rpc) * 2 * (int64_t)cols * (int64_t)sizeof(float);
}
extern "C" int db1_add2d_colsums(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, float* sum_a_acc,
                                 float* sum_b_acc, int64_t rows, int cols, int dt, void* ws_, int64_t ws_bytes, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "add2d_colsums: dtype");
    const int V = dt == DB1_F32 ? 4 : 8;
    if (rows <= 0 || cols <= 0 || lda < cols || ldb < cols || ldy < cols) DB1_FAIL(DB1_ERR_BAD_SHAPE, "add2d_colsums: shape");
    if (cols % V || lda % V || ldb % V || ldy % V || !db1_aligned16(a) || !db1_aligned16(b) || !db1_aligned16(y))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "add2d_colsums: needs 16-byte aligned rows (cols, strides multiples of %d)", V);
    if (!sum_a_acc || !sum_b_acc) DB1_FAIL(DB1_ERR_BAD_SHAPE, "add2d_colsums: null accumulator");
    hipStream_t st = (hipStream_t)stream;
    const int rpc = colsum_rpc(rows);
    const int nchunks = (int)((rows + rpc - 1) / rpc);
    DB1_NEED_WS(ws_, ws_bytes, db1_add2d_colsums_workspace_bytes(rows, cols), "add2d_colsums");
    float* ws = (float*)ws_;
    dim3 g((unsigned)((cols / V + 63) / 64), (unsigned)nchunks);
    DB1_DISPATCH_DT(dt, T, (add2d_colsums_kernel<T><<<g, 256, 0, st>>>((const T*)a, lda, (const T*)b, ldb, (T*)y, ldy, ws, rows, cols, rpc)));
    DB1_CHECK_LAUNCH("add2d_colsums");
    colsum_part_reduce_strided_kernel<<<(cols + 63) / 64, 64 * CSR_WAVES, 0, st>>>(ws, sum_a_acc, nchunks, cols, 2 * cols);
    colsum_part_reduce_strided_kernel<<<(cols + 63) / 64, 64 * CSR_WAVES, 0, st>>>(ws + cols, sum_b_acc, nchunks, cols, 2 * cols);
    DB1_CHECK_LAUNCH("add2d_colsums reduce");
    return DB1_OK;
}

// y = dropout(x) (y may alias x): the sites that are not fused into a neighbouring kernel -- the embeddings and the position table
// (transformer_xl.py:545,575; one pass over B x L x d per step) and the pre-LN residual branches
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, Db1Drop drp) {
    constexpr int V = Vec16<T>::N;
    const int64_t nv = n / V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        Vec16<T> a;
        a.load(x + i * V);
        db1_drop_apply<V>(drp, i * V, a.v);
        a.store(y + i * V);
    }
}
extern "C" int db1_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, uint32_t site, uint32_t step, const uint32_t* step_dev, int dt,
                           void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "dropout: dtype");
    const int V = dt == DB1_F32 ? 4 : 8;
    if (n <= 0 || (n % 8)) DB1_FAIL(DB1_ERR_BAD_SHAPE, "dropout: n=%lld must be a positive multiple of 8 (one Philox block)", (long long)n);
    if (p < 0.f || p >= 1.f) DB1_FAIL(DB1_ERR_BAD_SHAPE, "dropout: p=%g", (double)p);
    if (!db1_aligned16(x) || !db1_aligned16(y)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "dropout: alignment");
    const Db1Drop drp = db1_drop_make(p, seed, site, step, step_dev);
    DB1_DISPATCH_DT(dt, T, (dropout_kernel<T><<<grid_for(n / V), 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)y, n, drp)));
    DB1_CHECK_LAUNCH("dropout");
    return DB1_OK;
}

// base[off .. off+len) = 0 for every (off, len) pair of a device table: ONE launch clears the scattered small accumulators of the gradient
// arena (LayerNorm / bias / u, v / embedding-table gradients) -- the large weight gradients are not cleared at all, their first
// writer of a step is a GEMM with beta = 0 (bdm_db1_amd/engine.py)
__global__ __launch_bounds__(256) void zero_segments_kernel(float* __restrict__ base, const int64_t* __restrict__ seg) {
    const int64_t off = seg[2 * blockIdx.y], len = seg[2 * blockIdx.y + 1];
    float* p = base + off;
    if (((off | len) & 3) == 0) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (len >> 2); i += (int64_t)gridDim.x * 256)
            reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
    }
}
extern "C" int db1_zero_segments(float* base, const int64_t* segments, int n_segments, void* stream) {
    if (!base || !segments || n_segments <= 0 || n_segments > 65535) DB1_FAIL(DB1_ERR_BAD_SHAPE, "zero_segments: n=%d", n_segments);
    if (!db1_aligned16(base)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "zero_segments: base alignment");
    zero_segments_kernel<<<dim3(64, (unsigned)n_segments), 256, 0, (hipStream_t)stream>>>(base, segments);
    DB1_CHECK_LAUNCH("zero_segments");
    return DB1_OK;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) stf(y + i, ldf(x + i));
}
extern "C" int db1_cast(const void* x, void* y, int64_t n, int dtIn, int dtOut, void* stream) {
    if (!db1_dt_ok(dtIn) || !db1_dt_ok(dtOut)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "cast: dtype");
    if (n <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "cast: n");
    hipStream_t st = (hipStream_t)stream;
    unsigned g = grid_for(n);
    if (dtIn == DB1_F32 && dtOut == DB1_BF16) cast_kernel<float, bf16_t><<<g, 256, 0, st>>>((const float*)x, (bf16_t*)y, n);
    else if (dtIn == DB1_BF16 && dtOut == DB1_F32) cast_kernel<bf16_t, float><<<g, 256, 0, st>>>((const bf16_t*)x, (float*)y, n);
    else if (dtIn == DB1_F32) cast_kernel<float, float><<<g, 256, 0, st>>>((const float*)x, (float*)y, n);
    else cast_kernel<bf16_t, bf16_t><<<g, 256, 0, st>>>((const bf16_t*)x, (bf16_t*)y, n);
    DB1_CHECK_LAUNCH("cast");
    return DB1_OK;
}

// =====================================================================================
// embeddings (transformer_xl.py:621-672)
// =====================================================================================
template <typename TT, typename TO>
__global__ __launch_bounds__(256) void gather_kernel(const TT* __restrict__ table, const int64_t* __restrict__ ids, TO* __restrict__ out,
                                                     int64_t n_tokens, int d, int64_t ld_out, int64_t n_rows) {
    int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tokens) return;
    const int lane = threadIdx.x & 63;
    const int64_t id = ids[t];
    const bool ok = id >= 0 && id < n_rows;   // ids outside the table give zeros (the loads go to row 0 instead: unconditional, so that they
    const TT* row = table + (ok ? id : 0) * d; // are all in flight at once -- inside a branch each one was waited for: 17 us for ONE token)
    TO* dst = out + t * ld_out;
    if (sizeof(TT) == 2 && sizeof(TO) == 2 && (d % 8) == 0 && (ld_out % 8) == 0 && ((uintptr_t)table % 16) == 0 && ((uintptr_t)out % 16) == 0) {
#pragma unroll 4
        for (int i = lane * 8; i < d; i += 512) {
            uint4 v = *reinterpret_cast<const uint4*>(row + i);
            if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(dst + i) = v;
        }
        return;
    }
#pragma unroll 8
    for (int i = lane; i < d; i += 64) { const float v = ldf(row + i); stf(dst + i, ok ? v : 0.f); }
}
extern "C" int db1_embed_gather_fwd(const void* table, const int64_t* ids, void* out, int64_t n_tokens, int d, int64_t ld_out,
                                    int64_t n_table_rows, int dtTable, int dtOut, void* stream) {
    if (!db1_dt_ok(dtTable) || !db1_dt_ok(dtOut)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "embed_gather: dtype");
    if (n_tokens <= 0 || d <= 0 || ld_out < d || n_table_rows <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "embed_gather: shape");
    hipStream_t st = (hipStream_t)stream;
    dim3 g((unsigned)((n_tokens + 3) / 4));
#define L(A, B) gather_kernel<A, B><<<g, 256, 0, st>>>((const A*)table, ids, (B*)out, n_tokens, d, ld_out, n_table_rows)
    if (dtTable == DB1_F32 && dtOut == DB1_F32) L(float, float);
    else if (dtTable == DB1_BF16 && dtOut == DB1_BF16) L(bf16_t, bf16_t);
    else if (dtTable == DB1_F32) L(float, bf16_t);
    else L(bf16_t, float);
#undef L
    DB1_CHECK_LAUNCH("embed_gather");
    return DB1_OK;
}

// (the embedding-table gradient, db1_embed_scatter_add_bwd, lives in scatter.hip: sorted runs instead of float atomics)
int db1_scatter_add_impl(const void* dout, const int64_t* ids, float* dtable_acc, int64_t n_tokens, int d, int64_t ld_dout, int64_t n_table_rows, int dt,
                         void* ws, int64_t ws_bytes, hipStream_t st, const char* who);

// RL assembly: grid (L / 32 token chunks, B).  The rank of a -1 placeholder inside its row = placeholders before the chunk (counted
// by the whole block) + ballot / popcount prefix inside the chunk.  (One block per ROW left 240 of 256 CUs idle: 3.6 ms per call
// at B = 16.)  dvis must be zero-initialised by the caller: only the rows that have a placeholder are written.
#define RLA_TPB 32
template <typename TT, typename T, bool BWD>
__global__ __launch_bounds__(256) void rl_assemble_kernel(const TT* __restrict__ word_table, const TT* __restrict__ pos_table,
                                                          const T* vis, const int64_t* __restrict__ ids,
                                                          const int64_t* __restrict__ position_id, int64_t* labels, T* out,
                                                          float* dword, float* dpos, T* dvis, int L, int d, int nvis, int64_t n_word,
                                                          int64_t n_pos) {
    __shared__ int cnt_sm[4];
    __shared__ int rank_sm[RLA_TPB];
    const int b = blockIdx.y, c0 = blockIdx.x * RLA_TPB, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t* row = ids + (int64_t)b * L;
    int cnt = 0;
    for (int t = tid; t < c0; t += 256) cnt += row[t] == -1 ? 1 : 0;
    cnt = (int)wave_sum((float)cnt);  // < 2^24: exact in float
    if (lane == 0) cnt_sm[w] = cnt;
    __syncthreads();
    const int before = cnt_sm[0] + cnt_sm[1] + cnt_sm[2] + cnt_sm[3];
    if (w == 0) {
        const int t = c0 + lane;
        const bool ph = lane < RLA_TPB && t < L && row[t] == -1;
        const unsigned long long m = __ballot(ph);
        if (lane < RLA_TPB) rank_sm[lane] = ph ? before + __popcll(m & ((1ull << lane) - 1ull)) : -1;
    }
    __syncthreads();
    if (!BWD && labels) {
        const int t = c0 + tid;
        if (tid < RLA_TPB && t < L && labels[(int64_t)b * L + t] == -1) labels[(int64_t)b * L + t] = 0;
    }
    for (int tt = w; tt < RLA_TPB; tt += 4) {
        const int t = c0 + tt;
        if (t >= L) break;
        const int64_t id = row[t];
        const int rk = rank_sm[tt];
        const int64_t pid = position_id[(int64_t)b * L + t];
        const bool id_ok = id >= 0 && id < n_word, pid_ok = pid >= 0 && pid < n_pos;   // out-of-table ids contribute nothing
        const int64_t o = ((int64_t)b * L + t) * d;
        for (int i = lane; i < d; i += 64) {
            if (!BWD) {
                float v = 0.f;
                if (id_ok) v = ldf(word_table + id * d + i);
                else if (rk >= 0 && rk < nvis && vis) v = ldf(vis + ((int64_t)b * nvis + rk) * d + i);
                if (pid_ok) v += ldf(pos_table + pid * d + i);
                stf(out + o + i, v);
            } else {
                const float g = ldf(out + o + i);  // 'out' carries dout in the backward
                // (the word / position table gradients are two deterministic sorted-run scatters issued by the host function)
                if (!id_ok && rk >= 0 && rk < nvis && dvis) stf(dvis + ((int64_t)b * nvis + rk) * d + i, g);
            }
        }
    }
}

extern "C" int db1_rl_assemble_fwd(const void* word_table, const void* pos_table, const void* vis, const int64_t* ids,
                                   const int64_t* position_id, int64_t* labels, void* out, int B, int L, int d,
                                   int n_vis_per_row, int64_t n_word_rows, int64_t n_pos_rows, int dtTable, int dt, void* stream) {
    if (!db1_dt_ok(dt) || !db1_dt_ok(dtTable)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "rl_assemble_fwd: dtype");
    if (B <= 0 || L <= 0 || d <= 0 || L > 12288) DB1_FAIL(DB1_ERR_BAD_SHAPE, "rl_assemble_fwd: shape");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((unsigned)((L + RLA_TPB - 1) / RLA_TPB), (unsigned)B);
#define L_(TT, T) rl_assemble_kernel<TT, T, false><<<g, 256, 0, st>>>((const TT*)word_table, (const TT*)pos_table, (const T*)vis, ids, position_id, labels, (T*)out, nullptr, nullptr, nullptr, L, d, n_vis_per_row, n_word_rows, n_pos_rows)
    if (dtTable == DB1_F32 && dt == DB1_F32) L_(float, float);
    else if (dtTable == DB1_BF16 && dt == DB1_BF16) L_(bf16_t, bf16_t);
    else if (dtTable == DB1_F32) L_(float, bf16_t);
    else L_(bf16_t, float);
#undef L_
    DB1_CHECK_LAUNCH("rl_assemble_fwd");
    return DB1_OK;
}

extern "C" int64_t db1_embed_scatter_add_workspace_bytes(int64_t n_tokens);
extern "C" int64_t db1_rl_assemble_bwd_workspace_bytes(int B, int L) { return db1_embed_scatter_add_workspace_bytes((int64_t)B * L); }
extern "C" int db1_rl_assemble_bwd(const void* dout, const int64_t* ids, const int64_t* position_id, float* dword_acc,
                                   float* dpos_acc, void* dvis, int B, int L, int d, int n_vis_per_row, int64_t n_word_rows, int64_t n_pos_rows,
                                   int dt, void* ws, int64_t ws_bytes, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "rl_assemble_bwd: dtype");
    if (B <= 0 || L <= 0 || d <= 0 || L > 12288) DB1_FAIL(DB1_ERR_BAD_SHAPE, "rl_assemble_bwd: shape");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((unsigned)((L + RLA_TPB - 1) / RLA_TPB), (unsigned)B);
    if (dvis && n_vis_per_row > 0) hipMemsetAsync(dvis, 0, (size_t)B * n_vis_per_row * d * (dt == DB1_F32 ? 4 : 2), st);
    DB1_DISPATCH_DT(dt, T, (rl_assemble_kernel<float, T, true><<<g, 256, 0, st>>>(nullptr, nullptr, nullptr, ids, position_id, nullptr,
                                                                                  (T*)const_cast<void*>(dout), dword_acc, dpos_acc, (T*)dvis, L, d, n_vis_per_row, n_word_rows, n_pos_rows)));
    DB1_CHECK_LAUNCH("rl_assemble_bwd");
    int rc = db1_scatter_add_impl(dout, ids, dword_acc, (int64_t)B * L, d, d, n_word_rows, dt, ws, ws_bytes, st, "rl_assemble_bwd (word table)");
    if (rc) return rc;
    return db1_scatter_add_impl(dout, position_id, dpos_acc, (int64_t)B * L, d, d, n_pos_rows, dt, ws, ws_bytes, st, "rl_assemble_bwd (position table)");
}

// =====================================================================================
// masked cross-entropy over materialised logits (transformer_xl.py:602-609)
// one 256-thread block per token row, online (max, sum-exp) per thread, block reduction
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ mask, float* __restrict__ lse, float* sums,
                                                     float* __restrict__ tok_loss, int V, int64_t ld, int vec_ok) {
    __shared__ float sm[4];
    const int64_t t = blockIdx.x;
    const T* row = logits + t * ld;
    const int tid = threadIdx.x;
    float m = -3.0e38f, s = 0.f;
    constexpr int VN = Vec16<T>::N;
    if (vec_ok) {
        for (int c = tid * VN; c < V; c += 256 * VN) {
            Vec16<T> a;
            a.load(row + c);
#pragma unroll
            for (int j = 0; j < VN; j++) if (c + j >= V) a.v[j] = -3.0e38f;   // padded vocabulary columns
            float vm = a.v[0];
#pragma unroll
            for (int j = 1; j < VN; j++) vm = fmaxf(vm, a.v[j]);
            if (vm > m) { s *= __expf(m - vm); m = vm; }                        // one rescale per vector, rare after the first few
#pragma unroll
            for (int j = 0; j < VN; j++) s += __expf(a.v[j] - m);
        }
    } else {
        for (int c = tid; c < V; c += 256) {
            float x = ldf(row + c);
            if (x > m) { s = s * __expf(m - x) + 1.f; m = x; } else s += __expf(x - m);
        }
    }
    const float M = block_max256(m, sm);
    const float S = block_sum256(s * __expf(m - M), sm);
    if (tid == 0) {
        const float l = M + logf(S);
        lse[t] = l;
        int64_t y = labels[t];
        // a label outside [0, V) (e.g. torch's ignore_index -100, which the reference's CrossEntropyLoss skips) reads nothing and
        // adds no loss; the backward gives such a row no gradient
        const bool y_ok = y >= 0 && y < V;
        const float mk = y_ok ? mask[t] : 0.f;
        const float nll = y_ok ? l - ldf(row + y) : 0.f;
        tok_loss[t] = mk * nll;  // summed in a fixed order by ce_sum_kernel (65 536 same-address atomics took ~1 ms, and made the loss order-dependent)
    }
}
// sums[0] += sum_t tok_loss[t], sums[1] += sum_t mask[t]: one block, fixed order (deterministic loss)
__global__ __launch_bounds__(1024) void ce_sum_kernel(const float* __restrict__ tok_loss, const float* __restrict__ mask, float* sums, int64_t T_) {
    __shared__ float red[2][1024];
    float a = 0.f, b = 0.f;
    for (int64_t t = threadIdx.x; t < T_; t += 1024) { a += tok_loss[t]; b += mask[t]; }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[0] += red[0][0]; sums[1] += red[1][0]; }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* logits, const int64_t* __restrict__ labels, const float* __restrict__ mask,
                                                     const float* __restrict__ lse, const float* __restrict__ sums, T* dlogits,
                                                     int V, int64_t ld, float gscale, int vec_ok) {
    const int64_t t = blockIdx.x;
    const T* row = logits + t * ld;
    T* drow = dlogits + t * ld;
    const int tid = threadIdx.x;
    const float l = lse[t];
    const int64_t yl = labels[t];
    const int y = (yl >= 0 && yl < V) ? (int)yl : -1;
    const float w = (y >= 0 ? mask[t] : 0.f) / sums[1] * gscale;
    constexpr int VN = Vec16<T>::N;
    if (vec_ok) {
        for (int c = tid * VN; c < ld; c += 256 * VN) {
            Vec16<T> a, o;
            a.load(row + c);
#pragma unroll
            for (int j = 0; j < VN; j++) o.v[j] = (c + j < V) ? w * (__expf(a.v[j] - l) - (c + j == y ? 1.f : 0.f)) : 0.f;
            o.store(drow + c);
        }
    } else {
        for (int c = tid; c < ld; c += 256) stf(drow + c, c < V ? w * (__expf(ldf(row + c) - l) - (c == y ? 1.f : 0.f)) : 0.f);
    }
}

// Forward and backward of a row in ONE pass (the chunked head + loss sweep, lmhead_ce.hip): the row of logits (<= 256 x 8 x CE_MAXV
// elements: 34 816) stays in the registers of its workgroup between the (max, sum-exp) reduction and the gradient, so the chunk of logits
// is read once and written once instead of read twice and written once (3.3 -> 2.2 GB per 16 384-row chunk at the DB1 vocabulary).  Same
// arithmetic as ce_fwd_kernel + ce_bwd_kernel, expression for expression: lse, the per-token losses and dlogits are bit-equal.
#define CE_MAXV 17
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(T* logits, const int64_t* __restrict__ labels, const float* __restrict__ mask,
                                                         float* __restrict__ lse, const float* __restrict__ norm, float* __restrict__ tok_loss,
                                                         int V, int64_t ld, float gscale) {
    __shared__ float sm[4];
    constexpr int VN = Vec16<T>::N;
    const int64_t t = blockIdx.x;
    T* row = logits + t * ld;
    const int tid = threadIdx.x;
    uint4 raw[CE_MAXV];    // the row as loaded (16 bytes per piece: half the registers of the unpacked form -> twice the rows in flight per CU)
#pragma unroll
    for (int k = 0; k < CE_MAXV; k++) {   // (unconditional, at a clamped address: loads inside a branch are waited for one by one)
        const int c = (k * 256 + tid) * VN;
        raw[k] = *reinterpret_cast<const uint4*>(row + (c < ld ? c : (int)ld - VN));
    }
    auto unpack = [&](int k, float (&x)[VN]) {
        Vec16<T> a;
        a.load(reinterpret_cast<const T*>(&raw[k]));
#pragma unroll
        for (int j = 0; j < VN; j++) x[j] = a.v[j];
    };
    float m = -3.0e38f, s = 0.f;
#pragma unroll
    for (int k = 0; k < CE_MAXV; k++) {
        const int c = (k * 256 + tid) * VN;
        if (c < V) {
            float x[VN];
            unpack(k, x);
#pragma unroll
            for (int j = 0; j < VN; j++) x[j] = (c + j >= V) ? -3.0e38f : x[j];   // padded vocabulary columns
            float vm = x[0];
#pragma unroll
            for (int j = 1; j < VN; j++) vm = fmaxf(vm, x[j]);
            if (vm > m) { s *= __expf(m - vm); m = vm; }
#pragma unroll
            for (int j = 0; j < VN; j++) s += __expf(x[j] - m);
        }
    }
    const float M = block_max256(m, sm);
    const float S = block_sum256(s * __expf(m - M), sm);
    const float l = M + logf(S);
    const int64_t yl = labels[t];
    const int y = (yl >= 0 && yl < V) ? (int)yl : -1;
    const float mk = y >= 0 ? mask[t] : 0.f;
    if (tid == 0) {
        lse[t] = l;
        tok_loss[t] = mk * (y >= 0 ? l - ldf(row + y) : 0.f);     // (read before any thread overwrites the row: see the barrier below)
    }
    __syncthreads();
    const float w = mk / norm[1] * gscale;
#pragma unroll
    for (int k = 0; k < CE_MAXV; k++) {
        const int c = (k * 256 + tid) * VN;
        if (c < ld) {
            float x[VN];
            unpack(k, x);
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VN; j++) o.v[j] = (c + j < V) ? w * (__expf(x[j] - l) - (c + j == y ? 1.f : 0.f)) : 0.f;
            o.store(row + c);
        }
    }
}
// dlogits in place + lse + sums[0] += sum(mask * nll), sums[1] += sum(mask); norm[1] = the loss normaliser (sum(mask) over ALL rows of the step)
extern "C" int db1_masked_ce_fwd_bwd_supported(int V, int64_t ld, int dt) {
    const int VN = dt == DB1_F32 ? 4 : 8;
    return (db1_dt_ok(dt) && V > 0 && ld >= V && ld % VN == 0 && ld <= (int64_t)256 * VN * CE_MAXV) ? 1 : 0;
}
extern "C" int db1_masked_ce_fwd_bwd(void* logits, const int64_t* labels, const float* mask, float* lse, float* sums, const float* norm,
                                     int64_t T_, int V, int64_t ld, float gscale, int dt, void* ws_, int64_t ws_bytes, void* stream) {
    if (!db1_masked_ce_fwd_bwd_supported(V, ld, dt) || !db1_aligned16(logits)) DB1_FAIL(DB1_ERR_UNSUPPORTED, "masked_ce_fwd_bwd: V=%d ld=%lld (rows of at most %d elements, 16-byte aligned)", V, (long long)ld, 256 * 8 * CE_MAXV);
    if (T_ <= 0 || !labels || !mask || !lse || !sums || !norm) DB1_FAIL(DB1_ERR_BAD_SHAPE, "masked_ce_fwd_bwd: shape / null buffer");
    DB1_NEED_WS(ws_, ws_bytes, T_ * (int64_t)sizeof(float), "masked_ce_fwd_bwd");
    float* tok_loss = (float*)ws_;
    DB1_DISPATCH_DT(dt, T, (ce_fwd_bwd_kernel<T><<<(unsigned)T_, 256, 0, (hipStream_t)stream>>>((T*)logits, labels, mask, lse, norm, tok_loss, V, ld, gscale)));
    DB1_CHECK_LAUNCH("masked_ce_fwd_bwd");
    ce_sum_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(tok_loss, mask, sums, T_);
    DB1_CHECK_LAUNCH("masked_ce_fwd_bwd sum");
    return DB1_OK;
}

extern "C" int64_t db1_masked_ce_fwd_workspace_bytes(int64_t T_) { return T_ * (int64_t)sizeof(float); }  // per-token losses
extern "C" int db1_masked_ce_fwd(const void* logits, const int64_t* labels, const float* mask, float* lse, float* sums,
                                 int64_t T_, int V, int64_t ld, int dt, void* ws_, int64_t ws_bytes, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "masked_ce_fwd: dtype");
    if (T_ <= 0 || V <= 0 || ld < V) DB1_FAIL(DB1_ERR_BAD_SHAPE, "masked_ce_fwd: shape");
    const int VN = dt == DB1_F32 ? 4 : 8;
    const int vec_ok = (ld % VN == 0) && db1_aligned16(logits);
    DB1_NEED_WS(ws_, ws_bytes, db1_masked_ce_fwd_workspace_bytes(T_), "masked_ce_fwd");
    float* tok_loss = (float*)ws_;
    DB1_DISPATCH_DT(dt, T, (ce_fwd_kernel<T><<<(unsigned)T_, 256, 0, (hipStream_t)stream>>>((const T*)logits, labels, mask, lse, sums, tok_loss, V, ld, vec_ok)));
    DB1_CHECK_LAUNCH("masked_ce_fwd");
    ce_sum_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(tok_loss, mask, sums, T_);
    DB1_CHECK_LAUNCH("masked_ce_fwd sum");
    return DB1_OK;
}
extern "C" int db1_masked_ce_bwd(const void* logits, const int64_t* labels, const float* mask, const float* lse, const float* sums,
                                 void* dlogits, int64_t T_, int V, int64_t ld, float gscale, int dt, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "masked_ce_bwd: dtype");
    if (T_ <= 0 || V <= 0 || ld < V) DB1_FAIL(DB1_ERR_BAD_SHAPE, "masked_ce_bwd: shape");
    const int VN = dt == DB1_F32 ? 4 : 8;
    const int vec_ok = (ld % VN == 0) && db1_aligned16(logits) && db1_aligned16(dlogits);
    DB1_DISPATCH_DT(dt, T, (ce_bwd_kernel<T><<<(unsigned)T_, 256, 0, (hipStream_t)stream>>>((const T*)logits, labels, mask, lse, sums, (T*)dlogits, V, ld, gscale, vec_ok)));
    DB1_CHECK_LAUNCH("masked_ce_bwd");
    return DB1_OK;
}

// =====================================================================================
// optimizer: global-norm and fused Adam / AdamW (28-30 B per parameter per step, HBM-bound)
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* __restrict__ x, float* acc, int64_t n) {
    __shared__ float sm[4];
    float a = 0.f;
    constexpr int V = Vec16<T>::N;
    const int64_t nv = n / V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        Vec16<T> v;
        v.load(x + i * V);
#pragma unroll
        for (int j = 0; j < V; j++) a += v.v[j] * v.v[j];
    }
    if (blockIdx.x == 0) for (int64_t i = nv * V + threadIdx.x; i < n; i += 256) { float f = ldf(x + i); a += f * f; }
    a = block_sum256(a, sm);
    if (threadIdx.x == 0) acc[0] += a;     // (launched as ONE workgroup)
}
// the form for large vectors: per-workgroup partial sums into the caller's workspace, then ONE workgroup adds them in a fixed order
// (per-workgroup atomic adds would make the global norm -- and through the clip coefficient every parameter -- depend on arrival order)
template <typename T>
__global__ __launch_bounds__(256) void sumsq_part_kernel(const T* __restrict__ x, float* __restrict__ part, int64_t n) {
    __shared__ float sm[4];
    float a = 0.f;
    constexpr int V = Vec16<T>::N;
    const int64_t nv = n / V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        Vec16<T> v;
        v.load(x + i * V);
#pragma unroll
        for (int j = 0; j < V; j++) a += v.v[j] * v.v[j];
    }
    if (blockIdx.x == 0) for (int64_t i = nv * V + threadIdx.x; i < n; i += 256) { float f = ldf(x + i); a += f * f; }
    a = block_sum256(a, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nparts, float* acc, int overwrite) {
    __shared__ float sm[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];      // thread t: parts t, t + 256, ... in order
    a = block_sum256(a, sm);
    if (threadIdx.x == 0) acc[0] = overwrite ? a : acc[0] + a;
}
extern "C" int64_t db1_sumsq_det_workspace_bytes(int64_t n) { return n > 0 ? (int64_t)(256 * 16) * (int64_t)sizeof(float) : 0; }
extern "C" int db1_sumsq_det(const void* x, float* acc, int64_t n, int dt, int overwrite, void* ws, int64_t ws_bytes, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "sumsq_det: dtype");
    if (n <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "sumsq_det: n");
    if (!db1_aligned16(x)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "sumsq_det: alignment");
    DB1_NEED_WS(ws, ws_bytes, db1_sumsq_det_workspace_bytes(n), "sumsq_det");
    const int V = dt == DB1_F32 ? 4 : 8;
    const unsigned g = grid_for(n / V + 1);
    DB1_DISPATCH_DT(dt, T, (sumsq_part_kernel<T><<<g, 256, 0, (hipStream_t)stream>>>((const T*)x, (float*)ws, n)));
    DB1_CHECK_LAUNCH("sumsq_part");
    sumsq_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>((const float*)ws, (int)g, acc, overwrite);
    DB1_CHECK_LAUNCH("sumsq_final");
    return DB1_OK;
}

extern "C" int db1_sumsq_acc(const void* x, float* acc, int64_t n, int dt, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "sumsq: dtype");
    if (n <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "sumsq: n");
    if (!db1_aligned16(x)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "sumsq: alignment");
    // workspace-free form: ONE workgroup (deterministic; fine for a few million elements).  Large vectors: db1_sumsq_det / db1_grad_norm_sq --
    // refused here instead of running them at 1 / 256 of the chip (a silent cliff for a C caller: 16 M floats are ~1 ms in one workgroup)
    if (n > ((int64_t)1 << 24)) DB1_FAIL(DB1_ERR_UNSUPPORTED, "sumsq_acc: n=%lld > 2^24 runs in ONE workgroup; use db1_sumsq_det / db1_grad_norm_sq", (long long)n);
    DB1_DISPATCH_DT(dt, T, (sumsq_kernel<T><<<1, 256, 0, (hipStream_t)stream>>>((const T*)x, acc, n)));
    DB1_CHECK_LAUNCH("sumsq");
    return DB1_OK;
}

template <bool HAS_WORK, typename GT>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const GT* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ pw, int64_t n, float lr, float b1,
                                                   float b2, float omb1, float omb2, float eps, float wd, int adamw, float bc1, float rsqrt_bc2,
                                                   float gscale, float clip, const float* norm_sq) {
    float gs = gscale;
    if (clip > 0.f && norm_sq) {
        const float nrm = sqrtf(*norm_sq) * gscale;
        gs *= fminf(1.f, clip / (nrm + 1e-6f));
    }
    const float step_size = lr / bc1;
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        float gg[4];
        if (sizeof(GT) == 4) {
            float4 G = reinterpret_cast<const float4*>(g)[i];
            gg[0] = G.x; gg[1] = G.y; gg[2] = G.z; gg[3] = G.w;
        } else {  // bf16 gradients: the all-reduced staging copy of the data-parallel engine (8 B per 4 elements)
            uint2 G = reinterpret_cast<const uint2*>(g)[i];
            gg[0] = __uint_as_float(G.x << 16); gg[1] = __uint_as_float(G.x & 0xffff0000u);
            gg[2] = __uint_as_float(G.y << 16); gg[3] = __uint_as_float(G.y & 0xffff0000u);
        }
        float4 M = reinterpret_cast<float4*>(m)[i], Vv = reinterpret_cast<float4*>(v)[i];
        float pp[4] = {P.x, P.y, P.z, P.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {Vv.x, Vv.y, Vv.z, Vv.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float gr = gg[j] * gs;
            if (adamw) pp[j] *= (1.f - lr * wd); else gr += wd * pp[j];
            mm[j] = b1 * mm[j] + omb1 * gr;
            vv[j] = b2 * vv[j] + omb2 * gr * gr;
            const float denom = sqrtf(vv[j]) * rsqrt_bc2 + eps;
            pp[j] -= step_size * mm[j] / denom;
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (HAS_WORK) {
            uint2 o;
            o.x = f2bf_pk(pp[0], pp[1]);
            o.y = f2bf_pk(pp[2], pp[3]);
            reinterpret_cast<uint2*>(pw)[i] = o;
        }
    }
}
extern "C" int db1_adam_step(float* p32, const void* g, float* m, float* v, void* p_work, int64_t n, double lr, double beta1,
                             double beta2, double eps, double wd, int adamw, int step, float gscale, float clip,
                             const float* norm_sq, int dtGrad, int dtWork, void* stream) {
    if (n <= 0 || (n & 3)) DB1_FAIL(DB1_ERR_BAD_SHAPE, "adam: n=%lld must be a positive multiple of 4", (long long)n);
    if (step < 1) DB1_FAIL(DB1_ERR_BAD_SHAPE, "adam: step must be >= 1");
    if (!db1_dt_ok(dtGrad)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "adam: gradient dtype");
    if (!db1_aligned16(p32) || (((uintptr_t)g) & (dtGrad == DB1_F32 ? 15 : 7)) || !db1_aligned16(m) || !db1_aligned16(v) || (p_work && (((uintptr_t)p_work) & 7)))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "adam: alignment");
    if (p_work && dtWork != DB1_BF16) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "adam: working copy must be bf16");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipStream_t st = (hipStream_t)stream;
    unsigned gr = grid_for(n / 4);
#define DB1_ADAM_LAUNCH(HW, GT) adam_kernel<HW, GT><<<gr, 256, 0, st>>>(p32, (const GT*)g, m, v, (bf16_t*)p_work, n, (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)wd, adamw, (float)bc1, (float)(1.0 / sqrt(bc2)), gscale, clip, norm_sq)
    if (p_work) { if (dtGrad == DB1_F32) DB1_ADAM_LAUNCH(true, float); else DB1_ADAM_LAUNCH(true, bf16_t); }
    else { if (dtGrad == DB1_F32) DB1_ADAM_LAUNCH(false, float); else DB1_ADAM_LAUNCH(false, bf16_t); }
#undef DB1_ADAM_LAUNCH
    DB1_CHECK_LAUNCH("adam");
    return DB1_OK;
}

// =====================================================================================
// mu-law scalar tokenizer (scalar_tokenizer.py:28-45): float32 op order of the reference with a
// correctly rounded logarithm (float64 log rounded once) -> ids identical to torch-CPU.
// =====================================================================================
__global__ __launch_bounds__(256) void mulaw_kernel(const float* __restrict__ x, int32_t* __restrict__ ids, int64_t n, int is_action,
                                                    int nb, float mu, float den) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = x[i];
        if (!is_action) {
            const float t = __fadd_rn(__fmul_rn(fabsf(v), mu), 1.0f);
            const float lg = (float)log((double)t);
            const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
            float y = __fdiv_rn(__fmul_rn(sg, lg), den);
            if (v != v) y = v;
            v = fminf(fmaxf(y, -1.f), 1.f);
        }
        const float z = __fmul_rn(__fdiv_rn(__fadd_rn(v, 1.0f), 2.0f), (float)nb);
        int id = (int)z;  // truncation toward zero, as torch .int()
        id = id < 0 ? 0 : (id > nb - 1 ? nb - 1 : id);
        ids[i] = id;
    }
}
extern "C" int db1_mulaw_discretize(const float* x, int32_t* ids, int64_t n, int is_action, int num_bins, float mu, float M, void* stream) {
    if (n <= 0 || num_bins <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "mulaw: shape");
    const float den = (float)log((double)(float)(mu * M + 1.0f));
    mulaw_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(x, ids, n, is_action, num_bins, mu, den);
    DB1_CHECK_LAUNCH("mulaw");
    return DB1_OK;
}

// ContinuousScalarTokenizer.decode (scalar_tokenizer.py:47-63): ids clipped to [0, nb-1]; x = id/nb*2 - 1 in the reference's float32 op
// order; observations: sign(x) * (base^|x| - 1) / mu with base = 1 + M*mu, the power taken in float64 and rounded once (within 1 ulp of
// torch's float32 pow).  *oob (nullable) is set to 1 if any id was outside [0, nb-1] (the reference prints a warning there).
template <typename IT>
__global__ __launch_bounds__(256) void mulaw_decode_kernel(const IT* __restrict__ ids, float* __restrict__ out, int64_t n, int is_action,
                                                           int nb, float mu, float base, int* oob) {
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        long long id = (long long)ids[i];
        if (id < 0 || id > nb - 1) { bad = 1; id = id < 0 ? 0 : nb - 1; }
        float x = __fsub_rn(__fmul_rn(__fdiv_rn((float)id, (float)nb), 2.0f), 1.0f);
        if (!is_action) {
            const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
            const float pw = (float)pow((double)base, (double)fabsf(x));
            x = __fdiv_rn(__fmul_rn(sg, __fsub_rn(pw, 1.0f)), mu);
        }
        out[i] = x;
    }
    if (oob && bad) atomicOr(oob, 1);
}
extern "C" int db1_mulaw_decode(const void* ids, float* out, int64_t n, int ids_are_int64, int is_action, int num_bins, float mu, float M,
                                int* oob_flag, void* stream) {
    if (n <= 0 || num_bins <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "mulaw_decode: shape");
    const float base = (float)(1.0 + (double)M * (double)mu);
    if (ids_are_int64) mulaw_decode_kernel<int64_t><<<grid_for(n), 256, 0, (hipStream_t)stream>>>((const int64_t*)ids, out, n, is_action, num_bins, mu, base, oob_flag);
    else mulaw_decode_kernel<int32_t><<<grid_for(n), 256, 0, (hipStream_t)stream>>>((const int32_t*)ids, out, n, is_action, num_bins, mu, base, oob_flag);
    DB1_CHECK_LAUNCH("mulaw_decode");
    return DB1_OK;
}
