// Image-patch embedder pieces (src/tokenizer/vision_embedding.py:65-86): per-patch normalisation,
// im2col / col2im for the per-patch 3x3 convolutions (zero padding at PATCH borders, so patches are
// independent), NCHW<->NHWC shuffles around the GEMMs, GroupNorm(32)+GELU forward/backward.
// All HBM-bound streaming kernels; the convolutions themselves run on the GEMM kernels.
#include "db1_common.h"

// ---- patchify + normalise: one wave per (patch, channel) -> p*p values
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void patch_normalize_kernel(const TI* __restrict__ pix, TO* __restrict__ out, int64_t n_pc, int C, int Himg,
                                                              int Wimg, int p) {
    const int64_t pc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pc >= n_pc) return;
    const int lane = threadIdx.x & 63;
    const int c = (int)(pc % C);
    const int64_t patch = pc / C;
    const int hp = Himg / p, wp = Wimg / p;
    const int64_t img = patch / (hp * wp);
    const int ph = (int)((patch / wp) % hp), pw = (int)(patch % wp);
    const TI* src = pix + ((img * C + c) * Himg + (int64_t)ph * p) * Wimg + (int64_t)pw * p;
    const int n = p * p;
    float s = 0.f;
    for (int e = lane; e < n; e += 64) s += ldf(src + (int64_t)(e / p) * Wimg + (e % p));
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
    for (int e = lane; e < n; e += 64) { float d = ldf(src + (int64_t)(e / p) * Wimg + (e % p)) - mean; q += d * d; }
    const float stdv = sqrtf(wave_sum(q) / (float)(n - 1));  // torch.std default: unbiased
    const float inv = 1.f / ((1e-6f + stdv) * sqrtf((float)p));
    TO* dst = out + pc * n;
    for (int e = lane; e < n; e += 64) stf(dst + e, (ldf(src + (int64_t)(e / p) * Wimg + (e % p)) - mean) * inv);
}

extern "C" int db1_patch_normalize(const void* pixels, void* patches, int n_img, int C, int Himg, int Wimg, int p, int dtIn, int dtOut,
                                   void* stream) {
    if (!db1_dt_ok(dtIn) || !db1_dt_ok(dtOut)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "patch_normalize: dtype");
    if (n_img <= 0 || C <= 0 || p <= 1 || Himg % p || Wimg % p) DB1_FAIL(DB1_ERR_BAD_SHAPE, "patch_normalize: image %dx%d not divisible by patch %d", Himg, Wimg, p);
    const int64_t n_pc = (int64_t)n_img * (Himg / p) * (Wimg / p) * C;
    dim3 g((unsigned)((n_pc + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
#define L_(A, B) patch_normalize_kernel<A, B><<<g, 256, 0, st>>>((const A*)pixels, (B*)patches, n_pc, C, Himg, Wimg, p)
    if (dtIn == DB1_F32 && dtOut == DB1_F32) L_(float, float);
    else if (dtIn == DB1_F32) L_(float, bf16_t);
    else if (dtOut == DB1_F32) L_(bf16_t, float);
    else L_(bf16_t, bf16_t);
#undef L_
    DB1_CHECK_LAUNCH("patch_normalize");
    return DB1_OK;
}

// ---- im2col 3x3 pad 1: x [N, C, p, p] -> cols [N*p*p, C*9], column index = c*9 + ky*3 + kx (matches weight.reshape(Cout, C*9))
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ cols, int64_t total, int C, int p, int K) {
    // K = row stride of cols (>= C*9; the padding columns are written as zeros)
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int col = (int)(idx % K);
        const int64_t pix = idx / K;
        if (col >= C * 9) { cols[idx] = 0; continue; }
        const int c = col / 9, ky = (col % 9) / 3, kx = col % 3;
        const int xw = (int)(pix % p), yh = (int)((pix / p) % p);
        const int64_t n = pix / (p * p);
        const int yy = yh + ky - 1, xx = xw + kx - 1;
        T v = 0;
        if (yy >= 0 && yy < p && xx >= 0 && xx < p) v = x[((n * C + c) * p + yy) * p + xx];
        cols[idx] = v;
    }
}
extern "C" int db1_im2col3x3(const void* x, void* cols, int64_t N, int C, int p, int kpad, int dt, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "im2col: dtype");
    if (N <= 0 || C <= 0 || p <= 0 || kpad < C * 9) DB1_FAIL(DB1_ERR_BAD_SHAPE, "im2col: shape");
    const int64_t total = N * p * p * kpad;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    DB1_DISPATCH_DT(dt, T, (im2col_kernel<T><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)cols, total, C, p, kpad)));
    DB1_CHECK_LAUNCH("im2col");
    return DB1_OK;
}

// ---- col2im: dx[n,c,y,x] = sum over the (<= 9) column entries that read it (gather form, no atomics)
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcols, T* __restrict__ dx, int64_t total, int C, int p, int K) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int xw = (int)(idx % p), yh = (int)((idx / p) % p);
        const int c = (int)((idx / (p * p)) % C);
        const int64_t n = idx / ((int64_t)p * p * C);
        float a = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int oy = yh - ky + 1, ox = xw - kx + 1;  // output pixel whose window position (ky,kx) is this input pixel
                if (oy >= 0 && oy < p && ox >= 0 && ox < p) a += ldf(dcols + ((n * p + oy) * p + ox) * K + c * 9 + ky * 3 + kx);
            }
        stf(dx + idx, a);
    }
}
extern "C" int db1_col2im3x3(const void* dcols, void* dx, int64_t N, int C, int p, int kpad, int dt, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "col2im: dtype");
    if (N <= 0 || C <= 0 || p <= 0 || kpad < C * 9) DB1_FAIL(DB1_ERR_BAD_SHAPE, "col2im: shape");
    const int64_t total = N * C * p * p;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    DB1_DISPATCH_DT(dt, T, (col2im_kernel<T><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const T*)dcols, (T*)dx, total, C, p, kpad)));
    DB1_CHECK_LAUNCH("col2im");
    return DB1_OK;
}

// ---- layout shuffles: [N, hw, C] <-> [N, C, hw] through a 32x33 LDS tile
template <typename T, bool TO_NCHW>
__global__ __launch_bounds__(256) void nhwc_nchw_kernel(const T* __restrict__ x, T* __restrict__ y, int C, int hw) {
    __shared__ float tile[32][33];
    const int64_t n = blockIdx.z;
    const int c0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    // NHWC element (s, c) at (n*hw + s)*C + c ; NCHW element (c, s) at (n*C + c)*hw + s
    for (int k = ty; k < 32; k += 8) {
        if (TO_NCHW) { const int s = s0 + k, c = c0 + tx; tile[k][tx] = (s < hw && c < C) ? ldf(x + (n * hw + s) * C + c) : 0.f; }
        else { const int c = c0 + k, s = s0 + tx; tile[k][tx] = (s < hw && c < C) ? ldf(x + (n * C + c) * hw + s) : 0.f; }
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        if (TO_NCHW) { const int c = c0 + k, s = s0 + tx; if (s < hw && c < C) stf(y + (n * C + c) * hw + s, tile[tx][k]); }
        else { const int s = s0 + k, c = c0 + tx; if (s < hw && c < C) stf(y + (n * hw + s) * C + c, tile[tx][k]); }
    }
}
static int shuffle_launch(const void* x, void* y, int64_t N, int C, int hw, int dt, bool to_nchw, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "layout shuffle: dtype");
    if (N <= 0 || N > 65535 * 32 || C <= 0 || hw <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "layout shuffle: shape");
    hipStream_t st = (hipStream_t)stream;
    for (int64_t nb = 0; nb < N; nb += 65535) {  // gridDim.z limit
        const int64_t nn = N - nb < 65535 ? N - nb : 65535;
        dim3 g((unsigned)((C + 31) / 32), (unsigned)((hw + 31) / 32), (unsigned)nn);
        const int64_t off = nb * C * hw;
        DB1_DISPATCH_DT(dt, T, {
            if (to_nchw) nhwc_nchw_kernel<T, true><<<g, 256, 0, st>>>((const T*)x + off, (T*)y + off, C, hw);
            else nhwc_nchw_kernel<T, false><<<g, 256, 0, st>>>((const T*)x + off, (T*)y + off, C, hw);
        });
    }
    DB1_CHECK_LAUNCH("layout shuffle");
    return DB1_OK;
}
extern "C" int db1_nhwc_to_nchw(const void* x, void* y, int64_t N, int C, int hw, int dt, void* stream) { return shuffle_launch(x, y, N, C, hw, dt, true, stream); }
extern "C" int db1_nchw_to_nhwc(const void* x, void* y, int64_t N, int C, int hw, int dt, void* stream) { return shuffle_launch(x, y, N, C, hw, dt, false, stream); }

// ---- GroupNorm + GELU: one wave per (sample, group); group = cpg channels x hw contiguous elements in NCHW
template <typename T, typename TP>
__global__ __launch_bounds__(256) void gn_gelu_fwd_kernel(const T* __restrict__ x, const TP* __restrict__ gamma, const TP* __restrict__ beta,
                                                          T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t n_groups_total,
                                                          int groups, int cpg, int hw, float eps) {
    const int64_t gi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gi >= n_groups_total) return;
    const int lane = threadIdx.x & 63;
    const int g = (int)(gi % groups);
    const int n = cpg * hw;
    const T* xs = x + gi * n;
    float s = 0.f;
    for (int e = lane; e < n; e += 64) s += ldf(xs + e);
    const float mu = wave_sum(s) / (float)n;
    float q = 0.f;
    for (int e = lane; e < n; e += 64) { float d = ldf(xs + e) - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / (float)n + eps);
    if (lane == 0) { mean[gi] = mu; rstd[gi] = rs; }
    T* ys = y + gi * n;
    for (int e = lane; e < n; e += 64) {
        const int c = g * cpg + e / hw;
        const float h = (ldf(xs + e) - mu) * rs * ldf(gamma + c) + ldf(beta + c);
        stf(ys + e, gelu_erf(h));
    }
}
extern "C" int db1_groupnorm_gelu_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int64_t N,
                                      int C, int hw, int groups, float eps, int dt, int dtParam, void* stream) {
    if (!db1_dt_ok(dt) || !db1_dt_ok(dtParam)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "groupnorm_gelu_fwd: dtype");
    if (N <= 0 || C <= 0 || hw <= 0 || groups <= 0 || C % groups) DB1_FAIL(DB1_ERR_BAD_SHAPE, "groupnorm_gelu_fwd: shape");
    const int64_t ng = N * groups;
    dim3 g((unsigned)((ng + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
#define L_(T, TP) gn_gelu_fwd_kernel<T, TP><<<g, 256, 0, st>>>((const T*)x, (const TP*)gamma, (const TP*)beta, (T*)y, mean, rstd, ng, groups, C / groups, hw, eps)
    if (dt == DB1_F32 && dtParam == DB1_F32) L_(float, float);
    else if (dt == DB1_BF16 && dtParam == DB1_BF16) L_(bf16_t, bf16_t);
    else if (dt == DB1_BF16) L_(bf16_t, float);
    else L_(float, bf16_t);
#undef L_
    DB1_CHECK_LAUNCH("groupnorm_gelu_fwd");
    return DB1_OK;
}

// backward: dh = dy * gelu'(h); GN backward on dh; dgamma[c] += sum dh*xhat, dbeta[c] += sum dh
template <typename T, typename TP>
__global__ __launch_bounds__(256) void gn_gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const TP* __restrict__ gamma,
                                                          const TP* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          T* __restrict__ dx, float* dgamma, float* dbeta, int64_t n_groups_total, int groups,
                                                          int cpg, int hw) {
    const int64_t gi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gi >= n_groups_total) return;
    const int lane = threadIdx.x & 63;
    const int g = (int)(gi % groups);
    const int n = cpg * hw;
    const T* xs = x + gi * n;
    const T* dys = dy + gi * n;
    const float mu = mean[gi], rs = rstd[gi];
    float c1 = 0.f, c2 = 0.f;
    for (int ch = 0; ch < cpg; ch++) {
        const int c = g * cpg + ch;
        const float gm = ldf(gamma + c), bt = ldf(beta + c);
        float sg = 0.f, sb = 0.f;
        for (int e = lane; e < hw; e += 64) {
            const float xh = (ldf(xs + ch * hw + e) - mu) * rs;
            const float dh = ldf(dys + ch * hw + e) * gelu_erf_grad(xh * gm + bt);
            sg += dh * xh;
            sb += dh;
        }
        sg = wave_sum(sg);
        sb = wave_sum(sb);
        if (lane == 0) { atomicAdd(dgamma + c, sg); atomicAdd(dbeta + c, sb); }
        c1 += sb * gm;  // sum of (dh * gamma)
        c2 += sg * gm;  // sum of (dh * gamma * xhat)
    }
    c1 /= (float)n;
    c2 /= (float)n;
    T* dxs = dx + gi * n;
    for (int e = lane; e < n; e += 64) {
        const int c = g * cpg + e / hw;
        const float gm = ldf(gamma + c), bt = ldf(beta + c);
        const float xh = (ldf(xs + e) - mu) * rs;
        const float dh = ldf(dys + e) * gelu_erf_grad(xh * gm + bt);
        stf(dxs + e, rs * (dh * gm - c1 - xh * c2));
    }
}
extern "C" int db1_groupnorm_gelu_bwd(const void* dy, const void* x, const void* gamma, const void* beta, const float* mean, const float* rstd,
                                      void* dx, float* dgamma_acc, float* dbeta_acc, int64_t N, int C, int hw, int groups, int dt,
                                      int dtParam, void* stream) {
    if (!db1_dt_ok(dt) || !db1_dt_ok(dtParam)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "groupnorm_gelu_bwd: dtype");
    if (N <= 0 || C <= 0 || hw <= 0 || groups <= 0 || C % groups) DB1_FAIL(DB1_ERR_BAD_SHAPE, "groupnorm_gelu_bwd: shape");
    const int64_t ng = N * groups;
    dim3 g((unsigned)((ng + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
#define L_(T, TP) gn_gelu_bwd_kernel<T, TP><<<g, 256, 0, st>>>((const T*)dy, (const T*)x, (const TP*)gamma, (const TP*)beta, mean, rstd, (T*)dx, dgamma_acc, dbeta_acc, ng, groups, C / groups, hw)
    if (dt == DB1_F32 && dtParam == DB1_F32) L_(float, float);
    else if (dt == DB1_BF16 && dtParam == DB1_BF16) L_(bf16_t, bf16_t);
    else if (dt == DB1_BF16) L_(bf16_t, float);
    else L_(float, bf16_t);
#undef L_
    DB1_CHECK_LAUNCH("groupnorm_gelu_bwd");
    return DB1_OK;
}
