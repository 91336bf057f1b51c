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
        sb = wave_sum(sb);   // (the parameter gradients are formed by gn_param_grad_kernel: one workgroup per channel, fixed order, no atomics)
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
// dgamma[c] += sum_{n, e} dh * xhat, dbeta[c] += sum_{n, e} dh with dh = dy * gelu'(xhat * gamma + beta): ONE workgroup per channel walks the
// samples in order (thread t: elements t, t + 256, ... of every sample), block sum in a fixed order: deterministic (generic / fp32 parity path)
template <typename T, typename TP>
__global__ __launch_bounds__(256) void gn_param_grad_kernel(const T* __restrict__ dy, const T* __restrict__ x, const TP* __restrict__ gamma,
                                                            const TP* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* dgamma, float* dbeta, int64_t N, int C, int groups, int hw) {
    __shared__ float sm[4];
    const int c = blockIdx.x, cpg = C / groups, g = c / cpg;
    const float gm = ldf(gamma + c), bt = ldf(beta + c);
    float sg = 0.f, sb = 0.f;
    for (int64_t n = 0; n < N; n++) {
        const float mu = mean[n * groups + g], rs = rstd[n * groups + g];
        const T* xs = x + (n * C + c) * (int64_t)hw;
        const T* dys = dy + (n * C + c) * (int64_t)hw;
        for (int e = threadIdx.x; e < hw; e += 256) {
            const float xh = (ldf(xs + e) - mu) * rs;
            const float dh = ldf(dys + e) * gelu_erf_grad(xh * gm + bt);
            sg += dh * xh;
            sb += dh;
        }
    }
    sg = block_sum256(sg, sm);
    __syncthreads();
    sb = block_sum256(sb, sm);
    if (threadIdx.x == 0) { dgamma[c] += sg; dbeta[c] += sb; }
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
#define LP_(T, TP) gn_param_grad_kernel<T, TP><<<(unsigned)C, 256, 0, st>>>((const T*)dy, (const T*)x, (const TP*)gamma, (const TP*)beta, mean, rstd, dgamma_acc, dbeta_acc, N, C, groups, hw)
    if (dt == DB1_F32 && dtParam == DB1_F32) LP_(float, float);
    else if (dt == DB1_BF16 && dtParam == DB1_BF16) LP_(bf16_t, bf16_t);
    else if (dt == DB1_BF16) LP_(bf16_t, float);
    else LP_(float, bf16_t);
#undef LP_
    DB1_CHECK_LAUNCH("groupnorm_gelu_bwd parameters");
    return DB1_OK;
}

// =====================================================================================================================
// Channels-last ("NHWC") pipeline: activations are [N, p*p, C] (what the convolution GEMMs produce), the column matrix is
// tap-major, cols[(n,y,x)][(ky*3+kx)*C + c].  Every access below is a 16-byte vector of 8 consecutive channels, so a pixel's
// 64 channels are one 128-byte line; the NCHW kernels above read/wrote 2-byte elements 1152 bytes apart (col2im ran at
// ~0.15 TB/s: 28.8 ms per call on the RL workload, 21 % of its step).  GroupNorm(32, 64) has its 2-channel groups adjacent in
// this layout.  Only the k=s=16 projection still wants the (c, y, x) flattening of the reference's weight: one shuffle each way.
// =====================================================================================================================
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void patch_normalize_nhwc_kernel(const TI* __restrict__ pix, TO* __restrict__ out, int64_t n_pc, int C, int Himg,
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
    TO* dst = out + patch * n * C + c;
    for (int e = lane; e < n; e += 64) stf(dst + (int64_t)e * C, (ldf(src + (int64_t)(e / p) * Wimg + (e % p)) - mean) * inv);
}
extern "C" int db1_patch_normalize_nhwc(const void* pixels, void* patches, int n_img, int C, int Himg, int Wimg, int p, int dtIn, int dtOut,
                                        void* stream) {
    if (!db1_dt_ok(dtIn) || !db1_dt_ok(dtOut)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "patch_normalize_nhwc: dtype");
    if (n_img <= 0 || C <= 0 || p <= 1 || Himg % p || Wimg % p) DB1_FAIL(DB1_ERR_BAD_SHAPE, "patch_normalize_nhwc: image %dx%d not divisible by patch %d", Himg, Wimg, p);
    const int64_t n_pc = (int64_t)n_img * (Himg / p) * (Wimg / p) * C;
    dim3 g((unsigned)((n_pc + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
#define L_(A, B) patch_normalize_nhwc_kernel<A, B><<<g, 256, 0, st>>>((const A*)pixels, (B*)patches, n_pc, C, Himg, Wimg, p)
    if (dtIn == DB1_F32 && dtOut == DB1_F32) L_(float, float);
    else if (dtIn == DB1_F32) L_(float, bf16_t);
    else if (dtOut == DB1_F32) L_(bf16_t, float);
    else L_(bf16_t, bf16_t);
#undef L_
    DB1_CHECK_LAUNCH("patch_normalize_nhwc");
    return DB1_OK;
}

// im2col, tap-major.  VEC = channels per thread (8 x bf16 / 4 x f32 = 16 bytes when C allows, else 1)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const T* __restrict__ x, T* __restrict__ cols, int64_t npix, int C, int p, int K) {
    const int cv = C / VEC, per_pix = 9 * cv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < npix * per_pix; idx += (int64_t)gridDim.x * 256) {
        const int64_t pix = idx / per_pix;
        const int rem = (int)(idx - pix * per_pix);
        const int tap = rem / cv, c0 = (rem - tap * cv) * VEC;
        const int xw = (int)(pix % p), yh = (int)((pix / p) % p);
        const int yy = yh + tap / 3 - 1, xx = xw + tap % 3 - 1;
        const bool in = yy >= 0 && yy < p && xx >= 0 && xx < p;
        T* dst = cols + pix * K + tap * C + c0;
        const T* src = x + (pix + (int64_t)(yy - yh) * p + (xx - xw)) * C + c0;
        if (VEC * sizeof(T) == 16) {
            *reinterpret_cast<uint4*>(dst) = in ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
        } else {
            dst[0] = in ? src[0] : (T)0;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void zero_pad_cols_kernel(T* __restrict__ cols, int64_t npix, int K0, int K) {
    const int padw = K - K0;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < npix * padw; idx += (int64_t)gridDim.x * 256)
        cols[(idx / padw) * K + K0 + idx % padw] = (T)0;
}
static inline unsigned vis_grid(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (unsigned)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}
extern "C" int db1_im2col3x3_nhwc(const void* x, void* cols, int64_t N, int C, int p, int kpad, int dt, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "im2col_nhwc: dtype");
    if (N <= 0 || C <= 0 || p <= 0 || kpad < C * 9) DB1_FAIL(DB1_ERR_BAD_SHAPE, "im2col_nhwc: shape");
    const int64_t npix = N * p * p;
    const int V = dt == DB1_F32 ? 4 : 8;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (C % V) == 0 && (kpad % V) == 0 && db1_aligned16(x) && db1_aligned16(cols);
    DB1_DISPATCH_DT(dt, T, {
        constexpr int VV = 16 / sizeof(T);
        if (vec) im2col_nhwc_kernel<T, VV><<<vis_grid(npix * 9 * (C / VV)), 256, 0, st>>>((const T*)x, (T*)cols, npix, C, p, kpad);
        else im2col_nhwc_kernel<T, 1><<<vis_grid(npix * 9 * C), 256, 0, st>>>((const T*)x, (T*)cols, npix, C, p, kpad);
        if (kpad > C * 9) zero_pad_cols_kernel<T><<<vis_grid(npix * (kpad - C * 9)), 256, 0, st>>>((T*)cols, npix, C * 9, kpad);
    });
    DB1_CHECK_LAUNCH("im2col_nhwc");
    return DB1_OK;
}

// col2im, gather form (no atomics): dx[pix][c] = sum over the taps t of dcols[pix - offset(t)][t*C + c]
template <typename T, int VEC>
__global__ __launch_bounds__(256) void col2im_nhwc_kernel(const T* __restrict__ dcols, T* __restrict__ dx, int64_t npix, int C, int p, int K) {
    const int cv = C / VEC;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < npix * cv; idx += (int64_t)gridDim.x * 256) {
        const int64_t pix = idx / cv;
        const int c0 = (int)(idx - pix * cv) * VEC;
        const int xw = (int)(pix % p), yh = (int)((pix / p) % p);
        float a[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) a[j] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int oy = yh - tap / 3 + 1, ox = xw - tap % 3 + 1;  // the output pixel whose window position `tap` is this input pixel
            if (oy >= 0 && oy < p && ox >= 0 && ox < p) {
                const T* src = dcols + (pix + (int64_t)(oy - yh) * p + (ox - xw)) * K + tap * C + c0;
                if (VEC * sizeof(T) == 16) {
                    Vec16<T> v;
                    v.load(src);
#pragma unroll
                    for (int j = 0; j < VEC; j++) a[j] += v.v[j];
                } else {
                    a[0] += ldf(src);
                }
            }
        }
        if (VEC * sizeof(T) == 16) {
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VEC; j++) o.v[j] = a[j];
            o.store(dx + pix * C + c0);
        } else {
            stf(dx + pix * C + c0, a[0]);
        }
    }
}
extern "C" int db1_col2im3x3_nhwc(const void* dcols, void* dx, int64_t N, int C, int p, int kpad, int dt, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "col2im_nhwc: dtype");
    if (N <= 0 || C <= 0 || p <= 0 || kpad < C * 9) DB1_FAIL(DB1_ERR_BAD_SHAPE, "col2im_nhwc: shape");
    const int64_t npix = N * p * p;
    const int V = dt == DB1_F32 ? 4 : 8;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (C % V) == 0 && (kpad % V) == 0 && db1_aligned16(dcols) && db1_aligned16(dx);
    DB1_DISPATCH_DT(dt, T, {
        constexpr int VV = 16 / sizeof(T);
        if (vec) col2im_nhwc_kernel<T, VV><<<vis_grid(npix * (C / VV)), 256, 0, st>>>((const T*)dcols, (T*)dx, npix, C, p, kpad);
        else col2im_nhwc_kernel<T, 1><<<vis_grid(npix * C), 256, 0, st>>>((const T*)dcols, (T*)dx, npix, C, p, kpad);
    });
    DB1_CHECK_LAUNCH("col2im_nhwc");
    return DB1_OK;
}

// conv weight [Cout, Cin, 3, 3] <-> GEMM operand [Cout, kpad] in tap-major column order (zero padded)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void conv_weight_permute_kernel(const TI* __restrict__ w, TO* __restrict__ wp, int Cout, int Cin, int K) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cout * K) return;
    const int o = idx / K, col = idx % K;
    float v = 0.f;
    if (col < 9 * Cin) { const int tap = col / Cin, c = col % Cin; v = ldf(w + ((int64_t)o * Cin + c) * 9 + tap); }
    stf(wp + idx, v);
}
extern "C" int db1_conv_weight_permute(const void* w, void* wp, int Cout, int Cin, int kpad, int dtIn, int dtOut, void* stream) {
    if (!db1_dt_ok(dtIn) || !db1_dt_ok(dtOut)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "conv_weight_permute: dtype");
    if (Cout <= 0 || Cin <= 0 || kpad < 9 * Cin) DB1_FAIL(DB1_ERR_BAD_SHAPE, "conv_weight_permute: shape");
    const unsigned g = (unsigned)((Cout * kpad + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
#define L_(A, B) conv_weight_permute_kernel<A, B><<<g, 256, 0, st>>>((const A*)w, (B*)wp, Cout, Cin, kpad)
    if (dtIn == DB1_F32 && dtOut == DB1_F32) L_(float, float);
    else if (dtIn == DB1_F32) L_(float, bf16_t);
    else if (dtOut == DB1_F32) L_(bf16_t, float);
    else L_(bf16_t, bf16_t);
#undef L_
    DB1_CHECK_LAUNCH("conv_weight_permute");
    return DB1_OK;
}
// data-gradient operand: wp[c, tap*Cout + o] = w[o, c, tap]  (the transposed weight of the same tap-major layout; conv_implicit.hip)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void conv_weight_permute_t_kernel(const TI* __restrict__ w, TO* __restrict__ wp, int Cout, int Cin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * 9 * Cout) return;
    const int c = idx / (9 * Cout), rem = idx % (9 * Cout), tap = rem / Cout, o = rem % Cout;
    stf(wp + idx, ldf(w + ((int64_t)o * Cin + c) * 9 + tap));
}
extern "C" int db1_conv_weight_permute_t(const void* w, void* wp, int Cout, int Cin, int dtIn, int dtOut, void* stream) {
    if (!db1_dt_ok(dtIn) || !db1_dt_ok(dtOut)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "conv_weight_permute_t: dtype");
    if (Cout <= 0 || Cin <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "conv_weight_permute_t: shape");
    const unsigned g = (unsigned)((Cin * 9 * Cout + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
#define L_(A, B) conv_weight_permute_t_kernel<A, B><<<g, 256, 0, st>>>((const A*)w, (B*)wp, Cout, Cin)
    if (dtIn == DB1_F32 && dtOut == DB1_F32) L_(float, float);
    else if (dtIn == DB1_F32) L_(float, bf16_t);
    else if (dtOut == DB1_F32) L_(bf16_t, float);
    else L_(bf16_t, bf16_t);
#undef L_
    DB1_CHECK_LAUNCH("conv_weight_permute_t");
    return DB1_OK;
}
// g[o, c, tap] += gp[o, tap*Cin + c]   (float32 both: the weight gradient computed in tap-major order goes back to the parameter layout)
__global__ __launch_bounds__(256) void conv_wgrad_unpermute_kernel(const float* __restrict__ gp, float* __restrict__ g, int Cout, int Cin, int K) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cout * Cin * 9) return;
    const int o = idx / (Cin * 9), rem = idx % (Cin * 9), c = rem / 9, tap = rem % 9;
    g[idx] += gp[(int64_t)o * K + tap * Cin + c];
}
extern "C" int db1_conv_wgrad_unpermute(const float* gp, float* g_acc, int Cout, int Cin, int kpad, void* stream) {
    if (Cout <= 0 || Cin <= 0 || kpad < 9 * Cin || !gp || !g_acc) DB1_FAIL(DB1_ERR_BAD_SHAPE, "conv_wgrad_unpermute: shape");
    conv_wgrad_unpermute_kernel<<<(unsigned)((Cout * Cin * 9 + 255) / 256), 256, 0, (hipStream_t)stream>>>(gp, g_acc, Cout, Cin, kpad);
    DB1_CHECK_LAUNCH("conv_wgrad_unpermute");
    return DB1_OK;
}

// GroupNorm + GELU, channels-last, specialised for the embedder's shape family: C = 64 channels, hw = 256 pixels per sample
// (one 256-thread workgroup per sample; thread t owns the 8-channel vector t % 8 of pixels t / 8 + 32 k, k = 0..7, i.e. 64 values
// in registers = 4 groups of cpg = 2 channels when groups = 32).  Group statistics: per-thread partials -> LDS -> one thread per
// group adds the 32 partials in a fixed order (deterministic), two passes (mean, then centred sum) from registers.
#define GNV_C 64
#define GNV_HW 256
template <typename T, typename TP, bool BWD>
__global__ __launch_bounds__(256) void gn_gelu_nhwc_kernel(const T* __restrict__ x, const T* __restrict__ dy, const TP* __restrict__ gamma,
                                                           const TP* __restrict__ beta, T* __restrict__ out, float* __restrict__ mean,
                                                           float* __restrict__ rstd, float* dgamma, float* dbeta, int cpg, float eps,
                                                           float* __restrict__ pgrad,     // bwd: per-sample (dgamma | dbeta) rows [N][128]
                                                           const T* __restrict__ res) {   // bwd: added to dx in fp32 before the one rounding (the residual branch's gradient), or null
    __shared__ float part[256][8];
    __shared__ float stat[2][GNV_C];   // per channel: (mean, rstd) fwd / (c1, c2) bwd, replicated over the channels of a group
    const int t = threadIdx.x, chunk = t & 7, prow = t >> 3;
    const int64_t n = blockIdx.x;
    const int groups = GNV_C / cpg;
    const T* xs = x + n * GNV_HW * GNV_C + chunk * 8;
    float v[8][8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        Vec16<T> a;
        a.load(xs + (int64_t)(prow + 32 * k) * GNV_C);
#pragma unroll
        for (int j = 0; j < 8; j++) v[k][j] = a.v[j];
    }
    // fixed-order reduction of one float per (thread, channel) over the 32 threads that own the same channel chunk, then over the
    // cpg channels of each group; result broadcast per channel into dst[0..63]
    auto group_reduce = [&](const float* mine, float* dst) {
#pragma unroll
        for (int j = 0; j < 8; j++) part[t][j] = mine[j];
        __syncthreads();
        if (t < GNV_C) {
            float s = 0.f;
            const int g0 = (t / cpg) * cpg;          // first channel of my group
            for (int c = g0; c < g0 + cpg; c++)
                for (int r = 0; r < 32; r++) s += part[r * 8 + (c >> 3)][c & 7];
            dst[t] = s;
        }
        __syncthreads();
    };
    float loc[8];
    const float cnt = (float)(cpg * GNV_HW);
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < 8; j++) { loc[j] = 0.f; for (int k = 0; k < 8; k++) loc[j] += v[k][j]; }
        group_reduce(loc, stat[0]);
        float mu[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { mu[j] = stat[0][chunk * 8 + j] / cnt; loc[j] = 0.f; for (int k = 0; k < 8; k++) { const float d = v[k][j] - mu[j]; loc[j] += d * d; } }
        group_reduce(loc, stat[1]);
        float rs[8], gm[8], bt[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { rs[j] = rsqrtf(stat[1][chunk * 8 + j] / cnt + eps); gm[j] = ldf(gamma + chunk * 8 + j); bt[j] = ldf(beta + chunk * 8 + j); }
        if (prow == 0) {
#pragma unroll
            for (int j = 0; j < 8; j += 1) {
                const int c = chunk * 8 + j;
                if (c % cpg == 0) { mean[n * groups + c / cpg] = mu[j]; rstd[n * groups + c / cpg] = rs[j]; }
            }
        }
        T* ys = out + n * GNV_HW * GNV_C + chunk * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < 8; j++) o.v[j] = gelu_fwd_t<T>((v[k][j] - mu[j]) * rs[j] * gm[j] + bt[j]);
            o.store(ys + (int64_t)(prow + 32 * k) * GNV_C);
        }
    } else {
        float mu[8], rs[8], gm[8], bt[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = chunk * 8 + j;
            mu[j] = mean[n * groups + c / cpg]; rs[j] = rstd[n * groups + c / cpg]; gm[j] = ldf(gamma + c); bt[j] = ldf(beta + c);
        }
        const T* dys = dy + n * GNV_HW * GNV_C + chunk * 8;
        float sg[8], sb[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { sg[j] = 0.f; sb[j] = 0.f; }
        float dh[8][8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            Vec16<T> d;
            d.load(dys + (int64_t)(prow + 32 * k) * GNV_C);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float xh = (v[k][j] - mu[j]) * rs[j];
                float y_, dg;
                gelu_both_t<T>(xh * gm[j] + bt[j], y_, dg);
                const float h = d.v[j] * dg;
                v[k][j] = xh;      // keep xhat
                dh[k][j] = h;
                sg[j] += h * xh;
                sb[j] += h;
            }
        }
        // per-channel parameter gradients of this sample (sum over the 32 threads of the chunk), accumulated with atomics across samples
#pragma unroll
        for (int j = 0; j < 8; j++) part[t][j] = sg[j];
        __syncthreads();
        if (t < GNV_C) { float s = 0.f; for (int r = 0; r < 32; r++) s += part[r * 8 + (t >> 3)][t & 7]; pgrad[n * (2 * GNV_C) + t] = s; stat[1][t] = s * ldf(gamma + t); }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) part[t][j] = sb[j];
        __syncthreads();
        if (t < GNV_C) { float s = 0.f; for (int r = 0; r < 32; r++) s += part[r * 8 + (t >> 3)][t & 7]; pgrad[n * (2 * GNV_C) + GNV_C + t] = s; stat[0][t] = s * ldf(gamma + t); }
        __syncthreads();
        // group sums c1 = sum(dh * gamma), c2 = sum(dh * gamma * xhat) over the cpg channels of the group
        __shared__ float cg[2][GNV_C];
        if (t < GNV_C) {
            const int g0 = (t / cpg) * cpg;
            float a1 = 0.f, a2 = 0.f;
            for (int c = g0; c < g0 + cpg; c++) { a1 += stat[0][c]; a2 += stat[1][c]; }
            cg[0][t] = a1 / cnt;
            cg[1][t] = a2 / cnt;
        }
        __syncthreads();
        T* dxs = out + n * GNV_HW * GNV_C + chunk * 8;
        const T* rss = res ? res + n * GNV_HW * GNV_C + chunk * 8 : nullptr;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            Vec16<T> o, r;
            if (rss) r.load(rss + (int64_t)(prow + 32 * k) * GNV_C);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float g = rs[j] * (dh[k][j] * gm[j] - cg[0][chunk * 8 + j] - v[k][j] * cg[1][chunk * 8 + j]);
                o.v[j] = rss ? g + (float)r.v[j] : g;
            }
            o.store(dxs + (int64_t)(prow + 32 * k) * GNV_C);
        }
    }
}
extern "C" int db1_groupnorm_gelu_nhwc_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int64_t N,
                                           int C, int hw, int groups, float eps, int dt, int dtParam, void* stream) {
    if (dt != DB1_BF16 || !db1_dt_ok(dtParam)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "groupnorm_gelu_nhwc_fwd: bf16 activations only");
    if (N <= 0 || C != GNV_C || hw != GNV_HW || groups <= 0 || C % groups) DB1_FAIL(DB1_ERR_UNSUPPORTED, "groupnorm_gelu_nhwc_fwd: needs C=64, hw=256 (got %d, %d)", C, hw);
    hipStream_t st = (hipStream_t)stream;
    if (dtParam == DB1_BF16) gn_gelu_nhwc_kernel<bf16_t, bf16_t, false><<<(unsigned)N, 256, 0, st>>>((const bf16_t*)x, nullptr, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, mean, rstd, nullptr, nullptr, C / groups, eps, nullptr, nullptr);
    else gn_gelu_nhwc_kernel<bf16_t, float, false><<<(unsigned)N, 256, 0, st>>>((const bf16_t*)x, nullptr, (const float*)gamma, (const float*)beta, (bf16_t*)y, mean, rstd, nullptr, nullptr, C / groups, eps, nullptr, nullptr);
    DB1_CHECK_LAUNCH("groupnorm_gelu_nhwc_fwd");
    return DB1_OK;
}
extern "C" int64_t db1_groupnorm_gelu_nhwc_bwd_workspace_bytes(int64_t N) {   // per-sample parameter-gradient rows + the ordered column sum's partials
    return N > 0 ? ((N * 2 * GNV_C * (int64_t)sizeof(float) + 255) & ~(int64_t)255) + db1_colsum_acc_workspace_bytes(N, 2 * GNV_C) : 0;
}
extern "C" int db1_groupnorm_gelu_nhwc_bwd(const void* dy, const void* x, const void* gamma, const void* beta, const float* mean, const float* rstd,
                                           void* dx, const void* res, float* dgamma_acc, float* dbeta_acc, int64_t N, int C, int hw, int groups, int dt,
                                           int dtParam, void* ws, int64_t ws_bytes, void* stream) {
    if (dt != DB1_BF16 || !db1_dt_ok(dtParam)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "groupnorm_gelu_nhwc_bwd: bf16 activations only");
    if (N <= 0 || C != GNV_C || hw != GNV_HW || groups <= 0 || C % groups) DB1_FAIL(DB1_ERR_UNSUPPORTED, "groupnorm_gelu_nhwc_bwd: needs C=64, hw=256 (got %d, %d)", C, hw);
    hipStream_t st = (hipStream_t)stream;
    // every sample leaves its (dgamma | dbeta) row in the caller's workspace and the rows are summed in a fixed order (no atomic form).
    // (dgamma_acc and dbeta_acc must then be the two halves of ONE [128] accumulator, or are summed by two strided column sums below.)
    const int64_t rows_b = (N * 2 * GNV_C * (int64_t)sizeof(float) + 255) & ~(int64_t)255;
    DB1_NEED_WS(ws, ws_bytes, db1_groupnorm_gelu_nhwc_bwd_workspace_bytes(N), "groupnorm_gelu_nhwc_bwd");
    float* pgrad = (float*)ws;
    if (dtParam == DB1_BF16) gn_gelu_nhwc_kernel<bf16_t, bf16_t, true><<<(unsigned)N, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)dx, const_cast<float*>(mean), const_cast<float*>(rstd), dgamma_acc, dbeta_acc, C / groups, 0.f, pgrad, (const bf16_t*)res);
    else gn_gelu_nhwc_kernel<bf16_t, float, true><<<(unsigned)N, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, (const float*)gamma, (const float*)beta, (bf16_t*)dx, const_cast<float*>(mean), const_cast<float*>(rstd), dgamma_acc, dbeta_acc, C / groups, 0.f, pgrad, (const bf16_t*)res);
    if (pgrad) {
        DB1_CHECK_LAUNCH("groupnorm_gelu_nhwc_bwd");
        void* cws = (char*)ws + rows_b;
        int rc = db1_colsum_acc(pgrad, dgamma_acc, N, GNV_C, 2 * GNV_C, DB1_F32, cws, ws_bytes - rows_b, stream);
        if (rc) return rc;
        return db1_colsum_acc(pgrad + GNV_C, dbeta_acc, N, GNV_C, 2 * GNV_C, DB1_F32, cws, ws_bytes - rows_b, stream);
    }
    DB1_CHECK_LAUNCH("groupnorm_gelu_nhwc_bwd");
    return DB1_OK;
}
