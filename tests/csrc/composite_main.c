/* A plain-C host of the DB1 C ABI (include/db1_hip.h): no Python, no torch -- hipMalloc / hipMemcpy and two composite entry points.
 *   db1_grad_norm_sq     : sum of squares of a float32 gradient vector, against the host's own sum
 *   db1_patch_embed_fwd  : one 3 x 16 x 16 image through the patch embedder with weights chosen so that the result is known in closed
 *                          form: conv weights 0 -> both convolution outputs equal their biases; conv1.bias = c (per channel),
 *                          residual biases 0 -> residual sum y[c, :, :] = conv1.bias[c]; projection weight = 1/16384 everywhere,
 *                          bias 0.5 -> every output = mean(conv1.bias) + 0.5.
 * Built and run by tests/test_composite_gpu.py::test_plain_c_host_drives_the_composites. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "db1_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_DB1(x) do { int s_ = (x); if (s_ != 0) { printf("db1 status %d: %s\n", s_, db1_last_error()); return 3; } } while (0)

static uint16_t f2bf(float f) {   /* round to nearest even */
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static void* dev_bf16(const float* src, size_t n) {
    uint16_t* h = (uint16_t*)malloc(n * 2);
    for (size_t i = 0; i < n; i++) h[i] = f2bf(src[i]);
    void* d = NULL;
    if (hipMalloc(&d, n * 2) != hipSuccess) return NULL;
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    free(h);
    return d;
}

int main(void) {
    if (!db1_device_is_gfx950()) { printf("no gfx950 device\n"); return 1; }
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    /* ---- db1_grad_norm_sq */
    {
        const int64_t n = 1 << 20;
        float* h = (float*)malloc(n * 4);
        double want = 0.0;
        for (int64_t i = 0; i < n; i++) { h[i] = (float)((i * 2654435761u % 2001) - 1000) * 1e-3f; want += (double)h[i] * h[i]; }
        float *d = NULL, *acc = NULL, got = -1.f;
        void* nws = NULL;
        const int64_t nws_b = db1_grad_norm_sq_workspace_bytes(n);
        CHECK_HIP(hipMalloc((void**)&d, n * 4));
        CHECK_HIP(hipMalloc((void**)&acc, 4));
        CHECK_HIP(hipMalloc(&nws, nws_b));
        CHECK_HIP(hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice));
        CHECK_DB1(db1_grad_norm_sq(d, acc, n, 0 /* DB1_F32 */, nws, nws_b, st));
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(&got, acc, 4, hipMemcpyDeviceToHost));
        if (fabs(got - want) > 1e-5 * want) { printf("grad_norm_sq: got %g want %g\n", got, want); return 4; }
        printf("grad_norm_sq ok (%g)\n", got);
        free(h);
    }
    /* ---- db1_patch_embed_fwd */
    {
        const int n_img = 1, C = 3, Hh = 16, Ww = 16, p = 16, d = 64, N = 1;
        float* px = (float*)malloc(C * Hh * Ww * 4);
        for (int i = 0; i < C * Hh * Ww; i++) px[i] = (float)(i % 97);
        float w1[64 * 3 * 9] = {0}, w2[64 * 64 * 9], b1[64], zeros64[64] = {0}, ones64[64];
        memset(w2, 0, sizeof(w2));
        double mean_b1 = 0.0;
        for (int c = 0; c < 64; c++) { b1[c] = bf2f(f2bf(0.01f * (float)(c - 20))); ones64[c] = 1.f; mean_b1 += b1[c] / 64.0; }
        float* wp = (float*)malloc((size_t)d * 64 * 256 * 4);
        float* bp = (float*)malloc(d * 4);
        for (size_t i = 0; i < (size_t)d * 64 * 256; i++) wp[i] = 1.f / 16384.f;
        for (int i = 0; i < d; i++) bp[i] = 0.5f;
        const void* weights[12];
        weights[0] = dev_bf16(w1, 64 * 3 * 9);  weights[1] = dev_bf16(b1, 64);
        weights[2] = dev_bf16(ones64, 64);      weights[3] = dev_bf16(zeros64, 64);
        weights[4] = dev_bf16(w2, 64 * 64 * 9); weights[5] = dev_bf16(zeros64, 64);
        weights[6] = dev_bf16(ones64, 64);      weights[7] = dev_bf16(zeros64, 64);
        weights[8] = dev_bf16(w2, 64 * 64 * 9); weights[9] = dev_bf16(zeros64, 64);
        weights[10] = dev_bf16(wp, (size_t)d * 64 * 256); weights[11] = dev_bf16(bp, d);
        float* dpx = NULL;
        void *emb = NULL, *save = NULL, *ws = NULL;
        const int64_t nsave = db1_patch_embed_save_bytes(n_img, C, Hh, Ww, p), nws = db1_patch_embed_workspace_bytes(n_img, C, Hh, Ww, p, d, 0);
        CHECK_HIP(hipMalloc((void**)&dpx, C * Hh * Ww * 4));
        CHECK_HIP(hipMemcpy(dpx, px, C * Hh * Ww * 4, hipMemcpyHostToDevice));
        CHECK_HIP(hipMalloc(&emb, N * d * 2));
        CHECK_HIP(hipMalloc(&save, nsave));
        CHECK_HIP(hipMalloc(&ws, nws));
        CHECK_DB1(db1_patch_embed_fwd(dpx, weights, emb, save, n_img, C, Hh, Ww, p, d, ws, nws, st));
        CHECK_HIP(hipStreamSynchronize(st));
        uint16_t out[64];
        CHECK_HIP(hipMemcpy(out, emb, N * d * 2, hipMemcpyDeviceToHost));
        const double want = mean_b1 + 0.5;
        for (int i = 0; i < d; i++)
            if (fabs(bf2f(out[i]) - want) > 1e-2) { printf("patch_embed_fwd: out[%d] = %g, want %g\n", i, bf2f(out[i]), want); return 5; }
        /* a workspace that is too small is a status, not a crash */
        if (db1_patch_embed_fwd(dpx, weights, emb, save, n_img, C, Hh, Ww, p, d, ws, 256, st) == 0) { printf("small workspace accepted\n"); return 6; }
        printf("patch_embed_fwd ok (%g)\n", want);
    }
    return 0;
}
