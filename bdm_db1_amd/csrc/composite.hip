// Composite C-ABI entry points (db1_patch_embed_{fwd,bwd}, db1_relattn_{fwd,bwd}, db1_lmhead_ce_bwd, db1_grad_norm_sq): each
// sequences the library's own launches for one block of the reference's forward / backward, so that a host in ANY language drives the hot
// path with a handful of calls instead of re-implementing the orchestration of bdm_db1_amd/model/transformer_xl.py.  Same conventions as
// the rest of the ABI: device pointers owned by the caller, scratch through (ws, ws_bytes) with a size query, asynchronous on `stream`,
// int status + db1_last_error(), nothing allocated / freed / synchronised here.  bf16 activations (the production path); the fp32 parity
// path stays on the per-op entry points.
#include "db1_common.h"

static inline int64_t al256(int64_t x) { return (x + 255) & ~(int64_t)255; }
#define CK(expr)                \
    do {                        \
        const int rc_ = (expr); \
        if (rc_) return rc_;    \
    } while (0)

// ======================================================================================================= gradient norm
/* acc[0] = sum(g^2) over a flat gradient segment (bf16 or fp32): the global-norm clip's reduction (train_config.py:211-215), one call per
 * arena.  Deterministic: per-workgroup partial sums through the workspace, added in a fixed order (no atomics). */
extern "C" int db1_sumsq_det(const void* x, float* acc, int64_t n, int dt, int overwrite, void* ws, int64_t ws_bytes, void* stream);
extern "C" int64_t db1_sumsq_det_workspace_bytes(int64_t n);
extern "C" int64_t db1_grad_norm_sq_workspace_bytes(int64_t n) { return db1_sumsq_det_workspace_bytes(n); }
extern "C" int db1_grad_norm_sq(const void* g, float* acc, int64_t n, int dt, void* ws, int64_t ws_bytes, void* stream) {
    if (!g || !acc || n <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "grad_norm_sq: null buffer / empty segment");
    return db1_sumsq_det(g, acc, n, dt, 1, ws, ws_bytes, stream);
}

// ======================================================================================================= head backward
// backward of the tied head + masked CE from the (lse, sums) a previous db1_lmhead_ce_fwd left (transformer_xl.py:593-613): the logits are
// recomputed chunk by chunk (no tokens x vocabulary tensor), turned into dlogits in place and consumed by the two gradient GEMMs.
extern "C" int64_t db1_lmhead_ce_bwd_workspace_bytes(int64_t T_, int n_w_rows, int d, int chunk_rows, int dt) {
    return db1_lmhead_ce_workspace_bytes(T_, n_w_rows, d, chunk_rows, dt, 1);
}
extern "C" int db1_lmhead_ce_bwd(const void* h, const void* W, const int64_t* labels, const float* mask, const float* lse, const float* sums, void* dh,
                                 float* dW_acc, float beta_dw, float gscale, int64_t T_, int V, int n_w_rows, int d, int chunk_rows, int dt, void* ws,
                                 int64_t ws_bytes, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "lmhead_ce_bwd: dtype");
    if (T_ <= 0 || V <= 0 || n_w_rows < V || d <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "lmhead_ce_bwd: T=%lld V=%d rows=%d d=%d", (long long)T_, V, n_w_rows, d);
    if (!h || !W || !labels || !mask || !lse || !sums || !dh || !dW_acc) DB1_FAIL(DB1_ERR_BAD_SHAPE, "lmhead_ce_bwd: null buffer");
    int64_t chunk = chunk_rows > 0 ? chunk_rows : 16384;
    if (chunk > T_) chunk = T_;
    const int es = dt == DB1_F32 ? 4 : 2;
    const int64_t logits_b = al256(chunk * (int64_t)n_w_rows * es);
    DB1_NEED_WS(ws, ws_bytes, db1_lmhead_ce_bwd_workspace_bytes(T_, n_w_rows, d, chunk_rows, dt), "lmhead_ce_bwd");
    char* logits = (char*)ws;
    void* gws = logits + logits_b;
    const int64_t gws_b = ws_bytes - logits_b;
    for (int64_t r0 = 0; r0 < T_; r0 += chunk) {
        const int rows = (int)((T_ - r0 < chunk) ? (T_ - r0) : chunk);
        const char* hc = (const char*)h + r0 * d * es;
        CK(db1_gemm_strided(hc, W, logits, nullptr, rows, n_w_rows, d, dt, dt, dt, 0, d, 1, 1, d, n_w_rows, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, 0.f, gws, gws_b, stream));
        CK(db1_masked_ce_bwd(logits, labels + r0, mask + r0, lse + r0, sums, logits, rows, V, n_w_rows, gscale, dt, stream));
        CK(db1_gemm_strided(logits, hc, dW_acc, nullptr, n_w_rows, d, rows, dt, dt, DB1_F32, 0, 1, n_w_rows, d, 1, d, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f,
                            r0 == 0 ? beta_dw : 1.f, gws, gws_b, stream));
        CK(db1_gemm_strided(logits, W, (char*)dh + r0 * d * es, nullptr, rows, d, n_w_rows, dt, dt, dt, 0, n_w_rows, 1, d, 1, d, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f,
                            0.f, gws, gws_b, stream));
    }
    return DB1_OK;
}

// ======================================================================================================= relative-position attention
// RelPartialLearnableMultiHeadAttn score / softmax / P.V (transformer_xl.py:160-225) on the packed projections qkv [B, L, 3, H, D] (bf16,
// d_head 128, L % 128 == 0): forward = q+u / q+v + flash forward; backward = flash backward (dq_k, dk, dv, dT) + dq = dq_k + dT.R with
// the u / v gradients + dR = dT^T.(q+v).  `probs` / `mblk` (optional, both or neither): the forward keeps its probabilities
// ([B*H, L/32, L/16, 512] bf16 + [B*H, L/32, L] fp32) and the backward recomputes nothing.
extern "C" int db1_relattn_fwd(const void* qkv, const void* u, const void* vb, const void* R, void* qu, void* qv, void* out, float* lse, void* probs,
                               float* mblk, int B, int L, int H, int D, int shift, float scale, void* stream) {
    if (!db1_relattn_flash_supported(B, L, H, D, DB1_BF16)) DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_fwd: needs bf16, d_head = 128, L %% 128 == 0");
    if (!qkv || !u || !vb || !R || !qu || !qv || !out || !lse) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_fwd: null buffer");
    const int64_t HD = (int64_t)H * D;
    CK(db1_relattn_add_head_bias(qkv, u, vb, qu, qv, B, L, L, H, D, DB1_BF16, DB1_BF16, stream));
    const bf16_t* k = (const bf16_t*)qkv + HD;
    const bf16_t* v = (const bf16_t*)qkv + 2 * HD;
    return db1_relattn_flash_fwd(qu, qv, k, v, 3 * HD, (int64_t)L * 3 * HD, R, out, lse, B, L, H, D, shift, scale, probs, mblk, stream);
}

extern "C" int64_t db1_relattn_bwd_workspace_bytes(int B, int L, int H, int D, int have_probs) {
    const int64_t HD = (int64_t)H * D;
    int64_t w = al256(db1_relattn_flash_bwd_workspace_bytes(B, L, H, have_probs));
    const int64_t a = al256(db1_relattn_dqr_workspace_bytes(L, H));
    const int64_t g = al256(db1_gemm_workspace_bytes(L, D, B * L, DB1_BF16, DB1_BF16, DB1_BF16, 1, L, HD, 1, HD, 1, H, 1));
    w = w > a ? w : a;
    w = w > g ? w : g;
    return w + al256((int64_t)B * H * L * sizeof(float));   // + delta
}
/* dqkv [B, L, 3, H, D] (written), dR [L, H, D] (written), du_acc / dvb_acc [H, D] float32 (+=).  dT [H, B, L, L] bf16 is caller scratch with the
 * contract of db1_relattn_flash_bwd: zero above the causal diagonal on entry (with the plain causal window every other entry is rewritten,
 * so one zero-initialised buffer serves all layers and steps). */
extern "C" int db1_relattn_bwd(const void* qkv, const void* qu, const void* qv, const void* R, const void* out, const void* dout, const float* lse,
                               const void* probs, const float* mblk, void* dqkv, void* dR, float* du_acc, float* dvb_acc, void* dT, int B, int L, int H,
                               int D, int shift, float scale, void* ws, int64_t ws_bytes, void* stream) {
    if (!db1_relattn_flash_supported(B, L, H, D, DB1_BF16) || !db1_relattn_dqr_supported(B, L, H, D, DB1_BF16))
        DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_bwd: needs bf16, d_head = 128, L %% 128 == 0");
    if (shift < L) DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_bwd: the composite covers the plain causal window (shift >= L); sliding windows: the per-op entry points");
    if (!qkv || !qu || !qv || !R || !out || !dout || !lse || !dqkv || !dR || !du_acc || !dvb_acc || !dT) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_bwd: null buffer");
    const int64_t need = db1_relattn_bwd_workspace_bytes(B, L, H, D, probs != nullptr);
    DB1_NEED_WS(ws, ws_bytes, need, "relattn_bwd");
    const int64_t HD = (int64_t)H * D;
    const int64_t delta_b = al256((int64_t)B * H * L * sizeof(float));
    float* delta = (float*)ws;
    void* w2 = (char*)ws + delta_b;
    const int64_t w2_b = ws_bytes - delta_b;
    const bf16_t* k = (const bf16_t*)qkv + HD;
    const bf16_t* v = (const bf16_t*)qkv + 2 * HD;
    bf16_t* dq = (bf16_t*)dqkv;
    CK(db1_relattn_flash_bwd(qu, qv, k, v, 3 * HD, (int64_t)L * 3 * HD, R, out, dout, lse, delta, dq, dq + HD, dq + 2 * HD, 3 * HD, (int64_t)L * 3 * HD, dT,
                             B, L, H, D, shift, scale, probs, mblk, w2, w2_b, stream));
    // dq = dq_k + dT.R in place, du += colsum(dq_k), dv_bias += colsum(dq_r)
    CK(db1_relattn_dqr_fused(dT, R, HD, dq, 3 * HD, (int64_t)L * 3 * HD, du_acc, dvb_acc, B, L, H, D, w2, w2_b, stream));
    // dR[dist, h, :] = sum_{b, i} dT[h, b, i, dist] (q+v)[b, i, h, :]: per head a [L x (B L)] x [(B L) x D] product, k-tiles above the diagonal skipped
    return db1_gemm_strided_tri(dT, qv, dR, nullptr, L, D, B * L, DB1_BF16, DB1_BF16, DB1_BF16, 0, 1, L, HD, 1, HD, 1, H, 1, (int64_t)B * L * L, 0, D, 0, D,
                                0, 1.f, 0.f, 2, L, w2, w2_b, stream);
}

// ======================================================================================================= image-patch embedder
// PatchEmbeddings.forward (vision_embedding.py:65-86) on 16 x 16 patches, bf16, channels-last inside: per-patch normalisation, conv1 (3x3,
// C -> 64, explicit tap-major columns), [GroupNorm(32, 64) + GELU + implicit 3x3 conv 64 -> 64] x 2, residual, projection (16 384 -> d).
// weights[12] / grads[12] in the order of the reference's state dict: conv1.{weight, bias}, residual_path.0.{weight, bias} (GroupNorm),
// residual_path.2.{weight, bias} (conv), residual_path.3.{weight, bias} (GroupNorm), residual_path.5.{weight, bias} (conv),
// projection.{weight, bias}; weights bf16 in the reference's layouts, gradients float32 accumulators (+=).
// `save` (db1_patch_embed_save_bytes) carries the forward's activations to the backward.
struct PeSizes {
    int64_t N, rows, kp1;
    int64_t cols1, c1, a0, c2, a1, y, stat;           // `save` blocks (bytes)
    int64_t off_cols1, off_c1, off_a0, off_c2, off_a1, off_y, off_m0, off_r0, off_m1, off_r1, save_total;
};
static PeSizes pe_sizes(int n_img, int C, int Himg, int Wimg, int p) {
    PeSizes s;
    s.N = (int64_t)n_img * (Himg / p) * (Wimg / p);
    s.rows = s.N * p * p;
    s.kp1 = (9 * C + 7) / 8 * 8;
    s.cols1 = al256(s.rows * s.kp1 * 2);
    s.c1 = s.a0 = s.c2 = s.a1 = al256(s.rows * 64 * 2);
    s.y = al256(s.N * 64 * p * p * 2);
    s.stat = al256(s.N * 32 * 4);
    int64_t o = 0;
    s.off_cols1 = o; o += s.cols1;
    s.off_c1 = o; o += s.c1;
    s.off_a0 = o; o += s.a0;
    s.off_c2 = o; o += s.c2;
    s.off_a1 = o; o += s.a1;
    s.off_y = o; o += s.y;
    s.off_m0 = o; o += s.stat;
    s.off_r0 = o; o += s.stat;
    s.off_m1 = o; o += s.stat;
    s.off_r1 = o; o += s.stat;
    s.save_total = o;
    return s;
}
static int pe_check(int n_img, int C, int Himg, int Wimg, int p, int d, const char* who) {
    if (p != 16 || n_img <= 0 || C <= 0 || C > 8 || Himg % p || Wimg % p || d <= 0 || d % 8) DB1_FAIL(DB1_ERR_UNSUPPORTED, "%s: 16 x 16 patches, C <= 8, d %% 8 == 0 (got p=%d C=%d %dx%d d=%d)", who, p, C, Himg, Wimg, d);
    return DB1_OK;
}
extern "C" int64_t db1_patch_embed_save_bytes(int n_img, int C, int Himg, int Wimg, int p) { return pe_sizes(n_img, C, Himg, Wimg, p).save_total; }
extern "C" int64_t db1_patch_embed_workspace_bytes(int n_img, int C, int Himg, int Wimg, int p, int d, int backward) {
    const PeSizes s = pe_sizes(n_img, C, Himg, Wimg, p);
    const int K = 64 * p * p;
    int64_t g = db1_gemm_workspace_bytes((int)s.rows, 64, (int)s.kp1, DB1_BF16, DB1_BF16, DB1_BF16, s.kp1, 1, 1, s.kp1, 64, 1, 1, 1);
    int64_t g2 = db1_gemm_workspace_bytes((int)s.N, d, K, DB1_BF16, DB1_BF16, DB1_BF16, K, 1, 1, K, d, 1, 1, 1);
    g = g > g2 ? g : g2;
    int64_t fixed = al256(64 * s.kp1 * 2) + 2 * al256(64 * 576 * 2) + al256(s.rows * (C > 64 ? C : 64) * 2);   // permuted weights + one activation-sized temporary
    if (backward) {
        g2 = db1_gemm_workspace_bytes(d, K, (int)s.N, DB1_BF16, DB1_BF16, DB1_F32, 1, d, K, 1, K, 1, 1, 1);
        g = g > g2 ? g : g2;
        g2 = db1_gemm_workspace_bytes((int)s.N, K, d, DB1_BF16, DB1_BF16, DB1_BF16, d, 1, K, 1, K, 1, 1, 1);
        g = g > g2 ? g : g2;
        g2 = db1_gemm_workspace_bytes(64, (int)s.kp1, (int)s.rows, DB1_BF16, DB1_BF16, DB1_F32, 1, 64, s.kp1, 1, s.kp1, 1, 1, 1);
        g = g > g2 ? g : g2;
        g2 = db1_colsum_acc_workspace_bytes(s.rows, 64);
        g = g > g2 ? g : g2;
        g2 = db1_colsum_acc_workspace_bytes(s.N, d);
        g = g > g2 ? g : g2;
        g2 = db1_conv3x3_implicit_wgrad_workspace_bytes(s.N);
        g = g > g2 ? g : g2;
        g2 = db1_groupnorm_gelu_nhwc_bwd_workspace_bytes(s.N);
        g = g > g2 ? g : g2;
        fixed += s.y + 3 * al256(s.rows * 64 * 2) + al256(64 * 576 * 4) + al256(64 * s.kp1 * 4) + al256(64 * 576 * 2);   // dy (NCHW), three gradient tiles, permuted weight gradients, W^T operand
    }
    return fixed + al256(g) + 256;
}

extern "C" int db1_patch_embed_fwd(const float* pixels, const void* const* weights, void* emb, void* save, int n_img, int C, int Himg, int Wimg, int p,
                                   int d, void* ws, int64_t ws_bytes, void* stream) {
    CK(pe_check(n_img, C, Himg, Wimg, p, d, "patch_embed_fwd"));
    if (!pixels || !weights || !emb || !save) DB1_FAIL(DB1_ERR_BAD_SHAPE, "patch_embed_fwd: null buffer");
    DB1_NEED_WS(ws, ws_bytes, db1_patch_embed_workspace_bytes(n_img, C, Himg, Wimg, p, d, 0), "patch_embed_fwd");
    const PeSizes s = pe_sizes(n_img, C, Himg, Wimg, p);
    const int hw = p * p, K = 64 * hw;
    char* sv = (char*)save;
    char* w = (char*)ws;
    void* wp1 = w; w += al256(64 * s.kp1 * 2);
    void* wp2 = w; w += al256(64 * 576 * 2);
    void* wp3 = w; w += al256(64 * 576 * 2);
    void* tmp = w; w += al256(s.rows * (C > 64 ? C : 64) * 2);     // normalised patches, then the third convolution's output
    void* gws = w;
    const int64_t gws_b = ws_bytes - (w - (char*)ws);
    void *cols1 = sv + s.off_cols1, *c1 = sv + s.off_c1, *a0 = sv + s.off_a0, *c2 = sv + s.off_c2, *a1 = sv + s.off_a1, *y = sv + s.off_y;
    float *m0 = (float*)(sv + s.off_m0), *r0 = (float*)(sv + s.off_r0), *m1 = (float*)(sv + s.off_m1), *r1 = (float*)(sv + s.off_r1);
    const int bf = DB1_BF16;
    CK(db1_patch_normalize_nhwc(pixels, tmp, n_img, C, Himg, Wimg, p, DB1_F32, bf, stream));
    CK(db1_conv_weight_permute(weights[0], wp1, 64, C, (int)s.kp1, bf, bf, stream));
    CK(db1_im2col3x3_nhwc(tmp, cols1, s.N, C, p, (int)s.kp1, bf, stream));
    CK(db1_gemm_strided(cols1, wp1, c1, weights[1], (int)s.rows, 64, (int)s.kp1, bf, bf, bf, bf, s.kp1, 1, 1, s.kp1, 64, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, 0.f, gws, gws_b, stream));
    CK(db1_groupnorm_gelu_nhwc_fwd(c1, weights[2], weights[3], a0, m0, r0, s.N, 64, hw, 32, 1e-5f, bf, bf, stream));
    CK(db1_conv_weight_permute(weights[4], wp2, 64, 64, 576, bf, bf, stream));
    CK(db1_conv3x3_implicit_fwd(a0, wp2, weights[5], c2, s.N, 1, bf, stream));
    CK(db1_groupnorm_gelu_nhwc_fwd(c2, weights[6], weights[7], a1, m1, r1, s.N, 64, hw, 32, 1e-5f, bf, bf, stream));
    CK(db1_conv_weight_permute(weights[8], wp3, 64, 64, 576, bf, bf, stream));
    CK(db1_conv3x3_implicit_fwd(a1, wp3, weights[9], tmp, s.N, 1, bf, stream));
    CK(db1_add(c1, tmp, tmp, s.rows * 64, bf, stream));                                   // residual
    CK(db1_nhwc_to_nchw(tmp, y, s.N, 64, hw, bf, stream));                                // (c, y, x) flattening = the projection weight's layout
    return db1_gemm_strided(y, weights[10], emb, weights[11], (int)s.N, d, K, bf, bf, bf, bf, K, 1, 1, K, d, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, 0.f, gws, gws_b, stream);
}

extern "C" int db1_patch_embed_bwd(const void* demb, const void* const* weights, const void* save, float* const* grads, int n_img, int C, int Himg,
                                   int Wimg, int p, int d, void* ws, int64_t ws_bytes, void* stream) {
    CK(pe_check(n_img, C, Himg, Wimg, p, d, "patch_embed_bwd"));
    if (!demb || !weights || !save || !grads) DB1_FAIL(DB1_ERR_BAD_SHAPE, "patch_embed_bwd: null buffer");
    DB1_NEED_WS(ws, ws_bytes, db1_patch_embed_workspace_bytes(n_img, C, Himg, Wimg, p, d, 1), "patch_embed_bwd");
    const PeSizes s = pe_sizes(n_img, C, Himg, Wimg, p);
    const int hw = p * p, K = 64 * hw, bf = DB1_BF16;
    const char* sv = (const char*)save;
    char* w = (char*)ws;
    void* wp1 = w; w += al256(64 * s.kp1 * 2);
    void* wp2 = w; w += al256(64 * 576 * 2);
    void* wp3 = w; w += al256(64 * 576 * 2);
    void* t0 = w; w += al256(s.rows * (C > 64 ? C : 64) * 2);
    void* dyn = w; w += s.y;                                       // gradient w.r.t. the projection's input, [N, 64, p, p]
    void* t1 = w; w += al256(s.rows * 64 * 2);
    void* t2 = w; w += al256(s.rows * 64 * 2);
    void* t3 = w; w += al256(s.rows * 64 * 2);
    float* gp = (float*)w; w += al256(64 * 576 * 4);
    float* gp1 = (float*)w; w += al256(64 * s.kp1 * 4);
    void* wt = w; w += al256(64 * 576 * 2);
    void* gws = w;
    const int64_t gws_b = ws_bytes - (w - (char*)ws);
    hipStream_t st = (hipStream_t)stream;
    const void *cols1 = sv + s.off_cols1, *c1 = sv + s.off_c1, *a0 = sv + s.off_a0, *c2 = sv + s.off_c2, *a1 = sv + s.off_a1, *y = sv + s.off_y;
    const float *m0 = (const float*)(sv + s.off_m0), *r0 = (const float*)(sv + s.off_r0), *m1 = (const float*)(sv + s.off_m1), *r1 = (const float*)(sv + s.off_r1);
    // projection: dW += demb^T y, db += colsum(demb), dy = demb W
    CK(db1_gemm_strided(demb, y, grads[10], nullptr, d, K, (int)s.N, bf, bf, DB1_F32, 0, 1, d, K, 1, K, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, 1.f, gws, gws_b, stream));
    CK(db1_colsum_acc(demb, grads[11], s.N, d, d, bf, gws, gws_b, stream));
    CK(db1_gemm_strided(demb, weights[10], dyn, nullptr, (int)s.N, K, d, bf, bf, bf, 0, d, 1, K, 1, K, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, 0.f, gws, gws_b, stream));
    CK(db1_nchw_to_nhwc(dyn, t1, s.N, 64, hw, bf, stream));                                // t1 = dy (channels-last): gradient of the residual sum
    // conv3 (residual_path.5): weight / bias gradients, data gradient
    if (hipMemsetAsync(gp, 0, 64 * 576 * 4, st) != hipSuccess) DB1_FAIL(DB1_ERR_HIP, "patch_embed_bwd: memset");
    CK(db1_conv3x3_implicit_wgrad(t1, a1, gp, nullptr, s.N, gws, gws_b, stream));
    CK(db1_conv_wgrad_unpermute(gp, grads[8], 64, 64, 576, stream));
    CK(db1_colsum_acc(t1, grads[9], s.rows, 64, 64, bf, gws, gws_b, stream));
    CK(db1_conv_weight_permute_t(weights[8], wt, 64, 64, bf, bf, stream));
    CK(db1_conv3x3_implicit_fwd(t1, wt, nullptr, t2, s.N, -1, 0, stream));                  // t2 = da1
    CK(db1_groupnorm_gelu_nhwc_bwd(t2, c2, weights[6], weights[7], m1, r1, t3, nullptr, grads[6], grads[7], s.N, 64, hw, 32, bf, bf, gws, gws_b, stream));   // t3 = dc2
    // conv2 (residual_path.2)
    if (hipMemsetAsync(gp, 0, 64 * 576 * 4, st) != hipSuccess) DB1_FAIL(DB1_ERR_HIP, "patch_embed_bwd: memset");
    CK(db1_conv3x3_implicit_wgrad(t3, a0, gp, nullptr, s.N, gws, gws_b, stream));
    CK(db1_conv_wgrad_unpermute(gp, grads[4], 64, 64, 576, stream));
    CK(db1_colsum_acc(t3, grads[5], s.rows, 64, 64, bf, gws, gws_b, stream));
    CK(db1_conv_weight_permute_t(weights[4], wt, 64, 64, bf, bf, stream));
    CK(db1_conv3x3_implicit_fwd(t3, wt, nullptr, t2, s.N, -1, 0, stream));                  // t2 = da0
    CK(db1_groupnorm_gelu_nhwc_bwd(t2, c1, weights[2], weights[3], m0, r0, t3, t1, grads[2], grads[3], s.N, 64, hw, 32, bf, bf, gws, gws_b, stream));   // t3 = dc1: GroupNorm branch + the residual branch (t1) in one pass
    // conv1: weight / bias gradients only (the pixels need none)
    if (hipMemsetAsync(gp1, 0, 64 * s.kp1 * 4, st) != hipSuccess) DB1_FAIL(DB1_ERR_HIP, "patch_embed_bwd: memset");
    CK(db1_gemm_strided(t3, cols1, gp1, nullptr, 64, (int)s.kp1, (int)s.rows, bf, bf, DB1_F32, 0, 1, 64, s.kp1, 1, s.kp1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, 1.f, gws, gws_b, stream));
    CK(db1_conv_wgrad_unpermute(gp1, grads[0], 64, C, (int)s.kp1, stream));
    (void)wp1; (void)wp2; (void)wp3; (void)t0;
    return db1_colsum_acc(t3, grads[1], s.rows, 64, 64, bf, gws, gws_b, stream);
}
