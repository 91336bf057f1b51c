// PositionwiseFF with the GEGLU activation inside the GEMMs (transformer_xl.py:246-292, activations.py:19-32): the "bias + GEGLU" epilogue
// of SURVEY 8b's db1_gemm_{nt,nn} contract.
//   forward   z = x W1^T + b1            [M, 2 dff]   and   act = z[:, :dff] * gelu(z[:, dff:])   [M, dff]    from the same accumulators
//   backward  dact = dy W2 (never stored), dz = (dact * gelu(g), dact * v * gelu'(g)) [M, 2 dff], dbias1 += column sums of dz
// Large bf16 shapes run the 4-wave hand-scheduled kernel with these epilogues (gemm_w4.hip); every other shape / dtype runs the same
// arithmetic as separate launches (db1_gemm_* + db1_ffn_act_*), so the entry points work for every model size and the fp32 parity gate.
#include "gemm_tile.h"

int db1_colsum_part_reduce_launch(const float* part, float* out, int nchunks, int cols, hipStream_t st);   // elementwise.hip

static bool geglu_fused_fwd(int M, int dff, int K, int dt, int64_t lda, int64_t ldw, int64_t ldz, int64_t ldact, const void* A, const void* W,
                            const void* Z, const void* ACT) {
    return dt == DB1_BF16 && db1_knob(DB1_KNOB_GEGLU_EPI, 1) != 0 && db1_knob(DB1_KNOB_W4, 1) == 1 && db1_knob(DB1_KNOB_GEMM_TILE, 0) == 0 &&
           (int64_t)(M / 256) * (2 * dff / 256) >= 160 && db1_gemm_w4_geglu_supported(M, dff, K, lda, ldw, ldz, ldact, true) &&
           db1_aligned16(A) && db1_aligned16(W) && db1_aligned16(Z) && db1_aligned16(ACT);
}
static bool geglu_fused_bwd(int M, int dff, int K, int dt, int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz, const void* dY, const void* W,
                            const void* Z, const void* dZ) {
    return dt == DB1_BF16 && db1_knob(DB1_KNOB_GEGLU_EPI, 1) != 0 && db1_knob(DB1_KNOB_W4, 1) == 1 && db1_knob(DB1_KNOB_GEMM_TILE, 0) == 0 &&
           (int64_t)(M / 256) * (dff / 256) >= 160 && db1_gemm_w4_geglu_supported(M, dff, K, lddy, ldw, ldz, lddz, false) &&
           db1_aligned16(dY) && db1_aligned16(W) && db1_aligned16(Z) && db1_aligned16(dZ);
}

extern "C" int db1_gemm_nt_geglu_fused(int M, int dff, int K, int dt, int64_t lda, int64_t ldw, int64_t ldz, int64_t ldact) {
    return geglu_fused_fwd(M, dff, K, dt, lda, ldw, ldz, ldact, nullptr, nullptr, nullptr, nullptr) ? 1 : 0;
}
extern "C" int db1_gemm_nn_geglu_bwd_fused(int M, int dff, int K, int dt, int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz) {
    return geglu_fused_bwd(M, dff, K, dt, lddy, ldw, ldz, lddz, nullptr, nullptr, nullptr, nullptr) ? 1 : 0;
}

extern "C" int db1_gemm_nt_geglu(const void* A, const void* W1, const void* bias, void* Z, void* ACT, int M, int dff, int K, int64_t lda,
                                 int64_t ldw, int64_t ldz, int64_t ldact, int dt, void* ws, int64_t ws_bytes, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "gemm_nt_geglu: dtype");
    if (M <= 0 || dff <= 0 || K <= 0 || !A || !W1 || !Z || !ACT) DB1_FAIL(DB1_ERR_BAD_SHAPE, "gemm_nt_geglu: M=%d dff=%d K=%d / null operand", M, dff, K);
    if (geglu_fused_fwd(M, dff, K, dt, lda, ldw, ldz, ldact, A, W1, Z, ACT)) {
        GemmTileArgs t;
        t.A = (const bf16_t*)A; t.B = (const bf16_t*)W1; t.C = Z; t.bias = bias;
        t.M = M; t.N = 2 * dff; t.K = K; t.lda = lda; t.ldb = ldw; t.ldc = ldz;
        t.batch1 = 1; t.a_bs0 = t.a_bs1 = t.b_bs0 = t.b_bs1 = t.c_bs0 = t.c_bs1 = 0;
        t.alpha = 1.f; t.beta = 0.f; t.tiles_m = M / 256; t.tiles_n = 2 * dff / 256; t.ksplit = 1;
        t.tri_mode = 0; t.tri_period = 0;
        t.split_n = 0; t.Cu = nullptr; t.Cv = nullptr; t.bias_u = nullptr; t.bias_v = nullptr; t.ld_uv = 0;
        t.geglu_dff = dff; t.Cact = ACT; t.ld_act = ldact;
        return db1_gemm_w4_geglu_fwd_launch(t, dt, (hipStream_t)stream);
    }
    if (ldact != dff || ldz != 2 * dff) DB1_FAIL(DB1_ERR_BAD_SHAPE, "gemm_nt_geglu: the unfused path needs contiguous z / act (ldz=%lld ldact=%lld)", (long long)ldz, (long long)ldact);
    int rc = db1_gemm_nt(A, W1, Z, bias, M, 2 * dff, K, lda, ldw, ldz, dt, dt, 1.f, 0.f, ws, ws_bytes, stream);
    if (rc) return rc;
    return db1_ffn_act_fwd(Z, ACT, M, dff, DB1_ACT_GEGLU, dt, stream);
}

// workspace: fused = the per-row-block column sums [M / 128][2 dff] fp32; unfused = dact [M, dff] + what db1_ffn_act_bwd_bias and the GEMM ask for
static int64_t up256(int64_t x) { return (x + 255) / 256 * 256; }
extern "C" int64_t db1_gemm_nn_geglu_bwd_workspace_bytes(int M, int dff, int K, int dt, int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz) {
    if (M <= 0 || dff <= 0 || K <= 0) return 0;
    if (geglu_fused_bwd(M, dff, K, dt, lddy, ldw, ldz, lddz, nullptr, nullptr, nullptr, nullptr)) return (int64_t)(M / 128) * 2 * dff * (int64_t)sizeof(float);
    const int64_t es = db1_elt_size(dt);
    return up256((int64_t)M * dff * es) + up256(db1_ffn_act_bwd_bias_workspace_bytes(M, dff, DB1_ACT_GEGLU)) +
           db1_gemm_workspace_bytes(M, dff, K, dt, dt, dt, lddy, 1, ldw, 1, dff, 1, 1, 1);
}
// the fused backward WITHOUT the bias reduce: the column sums of dz per 128-row block stay in `parts` [M / 128][2 dff] float32 for the caller
// to add up later (db1_colsum_acc): gradient accumulation reduces once per optimizer step.  Fused shapes only (db1_gemm_nn_geglu_bwd_fused).
extern "C" int db1_gemm_nn_geglu_bwd_parts(const void* dY, const void* W2, const void* Z, void* dZ, float* parts, int M, int dff, int K, int64_t lddy,
                                           int64_t ldw, int64_t ldz, int64_t lddz, int dt, void* stream) {
    if (M <= 0 || dff <= 0 || K <= 0 || !dY || !W2 || !Z || !dZ || !parts) DB1_FAIL(DB1_ERR_BAD_SHAPE, "gemm_nn_geglu_bwd_parts: M=%d dff=%d K=%d / null operand", M, dff, K);
    if (!geglu_fused_bwd(M, dff, K, dt, lddy, ldw, ldz, lddz, dY, W2, Z, dZ) || !db1_aligned16(parts))
        DB1_FAIL(DB1_ERR_UNSUPPORTED, "gemm_nn_geglu_bwd_parts: only the shapes of the fused 4-wave kernel (db1_gemm_nn_geglu_bwd_fused)");
    GemmTileArgs t;
    t.A = (const bf16_t*)dY; t.B = (const bf16_t*)W2; t.C = dZ; t.bias = nullptr;
    t.M = M; t.N = dff; t.K = K; t.lda = lddy; t.ldb = ldw; t.ldc = lddz;
    t.batch1 = 1; t.a_bs0 = t.a_bs1 = t.b_bs0 = t.b_bs1 = t.c_bs0 = t.c_bs1 = 0;
    t.alpha = 1.f; t.beta = 0.f; t.tiles_m = M / 256; t.tiles_n = dff / 256; t.ksplit = 1;
    t.tri_mode = 0; t.tri_period = 0;
    t.split_n = 0; t.Cu = nullptr; t.Cv = nullptr; t.bias_u = nullptr; t.bias_v = nullptr; t.ld_uv = 0;
    t.geglu_dff = dff; t.Zin = (const bf16_t*)Z; t.ld_z = ldz; t.colpart = parts;
    return db1_gemm_w4_geglu_bwd_launch(t, (hipStream_t)stream);
}
extern "C" int db1_gemm_nn_geglu_bwd(const void* dY, const void* W2, const void* Z, void* dZ, float* dbias_acc, int M, int dff, int K, int64_t lddy,
                                     int64_t ldw, int64_t ldz, int64_t lddz, int dt, void* ws, int64_t ws_bytes, void* stream) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "gemm_nn_geglu_bwd: dtype");
    if (M <= 0 || dff <= 0 || K <= 0 || !dY || !W2 || !Z || !dZ || !dbias_acc) DB1_FAIL(DB1_ERR_BAD_SHAPE, "gemm_nn_geglu_bwd: M=%d dff=%d K=%d / null operand", M, dff, K);
    DB1_NEED_WS(ws, ws_bytes, db1_gemm_nn_geglu_bwd_workspace_bytes(M, dff, K, dt, lddy, ldw, ldz, lddz), "gemm_nn_geglu_bwd");
    hipStream_t st = (hipStream_t)stream;
    if (geglu_fused_bwd(M, dff, K, dt, lddy, ldw, ldz, lddz, dY, W2, Z, dZ)) {
        GemmTileArgs t;
        t.A = (const bf16_t*)dY; t.B = (const bf16_t*)W2; t.C = dZ; t.bias = nullptr;
        t.M = M; t.N = dff; t.K = K; t.lda = lddy; t.ldb = ldw; t.ldc = lddz;
        t.batch1 = 1; t.a_bs0 = t.a_bs1 = t.b_bs0 = t.b_bs1 = t.c_bs0 = t.c_bs1 = 0;
        t.alpha = 1.f; t.beta = 0.f; t.tiles_m = M / 256; t.tiles_n = dff / 256; t.ksplit = 1;
        t.tri_mode = 0; t.tri_period = 0;
        t.split_n = 0; t.Cu = nullptr; t.Cv = nullptr; t.bias_u = nullptr; t.bias_v = nullptr; t.ld_uv = 0;
        t.geglu_dff = dff; t.Zin = (const bf16_t*)Z; t.ld_z = ldz; t.colpart = (float*)ws;
        int rc = db1_gemm_w4_geglu_bwd_launch(t, st);
        if (rc) return rc;
        return db1_colsum_part_reduce_launch((const float*)ws, dbias_acc, M / 128, 2 * dff, st);
    }
    if (ldz != 2 * dff || lddz != 2 * dff) DB1_FAIL(DB1_ERR_BAD_SHAPE, "gemm_nn_geglu_bwd: the unfused path needs contiguous z / dz");
    const int64_t es = db1_elt_size(dt);
    char* w = (char*)ws;
    void* dact = w;
    w += up256((int64_t)M * dff * es);
    void* ws_act = w;
    const int64_t n_act = db1_ffn_act_bwd_bias_workspace_bytes(M, dff, DB1_ACT_GEGLU);
    w += up256(n_act);
    const int64_t n_gemm = ws_bytes - (w - (char*)ws);
    int rc = db1_gemm_nn(dY, W2, dact, nullptr, M, dff, K, lddy, ldw, dff, dt, dt, 1.f, 0.f, n_gemm > 0 ? w : nullptr, n_gemm > 0 ? n_gemm : 0, stream);
    if (rc) return rc;
    return db1_ffn_act_bwd_bias(Z, dact, dZ, dbias_acc, M, dff, DB1_ACT_GEGLU, dt, ws_act, n_act, stream);
}
