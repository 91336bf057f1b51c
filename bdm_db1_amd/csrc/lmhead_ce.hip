// Tied LM head + masked cross-entropy without the (tokens x vocabulary) logits tensor (transformer_xl.py:593-613; SURVEY 8b
// db1_lmhead_ce_*).  The token rows are processed in chunks: one chunk of logits (chunk_rows x n_w_rows) lives in the caller's
// workspace, is reduced to (lse, loss) by the CE kernel, and -- in the training entry point -- is turned into dlogits in place and
// consumed by the two gradient GEMMs before the next chunk overwrites it.  At DB1-1.3B / 65 536 tokens that is 1.1 GB of scratch
// instead of a 4.4 GB logits buffer (+ a second one when the caller wants to keep the logits), and no recomputation: the loss
// normaliser sum(mask) is known before the sweep, so forward and backward of the head can share one pass.
// The GEMMs and the CE kernels are the library's own entry points (tile kernels, deterministic reductions); this file only sequences them.
#include "db1_common.h"

static inline int64_t al256(int64_t x) { return (x + 255) & ~(int64_t)255; }

__global__ __launch_bounds__(1024) void lmhead_mask_sum_kernel(const float* __restrict__ mask, int64_t T_, float* out2) {
    __shared__ float red[1024];
    float a = 0.f;
    for (int64_t t = threadIdx.x; t < T_; t += 1024) a += mask[t];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = 0.f; out2[1] = red[0]; }   // {loss accumulator of the sweep, sum(mask)}: fixed order, deterministic
}
__global__ void lmhead_add_loss_kernel(const float* chunk2, float* acc2) { acc2[0] += chunk2[0]; }              // chunk order: deterministic
__global__ void lmhead_finish_kernel(const float* acc2, float* sums) { sums[0] += acc2[0]; sums[1] += acc2[1]; }

struct LmheadPlan { int64_t chunk, logits_b, gemm_b, ce_b, total; };
static LmheadPlan lmhead_plan(int64_t T_, int n_w_rows, int d, int chunk_rows, int dt, bool train) {
    LmheadPlan p;
    p.chunk = chunk_rows > 0 ? chunk_rows : 16384;
    if (p.chunk > T_) p.chunk = T_;
    const int es = dt == DB1_F32 ? 4 : 2;
    p.logits_b = al256(p.chunk * (int64_t)n_w_rows * es);
    int64_t g = db1_gemm_workspace_bytes((int)p.chunk, n_w_rows, d, dt, dt, dt, d, 1, 1, d, n_w_rows, 1, 1, 1);                 // logits = h W^T
    if (train) {
        const int64_t g2 = db1_gemm_workspace_bytes(n_w_rows, d, (int)p.chunk, dt, dt, DB1_F32, 1, n_w_rows, d, 1, d, 1, 1, 1);  // dW = dlogits^T h
        const int64_t g3 = db1_gemm_workspace_bytes((int)p.chunk, d, n_w_rows, dt, dt, dt, n_w_rows, 1, d, 1, d, 1, 1, 1);       // dh = dlogits W
        g = g > g2 ? g : g2;
        g = g > g3 ? g : g3;
    }
    p.gemm_b = al256(g);
    p.ce_b = al256(db1_masked_ce_fwd_workspace_bytes(p.chunk));
    p.total = p.logits_b + p.gemm_b + p.ce_b + 256;
    return p;
}

extern "C" int64_t db1_lmhead_ce_workspace_bytes(int64_t T_, int n_w_rows, int d, int chunk_rows, int dt, int train) {
    if (T_ <= 0 || n_w_rows <= 0 || d <= 0) return 0;
    return lmhead_plan(T_, n_w_rows, d, chunk_rows, dt, train != 0).total;
}

static int lmhead_sweep(const void* h, const void* W, const int64_t* labels, const float* mask, float* lse, float* sums, void* dh, float* dW_acc,
                        float beta_dw, float gscale, int64_t T_, int V, int n_w_rows, int d, int chunk_rows, int dt, void* ws, int64_t ws_bytes,
                        void* stream, bool train, const char* who) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "%s: dtype", who);
    if (T_ <= 0 || V <= 0 || n_w_rows < V || d <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "%s: T=%lld V=%d rows=%d d=%d", who, (long long)T_, V, n_w_rows, d);
    if (!h || !W || !labels || !mask || !lse || !sums || (train && (!dh || !dW_acc))) DB1_FAIL(DB1_ERR_BAD_SHAPE, "%s: null buffer", who);
    const LmheadPlan p = lmhead_plan(T_, n_w_rows, d, chunk_rows, dt, train);
    DB1_NEED_WS(ws, ws_bytes, p.total, who);
    hipStream_t st = (hipStream_t)stream;
    const int es = dt == DB1_F32 ? 4 : 2;
    char* logits = (char*)ws;
    void* gws = logits + p.logits_b;
    void* cws = (char*)gws + p.gemm_b;
    float* acc2 = (float*)((char*)cws + p.ce_b);     // {sum of masked losses so far, sum(mask) over ALL rows (constant during the sweep)}
    float* chunk2 = acc2 + 4;                          // per-chunk {loss, mask} pair of db1_masked_ce_fwd
    lmhead_mask_sum_kernel<<<1, 1024, 0, st>>>(mask, T_, acc2);
    DB1_CHECK_LAUNCH("lmhead mask sum");
    for (int64_t r0 = 0; r0 < T_; r0 += p.chunk) {
        const int rows = (int)((T_ - r0 < p.chunk) ? (T_ - r0) : p.chunk);
        const char* hc = (const char*)h + r0 * d * es;
        int rc = db1_gemm_strided(hc, W, logits, nullptr, rows, n_w_rows, d, dt, dt, dt, 0, d, 1, 1, d, n_w_rows, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f, 0.f,
                                  gws, p.gemm_b, stream);                                                        // logits = h W^T
        if (rc) return rc;
        if (hipMemsetAsync(chunk2, 0, 2 * sizeof(float), st) != hipSuccess) DB1_FAIL(DB1_ERR_HIP, "%s: memset", who);
        if (train && db1_masked_ce_fwd_bwd_supported(V, n_w_rows, dt)) {
            // loss and dlogits (in place; normaliser = sum(mask) over all rows = acc2[1]) from ONE pass over the chunk: the row stays in registers
            rc = db1_masked_ce_fwd_bwd(logits, labels + r0, mask + r0, lse + r0, chunk2, acc2, rows, V, n_w_rows, gscale, dt, cws, p.ce_b, stream);
            if (rc) return rc;
            lmhead_add_loss_kernel<<<1, 1, 0, st>>>(chunk2, acc2);
            DB1_CHECK_LAUNCH("lmhead chunk loss");
        } else {
            rc = db1_masked_ce_fwd(logits, labels + r0, mask + r0, lse + r0, chunk2, rows, V, n_w_rows, dt, cws, p.ce_b, stream);
            if (rc) return rc;
            lmhead_add_loss_kernel<<<1, 1, 0, st>>>(chunk2, acc2);
            DB1_CHECK_LAUNCH("lmhead chunk loss");
            if (!train) continue;
            // dlogits in place (normaliser = sum(mask) over all rows = acc2[1]), then its two products before the next chunk overwrites it
            rc = db1_masked_ce_bwd(logits, labels + r0, mask + r0, lse + r0, acc2, logits, rows, V, n_w_rows, gscale, dt, stream);
            if (rc) return rc;
        }
        rc = db1_gemm_strided(logits, hc, dW_acc, nullptr, n_w_rows, d, rows, dt, dt, DB1_F32, 0, 1, n_w_rows, d, 1, d, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.f,
                              r0 == 0 ? beta_dw : 1.f, gws, p.gemm_b, stream);                                   // dW (+)= dlogits^T h
        if (rc) return rc;
        rc = db1_gemm_strided(logits, W, (char*)dh + r0 * d * es, nullptr, rows, d, n_w_rows, dt, dt, dt, 0, n_w_rows, 1, d, 1, d, 1, 1, 1, 0, 0, 0, 0, 0,
                              0, 1.f, 0.f, gws, p.gemm_b, stream);                                               // dh = dlogits W
        if (rc) return rc;
    }
    lmhead_finish_kernel<<<1, 1, 0, st>>>(acc2, sums);   // what db1_masked_ce_fwd over the whole logits tensor would have added
    DB1_CHECK_LAUNCH("lmhead finish");
    return DB1_OK;
}

/* loss only (evaluation, or a forward whose caller does not want the logits): lse [T], sums[0] += sum(mask * nll), sums[1] += sum(mask) */
extern "C" int db1_lmhead_ce_fwd(const void* h, const void* W, const int64_t* labels, const float* mask, float* lse, float* sums, int64_t T_, int V,
                                 int n_w_rows, int d, int chunk_rows, int dt, void* ws, int64_t ws_bytes, void* stream) {
    return lmhead_sweep(h, W, labels, mask, lse, sums, nullptr, nullptr, 0.f, 1.f, T_, V, n_w_rows, d, chunk_rows, dt, ws, ws_bytes, stream, false,
                        "lmhead_ce_fwd");
}
/* training: the same sweep also produces dh [T, d] = d(loss * gscale) / dh and dW_acc [n_w_rows, d] (float32) = beta_dw * dW_acc + d(loss * gscale) / dW */
extern "C" int db1_lmhead_ce_fwd_bwd(const void* h, const void* W, const int64_t* labels, const float* mask, float* lse, float* sums, void* dh,
                                     float* dW_acc, float beta_dw, float gscale, int64_t T_, int V, int n_w_rows, int d, int chunk_rows, int dt, void* ws,
                                     int64_t ws_bytes, void* stream) {
    return lmhead_sweep(h, W, labels, mask, lse, sums, dh, dW_acc, beta_dw, gscale, T_, V, n_w_rows, d, chunk_rows, dt, ws, ws_bytes, stream, true,
                        "lmhead_ce_fwd_bwd");
}
