/*
 * db1_hip.h -- C ABI of libdb1_hip.so: the MI355X (gfx950) kernels behind the DB1
 * hot path (Transformer-XL decoder fwd/bwd, image-patch embedder, masked CE,
 * fused Adam, mu-law tokenizer).
 *
 * The reference (Shanghai-Digital-Brain-Laboratory/BDM-DB1) has NO native boundary
 * on this path: it is eager PyTorch (src/model/transformer_xl.py) under DeepSpeed.
 * Each entry point below therefore cites the reference Python lines whose
 * arithmetic it replaces; bdm_db1_amd/ binds them with ctypes (INTEGRATION.md
 * shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm tensors are
 *     only containers).  The library never allocates, frees or synchronises: there is no
 *     hipMalloc / hipFree / hipDeviceSynchronize behind any entry point.  An entry point
 *     that needs scratch takes (void* ws, int64_t ws_bytes) -- 16-byte aligned device memory
 *     the caller owns -- and has a query  int64_t db1_<op>_workspace_bytes(shape...)  that
 *     returns what that call needs (0 = none, ws may then be NULL); too little workspace is
 *     DB1_ERR_WORKSPACE_TOO_SMALL, except for the GEMMs, which then take a kernel that needs
 *     none.  The scratch is dead when the call's work on the stream is done: one buffer of the
 *     largest size serves every call of a stream;
 *   - every call is asynchronous on the given hipStream_t (passed as void*) and may be
 *     captured into a hipGraph;
 *   - return value: 0 = ok, negative = DB1_ERR_*; db1_last_error() gives a
 *     thread-local message; no C++ exception crosses the boundary;
 *   - dtype codes: DB1_F32 / DB1_BF16; "acc" outputs are float32 and are
 *     ACCUMULATED into (+=), so gradient accumulation and tied parameters are free;
 *   - no global mutable state: the library keeps one immutable per-device cache ("this kernel's
 *     dynamic-LDS attribute is set", std::call_once) and reads its A/B environment switches
 *     once; the kernel-steering hooks of the tests are thread-local and live in db1_hip_test.h.
 *     Any number of host threads / devices / streams may call concurrently;
 *   - no float atomics anywhere: every reduction across workgroups goes through partial sums (in the workspace) that are added in a fixed
 *     order, or has a single contributor per output -- the same inputs give the same bits, with or without a workspace.
 */
#ifndef DB1_HIP_H
#define DB1_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DB1_F32 = 0, DB1_BF16 = 1 };
enum {
    DB1_OK = 0,
    DB1_ERR_BAD_SHAPE = -1,
    DB1_ERR_BAD_ALIGN = -2,
    DB1_ERR_UNSUPPORTED_DTYPE = -3,
    DB1_ERR_WORKSPACE_TOO_SMALL = -4,
    DB1_ERR_HIP = -5,
    DB1_ERR_UNSUPPORTED = -6
};
enum { DB1_ACT_GEGLU = 0, DB1_ACT_GELU = 1, DB1_ACT_RELU = 2 };

int db1_version(void);
const char* db1_last_error(void);
/* 1 if the running device is gfx950; product code refuses to run otherwise. */
int db1_device_is_gfx950(void);

/* ------------------------------------------------------------------ GEMM
 * C[z][m,n] = alpha * sum_k A[z][m,k] * B[z][k,n] + beta * C[z][m,n] + bias[n]
 * Fully strided (element (m,k) of A at A + m*a_rs + k*a_cs, etc.), two-level batch
 * z = z0*batch1 + z1 with per-level strides.  fp32 accumulation always.  Replaces
 * nn.Linear / torch.einsum (transformer_xl.py:136-141,163-170,220,228,264-268,595).
 * Dispatch: bf16 operands whose strides form a K-major/M-major pattern with the
 * alignment the MFMA tile kernels need go to the 128x128x64 bf16 MFMA kernels
 * (NT / NN / TN); everything else (fp32 parity gate, odd shapes, tiny models) goes to
 * a strided kernel on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
 */
int db1_gemm_strided(const void* A, const void* B, void* C, const void* bias,
                     int M, int N, int K, int dtA, int dtB, int dtC, int dtBias,
                     int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int64_t c_rs, int64_t c_cs,
                     int batch0, int batch1,
                     int64_t a_bs0, int64_t a_bs1, int64_t b_bs0, int64_t b_bs1, int64_t c_bs0, int64_t c_bs1,
                     float alpha, float beta, void* ws, int64_t ws_bytes, void* stream);
/* Scratch of the deterministic split-K path (fp32 partial sums of outputs too small to fill the chip, added in a fixed order).
 * Without it (ws NULL / smaller) the same product runs on a kernel that needs none: results are equally valid, large-K weight
 * gradients are slower. */
int64_t db1_gemm_workspace_bytes(int M, int N, int K, int dtA, int dtB, int dtC,
                                 int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int64_t c_rs, int64_t c_cs,
                                 int batch0, int batch1);
/* which kernel db1_gemm_strided takes for this product given ws_bytes of workspace (-1: as much as it wants):
 * 0 strided fp32-MFMA, 1 128x128 tile, 2 256x128 tile, 3 / 4 256x256 8-wave (k64 x 2 / k32 x 4), 5 256x256 4-wave hand-scheduled,
 * 6 skinny W-streaming, 7 256x128 4-wave hand-scheduled; +16 split-K through the workspace; +32 the last partial wave of tile rows runs as a second call */
int db1_gemm_kernel_choice(int M, int N, int K, int dtA, int dtB, int dtC,
                           int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int64_t c_rs, int64_t c_cs,
                           int batch0, int batch1, float beta, int64_t ws_bytes);
/* Row-major 2-D conveniences.  nt: C = A[M,K] * B[N,K]^T (y = x W^T);
 * nn: C = A[M,K] * B[K,N] (dx = dy W);  tn: C = A[K,M]^T * B[K,N] (dW = dy^T x). */
/* db1_gemm_strided with a structural-zero hint for A (an optimisation only: results are those of the plain call):
 *   tri_mode 1: A[m, k] == 0 for k > m;   tri_mode 2: A[m, k] == 0 for (k mod tri_period) < m.
 * Used for the two contractions over dT (dS re-indexed by distance, zero above the causal diagonal): dq_r = dT.R (mode 1) and
 * dR = dT^T.(q+v) over k = (batch, query) (mode 2, period = L): the k-tiles that are zero by construction are never read. */
int db1_gemm_strided_tri(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int dtA, int dtB, int dtC, int dtBias,
                         int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int64_t c_rs, int64_t c_cs, int batch0, int batch1,
                         int64_t a_bs0, int64_t a_bs1, int64_t b_bs0, int64_t b_bs1, int64_t c_bs0, int64_t c_bs1, float alpha, float beta,
                         int tri_mode, int tri_period, void* ws, int64_t ws_bytes, void* stream);
int db1_gemm_nt(const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                int64_t lda, int64_t ldb, int64_t ldc, int dtAB, int dtC, float alpha, float beta, void* ws, int64_t ws_bytes, void* stream);
/* The attention input projection (transformer_xl.py:136-141,160-175): C[M,N] = A[M,K] * W[N,K]^T in bf16, except that the columns n < split_n
 * (the query block) are written as acc + bias_u[n] to Cu and acc + bias_v[n] to Cv (row stride ld_uv) instead of to C: q + r_w_bias and
 * q + r_r_bias leave the GEMM's accumulators directly (no separate pass, one rounding).  Large shapes only (see _supported). */
int db1_gemm_nt_headbias_supported(int M, int N, int K, int split_n);
int db1_gemm_nt_headbias(const void* A, const void* W, void* C, void* Cu, void* Cv, const void* bias_u, const void* bias_v, int M, int N, int K,
                         int split_n, int64_t lda, int64_t ldw, int64_t ldc, int64_t ld_uv, void* stream);
int db1_gemm_nn(const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                int64_t lda, int64_t ldb, int64_t ldc, int dtAB, int dtC, float alpha, float beta, void* ws, int64_t ws_bytes, void* stream);
int db1_gemm_tn(const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                int64_t lda, int64_t ldb, int64_t ldc, int dtAB, int dtC, float alpha, float beta, void* ws, int64_t ws_bytes, void* stream);
/* The "bias + GEGLU" epilogue of the feed-forward GEMMs (PositionwiseFF, transformer_xl.py:246-292; GEGLU = a * gelu_erf(b),
 * activations.py:19-32).  Forward: Z[M, 2 dff] = A[M, K] * W1[2 dff, K]^T + bias AND ACT[M, dff] = Z[:, :dff] * gelu(Z[:, dff:]) from the
 * same accumulators (ACT is computed from the ROUNDED Z, i.e. it equals db1_ffn_act_fwd on the stored Z bit for bit).  Backward:
 * dact = dY[M, K] * W2[K, dff] is never stored; the epilogue reads Z and writes dZ[M, 2 dff] = (dact * gelu(g), dact * v * gelu'(g)) and
 * dbias_acc[2 dff] (fp32) += the column sums of the stored dZ, added in a fixed order (the same arithmetic per element as db1_gemm_nn +
 * db1_ffn_act_bwd_bias).  Large bf16 shapes run fused inside the 4-wave tile GEMM (..._fused() says which: 1 = fused, the separate
 * activation passes over Z disappear); every other shape / dtype runs the equivalent separate launches.  dt: operands, Z, ACT, bias. */
int db1_gemm_nt_geglu_fused(int M, int dff, int K, int dt, int64_t lda, int64_t ldw, int64_t ldz, int64_t ldact);
int db1_gemm_nt_geglu(const void* A, const void* W1, const void* bias, void* Z, void* ACT, int M, int dff, int K,
                      int64_t lda, int64_t ldw, int64_t ldz, int64_t ldact, int dt, void* ws, int64_t ws_bytes, void* stream);
int db1_gemm_nn_geglu_bwd_fused(int M, int dff, int K, int dt, int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz);
int64_t db1_gemm_nn_geglu_bwd_workspace_bytes(int M, int dff, int K, int dt, int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz);
int db1_gemm_nn_geglu_bwd(const void* dY, const void* W2, const void* Z, void* dZ, float* dbias_acc, int M, int dff, int K,
                          int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz, int dt, void* ws, int64_t ws_bytes, void* stream);
/* ... WITHOUT the bias reduce (fused shapes only): parts [M / 128][2 dff] float32 = the column sums of dZ per 128-row block */
int db1_gemm_nn_geglu_bwd_parts(const void* dY, const void* W2, const void* Z, void* dZ, float* parts, int M, int dff, int K,
                                int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz, int dt, void* stream);
/* which kernel db1_gemm_strided would pick: 0 = strided fp32-MFMA, 1 = bf16 MFMA tile kernel */
int db1_gemm_would_use_fast(int M, int N, int K, int dtA, int dtB, int dtC,
                            int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int64_t c_rs, int64_t c_cs);

/* ------------------------------------------------------------------ residual + LayerNorm
 * s = alpha * x + r ; y = LayerNorm(s) * gamma + beta   (transformer_xl.py:231-238, 288-290;
 * alpha = DeepNorm alpha or 1).  r may be NULL (plain LN, pre-LN models).  s_out (may alias r,
 * may be NULL) receives s for the backward.  mean/rstd: float32 [rows]. */
/* Dropout (transformer_xl.py:229,262-269: on the attention / feed-forward output before the residual sum) is part of this kernel:
 * with drop_p > 0, s = alpha * x + dropout(r).  No mask is stored: keep decisions are a counter-based function (Philox4x32-10) of
 * (drop_seed, drop_step, drop_site, element index) -- see db1_dropout -- and the backward regenerates them.
 * drop_step_dev (nullable, device): the step used is drop_step + *drop_step_dev, read when the kernel RUNS -- a captured hipGraph of a
 * training micro-step (bdm_db1_amd.GraphedTrainStep) draws new masks at every replay by bumping that counter. */
int db1_layernorm_residual_fwd(const void* x, const void* r, float alpha, const void* gamma, const void* beta,
                               void* y, void* s_out, float* mean, float* rstd,
                               int64_t rows, int d, float eps,
                               float drop_p, uint64_t drop_seed, uint32_t drop_site, uint32_t drop_step, const uint32_t* drop_step_dev,
                               int dt, int dtParam, void* stream);
/* ds = dL/ds (dtype dt); dr_out (nullable) = dL/dr = ds under the forward's keep decisions (== ds when drop_p = 0);
 * dgamma_acc / dbeta_acc: float32 [d], accumulated.
 * drop_rows_per_step (0: off): the rows are SEVERAL micro-steps' rows, drop_rows_per_step each, run through one backward (a whole gradient-
 * accumulation window, TransformerXL's deferred backward): row r takes the keep decisions of step drop_step + r / drop_rows_per_step, its
 * elements counted from the first row of its block -- what that micro-step's own forward drew. */
int64_t db1_layernorm_residual_bwd_workspace_bytes(int64_t rows, int d, int dt);   /* per-block parameter-gradient partials */
int db1_layernorm_residual_bwd(const void* dy, const void* s, const void* gamma, const float* mean, const float* rstd,
                               void* ds, void* dr_out, float* dgamma_acc, float* dbeta_acc,
                               int64_t rows, int d,
                               float drop_p, uint64_t drop_seed, uint32_t drop_site, uint32_t drop_step, const uint32_t* drop_step_dev,
                               int64_t drop_rows_per_step, int dt, int dtParam, void* ws, int64_t ws_bytes, void* stream);
/* the same launch WITHOUT the parameter reduce: the per-block partial sums [blocks][2][d] float32 (parts_bytes >=
 * db1_layernorm_residual_bwd_workspace_bytes; bf16 rows of the register-resident widths only, else DB1_ERR_UNSUPPORTED) stay in `parts`;
 * db1_colsum_acc over that [blocks, 2 d] matrix yields (dgamma | dbeta).  For gradient accumulation: one reduce per optimizer step. */
int db1_layernorm_residual_bwd_parts(const void* dy, const void* s, const void* gamma, const float* mean, const float* rstd,
                                     void* ds, void* dr_out, float* parts, int64_t parts_bytes, int64_t rows, int d,
                                     float drop_p, uint64_t drop_seed, uint32_t drop_site, uint32_t drop_step, const uint32_t* drop_step_dev,
                                     int64_t drop_rows_per_step, int dt, int dtParam, void* stream);

/* ------------------------------------------------------------------ dropout
 * y[e] = x[e] * keep(e) * 65536 / (65536 - thr),  thr = round(p * 65536)   (nn.Dropout, transformer_xl.py:409,545,575; y may alias x).
 * keep(e) = [ u16(e) >= thr ] where the eight 16-bit uniforms of elements 8b .. 8b+7 are the 128 output bits of Philox4x32-10 on
 * counter (b lo, b hi, site, step) under key (seed lo, seed hi), low half-word first.  The same call is its own backward
 * (dx = dropout(dy) with the same seed / site / step).  n: a multiple of 8. */
int db1_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, uint32_t site, uint32_t step, const uint32_t* step_dev, int dt, void* stream);

/* ------------------------------------------------------------------ feed-forward activation
 * GEGLU: out[r, j] = z[r, j] * gelu_erf(z[r, n + j]) (activations.py:19-32); GELU/RELU: elementwise. */
int db1_ffn_act_fwd(const void* z, void* out, int64_t rows, int n_out, int act, int dt, void* stream);
int db1_ffn_act_bwd(const void* z, const void* dout, void* dz, int64_t rows, int n_out, int act, int dt, void* stream);
/* same, and dbias_acc[c] += sum_r dz[r, c] over all columns of dz (float32 [2*n_out] for GEGLU, [n_out] otherwise): the
 * gradient of the first feed-forward bias (transformer_xl.py:264) from the same pass; fixed summation order. */
int64_t db1_ffn_act_bwd_bias_workspace_bytes(int64_t rows, int n_out, int act);
int db1_ffn_act_bwd_bias(const void* z, const void* dout, void* dz, float* dbias_acc, int64_t rows, int n_out, int act, int dt,
                         void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ inference with memory: the linear maps between two attention launches
 * ONE new token (batch 1) through o_net (fed by the merge of the decode attention's chunk partials, as db1_linear_decode_attn) -> LayerNorm
 * -> ff1 + GEGLU -> ff2 -> LayerNorm -> the NEXT layer's qkv projection, as one launch of 256 persistent workgroups whose weight stream never
 * stops (csrc/decode_chain.hip; transformer_xl.py:227-243,246-292,136; evaluate_rl.py:157-266).  w_qkv_next NULL = the last layer: the launch
 * ends with ff2 and hands back h1_out (LN1's row) and f_out (ff2's row) for the head's own input LayerNorm.  All rows bf16.
 * The workgroups hand the stage vectors over through scratch rows of tagged 32-bit words ({bf16, tag = slot + 1}; polled until every word
 * carries the launch's tag): scratch = db1_decode_chain_scratch_bytes() bytes, ZEROED ONCE when allocated and then left alone; consecutive
 * launches on one scratch must use DIFFERENT slots (0 .. 65534; the layer index, so n_layer >= 2) and run one after the other (one stream).
 * w_o_next (or NULL): the w_o of the launch that follows; its lines are touched so that they wait in L2 / the memory-side cache.
 * The int at db1_decode_chain_error_offset() is set to 1 when a poll ran into its limit (all 256 workgroups must be resident at once;
 * results invalid).  Built for d = 2048, dff = 4096, d_head = 128 on a device with >= 256 CUs (db1_decode_chain_supported checks both). */
int db1_decode_chain_supported(int d, int dff, int H, int D, int nunit);
int64_t db1_decode_chain_scratch_bytes(void);
int64_t db1_decode_chain_error_offset(void);
int db1_decode_chain(const float* att_part, int nunit, int H, const void* x_res, const void* w_o, const void* w1, const void* b1, const void* w2,
                     const void* b2, const void* w_qkv_next, const void* w_o_next, const void* g1, const void* be1, const void* g2, const void* be2,
                     float alpha, float eps, void* h1_out, void* f_out, void* x_next, void* qkv_next, void* scratch, int slot, int d, int dff, void* stream);

/* out_acc[c] += sum_r x[r, c]  (bias / u / v gradients). ldx = row stride in elements. */
int64_t db1_colsum_acc_workspace_bytes(int64_t rows, int cols);   /* per-chunk partials, added in a fixed order */
int db1_colsum_acc(const void* x, float* out_acc, int64_t rows, int cols, int64_t ldx, int dt, void* ws, int64_t ws_bytes, void* stream);
/* y = a + b (elementwise), used by pre-LN residuals and dq = dq_k + dq_r */
int db1_add(const void* a, const void* b, void* y, int64_t n, int dt, void* stream);
/* y = a + b and, from the same pass, sum_a_acc[c] += sum_r a[r, c], sum_b_acc[c] += sum_r b[r, c] (float32, fixed order; y may alias b).
 * The attention backward's dq = dq_k + dq_r with the r_w_bias / r_r_bias gradients (transformer_xl.py:160-209): one pass instead of three. */
int64_t db1_add2d_colsums_workspace_bytes(int64_t rows, int cols);
int db1_add2d_colsums(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, float* sum_a_acc, float* sum_b_acc,
                      int64_t rows, int cols, int dt, void* ws, int64_t ws_bytes, void* stream);
/* y[r, c] = a[r, c] + b[r, c] with row strides (a may have a different dtype; y may alias b) */
int db1_add2d(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int64_t rows, int cols,
              int dtA, int dt, void* stream);
/* base[segments[2i] .. + segments[2i+1]) = 0 for i < n_segments (offsets / lengths in elements, a device table of int64 pairs): one
 * launch clears the scattered small float32 accumulators of the gradient arena after an optimizer step. */
int db1_zero_segments(float* base, const int64_t* segments, int n_segments, void* stream);
/* y[i] = (dtOut) x[i] */
int db1_cast(const void* x, void* y, int64_t n, int dtIn, int dtOut, void* stream);

/* ------------------------------------------------------------------ embeddings
 * out[t, :] = table[ids[t], :] (zeros where ids[t] < 0)   (transformer_xl.py:627-629, 665, 677, 686) */
int db1_embed_gather_fwd(const void* table, const int64_t* ids, void* out, int64_t n_tokens, int d,
                         int64_t ld_out, int64_t n_table_rows, int dtTable, int dtOut, void* stream);
/* out[t, :] += row_table[row_ids[t], :] + col_table[col_ids[t], :]  (VisionEmbedding's position term, vision_embedding.py:117-180, added
 * onto the patch embeddings in one pass; (out + row) + col in fp32, one rounding; ids outside the tables add nothing) */
int db1_vision_pos_add(void* out, const void* row_table, const void* col_table, const int64_t* row_ids, const int64_t* col_ids, int64_t n,
                       int d, int64_t n_table_rows, int dtTable, int dt, void* stream);
/* dtable_acc[ids[t], :] += dout[t, :]; ids outside [0, n_table_rows) are skipped, as they read zeros in the forward.  Deterministic: no
 * float atomics -- the tokens are ordered by table row with a stable device radix sort (csrc/scatter.hip: the library's own LSD passes,
 * no vendor primitive) and every table row is written by the one wave that adds its tokens in token order.  d and ld_dout: multiples of
 * 16 bytes. */
int64_t db1_embed_scatter_add_workspace_bytes(int64_t n_tokens);   /* sort keys / values + the sort's own scratch + partial rows of long runs (d <= 8192) */
int db1_embed_scatter_add_bwd(const void* dout, const int64_t* ids, float* dtable_acc, int64_t n_tokens, int d,
                              int64_t ld_dout, int64_t n_table_rows, int dt, void* ws, int64_t ws_bytes, void* stream);
/* RL sequence assembly (transformer_xl.py:621-660): rows with ids >= 0 take word_table[ids];
 * the k-th "-1" placeholder of row b takes vis[b, k, :]; then + pos_table[position_id].
 * labels (may be NULL): label == -1 -> 0 in place (:644-645).  Token ids >= n_word_rows / position ids outside [0, n_pos_rows)
 * contribute nothing (no out-of-table access). */
int db1_rl_assemble_fwd(const void* word_table, const void* pos_table, const void* vis, const int64_t* ids,
                        const int64_t* position_id, int64_t* labels, void* out,
                        int B, int L, int d, int n_vis_per_row, int64_t n_word_rows, int64_t n_pos_rows, int dtTable, int dt, void* stream);
int64_t db1_rl_assemble_bwd_workspace_bytes(int B, int L);   /* the two table gradients are sorted-run scatters (db1_embed_scatter_add_bwd) */
int db1_rl_assemble_bwd(const void* dout, const int64_t* ids, const int64_t* position_id,
                        float* dword_acc, float* dpos_acc, void* dvis,
                        int B, int L, int d, int n_vis_per_row, int64_t n_word_rows, int64_t n_pos_rows, int dt,
                        void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ masked cross-entropy on materialised logits
 * (transformer_xl.py:602-609). logits [T, ld] (columns >= V are padding); a label outside [0, V) adds no loss and gets no gradient
 * (torch's ignore_index).  fwd: lse[t], and
 * sums[0] += sum_t mask*nll, sums[1] += sum_t mask.  bwd: dlogits = mask/sums[1] * gscale * (softmax - onehot)
 * (zeros in the padding), may be in place. */
int64_t db1_masked_ce_fwd_workspace_bytes(int64_t T);   /* per-token losses, summed in a fixed order */
int db1_masked_ce_fwd(const void* logits, const int64_t* labels, const float* mask, float* lse, float* sums,
                      int64_t T, int V, int64_t ld, int dt, void* ws, int64_t ws_bytes, void* stream);
int db1_masked_ce_bwd(const void* logits, const int64_t* labels, const float* mask, const float* lse, const float* sums,
                      void* dlogits, int64_t T, int V, int64_t ld, float gscale, int dt, void* stream);
/* both in ONE pass over the rows (each row stays in the registers of its workgroup: ld <= 34 816 elements in bf16, 17 408 in fp32): lse,
 * sums[0] += sum(mask * nll), sums[1] += sum(mask), and logits overwritten by dlogits = mask / norm[1] * gscale * (softmax - onehot);
 * norm[1] = the loss normaliser (a device float pair like `sums`).  Bit-equal to db1_masked_ce_fwd followed by db1_masked_ce_bwd.
 * ws: db1_masked_ce_fwd_workspace_bytes(T). */
int db1_masked_ce_fwd_bwd_supported(int V, int64_t ld, int dt);
int db1_masked_ce_fwd_bwd(void* logits, const int64_t* labels, const float* mask, float* lse, float* sums, const float* norm,
                          int64_t T, int V, int64_t ld, float gscale, int dt, void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ tied LM head + masked cross-entropy, chunked over the token rows
 * (transformer_xl.py:593-613).  The (T x vocabulary) logits tensor is never materialised: chunk_rows rows of logits at a time
 * live in the workspace (chunk_rows <= 0: 16 384).  h [T, d], W [n_w_rows, d] (rows >= V are zero padding, n_w_rows a multiple of
 * 256 keeps the tile GEMMs on their fast path), both of dtype dt.  Same loss / lse / gradient arithmetic as db1_gemm_* +
 * db1_masked_ce_* on a materialised tensor (the reductions inside a chunk and across chunks run in a fixed order).
 *   _fwd     : lse [T]; sums[0] += sum(mask * nll), sums[1] += sum(mask)                     (evaluation, logits-free forward)
 *   _fwd_bwd : additionally dh [T, d] (dtype dt) and dW_acc [n_w_rows, d] float32 = beta_dw * dW_acc + dW, both for
 *              loss * gscale -- one sweep, no recomputation: the normaliser sum(mask) is known before the first chunk. */
int64_t db1_lmhead_ce_workspace_bytes(int64_t T, int n_w_rows, int d, int chunk_rows, int dt, int train);
int db1_lmhead_ce_fwd(const void* h, const void* W, const int64_t* labels, const float* mask, float* lse, float* sums,
                      int64_t T, int V, int n_w_rows, int d, int chunk_rows, int dt, void* ws, int64_t ws_bytes, void* stream);
int db1_lmhead_ce_fwd_bwd(const void* h, const void* W, const int64_t* labels, const float* mask, float* lse, float* sums,
                          void* dh, float* dW_acc, float beta_dw, float gscale,
                          int64_t T, int V, int n_w_rows, int d, int chunk_rows, int dt, void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ relative-position attention for inference with memory
 * (evaluate_rl.py:157-266; transformer_xl.py:124-133,160-225).  q = 1..64 new queries per sequence against klen = mlen + q cached
 * keys / values (bf16, d_head 128).  qu, qv: [B, q, H, D]; k, v: row j of sequence b at k + b*kv_batch_stride + j*kv_row_stride
 * (+ h*D); R: [nd, H*D] rows indexed by distance mlen + i - j; out: [B, q, H, D].  Same visibility predicate as the
 * materialised path: i - shift < j <= i + mlen. */
int db1_relattn_decode_supported(int B, int q, int klen, int H, int D, int dt);
int64_t db1_relattn_decode_workspace_bytes(int B, int q, int klen, int H);   /* per-key-chunk partial outputs of the flash-decoding merge */
int db1_relattn_decode_fwd(const void* qu, const void* qv, const void* k, const void* v, int64_t kv_row_stride,
                           int64_t kv_batch_stride, const void* R, int nd, void* out, int B, int q, int klen, int mlen, int H,
                           int D, int shift, float scale, void* ws, int64_t ws_bytes, void* stream);

/* The same attention over a RING of cached keys / values (hipGraph-friendly inference: nothing is concatenated or copied per call):
 * kv_ring [B, cap, 2, H, D] bf16, cap >= mlen + q; logical key j < mlen is ring row (ring_state[0] + j) % cap; the q new keys / values come
 * from this call's packed projections qkv_new [B, q, 3, H, D] and are appended at rows (ring_state[0] + mlen + i) % cap by the same
 * launch; q + r_w_bias / q + r_r_bias are formed inside (u, vb [H, D]).  ring_state is a DEVICE int (the row of logical key 0), advanced
 * by db1_ring_advance(state, q, cap) after the last layer of a call -- so one captured graph serves every call.  klen <= 2048. */
int64_t db1_relattn_decode_ring_workspace_bytes(int B, int q, int klen, int H);
int db1_relattn_decode_ring_fwd(const void* qkv_new, const void* u, const void* vb, void* kv_ring, const int* ring_state, int cap, const void* R,
                                int nd, void* out, int B, int q, int mlen, int H, int D, int shift, float scale, void* ws, int64_t ws_bytes,
                                void* tickets, void* stream);
int db1_ring_advance(int* state, int q, int cap, void* stream);
/* `tickets` (optional): db1_linear_decode_tickets_bytes() bytes, ZERO before the first launch and left zero by every launch that uses them.
 * With it (and q <= 16) the key chunk that finishes last for a (batch, head) merges the partial results inside the attention launch.
 * out == NULL (q <= 16): no merge at all, the per-chunk partial results stay in ws for db1_linear_decode_attn. */

/* ------------------------------------------------------------------ the inference layer's linear maps (M <= 64 new tokens), fused
 * y[M, N] = x[M, K] . W^T + bias (bf16, W K-major like nn.Linear.weight, leading dimension K) as a stream over W, finishing the layer's
 * small follow-up work inside the same launch:
 *   geglu != 0: W is [2 N, K] (value rows, then gate rows), bias [2 N]; y = (x W_v^T + b_v) * gelu_erf(x W_g^T + b_g), both halves rounded
 *               to bf16 first like the reference's z tensor (transformer_xl.py:262-269 with activations.py:19-32);
 *   pre_out != NULL (M <= 16, K <= 2048, K % 256 == 0): the INPUT rows are normalised on the way in, x_eff = LayerNorm(pre_alpha * pre_res + x)
 *               * pre_gamma + pre_beta (two passes in fp32 over the sum rounded to bf16, like db1_layernorm_residual_fwd; the row sums are
 *               grouped differently, so the bf16 result can differ from that launch in the last place), stored to pre_out [M, K] as well --
 *               the residual LayerNorm that closes the PREVIOUS sub-layer costs no launch and no inter-workgroup hand-off;
 *   ln_out != NULL: after the last column group has stored y, ln_out[M, N] = LayerNorm(alpha * res + y) * gamma + beta (N = 512, 1024 or
 *               2048; the arithmetic of db1_layernorm_residual_fwd on the stored bf16 y: equal results) -- for M > 16.
 * K is split over workgroups when N alone gives too few of them; the partial tiles are added in split order by the last arrival
 * (deterministic).  `tickets`: db1_linear_decode_tickets_bytes() bytes, zero before the first launch, left zero.
 * db1_linear_decode_supported(..., ln): bit 0 = with ln_out, bit 1 = with pre_out. */
int64_t db1_linear_decode_tickets_bytes(void);
/* ... and the attention output projection fed by the attention's per-chunk partial results: db1_relattn_decode_ring_fwd with out == NULL
 * (q <= 16) leaves [B*H][chunks = ceil(klen / 128)][64][D + 2] floats (unnormalised output, maximum, sum per query) in ITS workspace; this
 * launch merges them on the way in (the arithmetic of the attention's own merge) and multiplies by W [N, H D].  B q <= 2 rows, <= 12 chunks. */
int db1_linear_decode_attn_supported(int B, int q, int H, int D, int nunit, int N);
int db1_linear_decode_attn(const float* attn_part, int nunit, int B, int q, int H, int D, const void* w, void* y, int64_t ldy, int N, void* stream);
int64_t db1_linear_decode_workspace_bytes(int M, int N, int K, int geglu);
int db1_linear_decode_supported(int M, int N, int K, int geglu, int ln);
int db1_linear_decode(const void* x, int64_t ldx, const void* w, const void* bias, int dtBias, void* y, int64_t ldy, int M, int N, int K,
                      int geglu, const void* pre_res, int64_t ld_pre_res, float pre_alpha, const void* pre_gamma, const void* pre_beta,
                      float pre_eps, void* pre_out, int64_t ld_pre_out, const void* res, int64_t ld_res, float alpha, const void* gamma,
                      const void* beta, float eps, void* ln_out, int64_t ld_out, int dtParam, void* tickets, void* ws, int64_t ws_bytes,
                      void* stream);

/* ------------------------------------------------------------------ relative-position attention, materialised path
 * (fp32 parity gate, any head size).  Buffers S,T are float32 in [H][B][Lq][*] layout.
 * qu = q + u, qv = q + v_bias from the packed qkv activations [B, L, 3, H, D]  (transformer_xl.py:161,167). */
int db1_relattn_add_head_bias(const void* qkv, const void* u, const void* vb, void* qu, void* qv,
                              int B, int Lq, int Lk, int H, int D, int dt, int dtParam, void* stream);
/* P[h,b,i,j] = softmax_j( (AC[h,b,i,j] + T[h,b,i, mlen+i-j]) * scale ) over visible keys
 * i - shift < j <= i + mlen  (closed form of _rel_shift + mask, transformer_xl.py:98-110,171-209,551-567);
 * written in place over AC.  lse (nullable) [H,B,Lq].  A row with NO visible key gets the uniform distribution over all Lk keys,
 * as the reference's masked_fill(-1e30) + softmax does (and no gradient flows into its scores). */
int db1_relattn_softmax_fwd(float* AC, const float* T, float* lse, int H, int B, int Lq, int Lk, int nd,
                            int mlen, int shift, float scale, void* stream);
/* dS = P * (dP - rowsum(P*dP)) * scale, in place over dP; dT[h,b,i,mlen+i-j] = dS[h,b,i,j] (zero elsewhere). */
int db1_relattn_softmax_bwd(const float* P, float* dP, float* dT, int H, int B, int Lq, int Lk, int nd,
                            int mlen, int shift, float scale, void* stream);

/* ------------------------------------------------------------------ relative-position flash attention (bf16, D = 128)
 * Fused QK^T + skewed Q~R^T + online softmax + PV; never materialises (L x L).
 * R [L,H,D] bf16 indexed by distance; out [B,L,H,D] bf16, lse [B,H,L] f32.
 * Training shape only: Lq == Lk, causal window i - shift < j <= i (shift >= L means plain causal). */
int db1_relattn_flash_supported(int B, int L, int H, int D, int dt);
/* qu = q + u, qv = q + v_bias: [B,L,H,D] contiguous (db1_relattn_add_head_bias); k, v: pointers INTO the packed
 * qkv activations with their row / batch strides in elements.
 * probs / mblk (optional, both or neither): keep the unnormalised probabilities p~ = exp2((s - m_blk) c2), c2 = scale log2 e, as bf16 MFMA
 * fragment images [64 lanes][8] per (key block jb of 32, 16-query tile qt) and the running maxima m_blk c2 they refer to [B*H][L/32][L] f32,
 * for db1_relattn_flash_bwd.  Only tiles on or below the causal diagonal exist (qt >= 2 jb); per (batch, head) they are stored as a triangle,
 * key block major: tile index qt + jb (L/16 - 1 - jb) -- db1_relattn_flash_probs_bytes(B, L, H) bytes in all (round 6: 25 instead of 49 GiB
 * for DB1-1.3B at 64 x 1024 tokens).  Tiles of blocks outside a sliding window are not written. */
int64_t db1_relattn_flash_probs_bytes(int B, int L, int H);
int db1_relattn_flash_fwd(const void* qu, const void* qv, const void* k, const void* v, int64_t kv_row_stride,
                          int64_t kv_batch_stride, const void* R, void* out, float* lse,
                          int B, int L, int H, int D, int shift, float scale, void* probs, float* mblk, void* stream);
/* dq (the (q+u).k branch only), dk, dv are written with their own row / batch strides (they live inside dqkv);
 * dT [H,B,L,L] bf16 = dS re-indexed by distance (input of the dq_r / dR GEMMs, zero where nothing is visible);
 * delta [B,H,L] f32 scratch (rowsum(dout * out): written by the query-side kernel, read by the key-side kernel).
 * Three ways to run it, by what the caller provides (all valid backward passes; P enters dV as bf16 in each):
 *   probs + mblk (what db1_relattn_flash_fwd stored) and db1_relattn_flash_bwd_workspace_bytes(B, L, H, 1) bytes of scratch:
 *       no score is recomputed -- P = p~ exp2(m_blk c2 - lse log2 e) on both sides, the factors go from the query side to the key side
 *       through the scratch (B*H*L*L/32 floats), and each side forms its own dS = P (dP - delta) scale from its own dP = dO.V^T
 *       (memory for time: B*H*L*L bf16 + B*H*L*L/32 floats per LAYER kept from the forward);
 *   no probs, db1_relattn_flash_bwd_workspace_bytes(B, L, H, 0) bytes of scratch (2 x B*H*L*L bf16): the query side recomputes scores,
 *       relative term and softmax and leaves P and dS in the scratch for the key side;
 *   neither (ws NULL / smaller): both sides recompute. */
int64_t db1_relattn_flash_bwd_workspace_bytes(int B, int L, int H, int have_probs);
int db1_relattn_flash_bwd(const void* qu, const void* qv, const void* k, const void* v, int64_t kv_row_stride,
                          int64_t kv_batch_stride, const void* R, const void* out, const void* dout, const float* lse,
                          float* delta, void* dq, void* dk, void* dv, int64_t dqkv_row_stride, int64_t dqkv_batch_stride,
                          void* dT, int B, int L, int H, int D, int shift, float scale, const void* probs, const float* mblk,
                          void* ws, int64_t ws_bytes, void* stream);

/* dq_r[b, i, h, :] = sum_{dist} dT[h, b, i, dist] * R[dist, h, :] (the (q+v).R branch of the query gradient, transformer_xl.py:160-209) as a
 * stream over dT with R stationary in registers: dT [H,B,L,L] bf16 (zero for dist > i), R [L, H*128] bf16 with row stride r_row_stride,
 * out [B,L,H,128] bf16 with row / batch strides in elements.  Same result as db1_gemm_strided_tri on the same operands. */
int db1_relattn_dqr_supported(int B, int L, int H, int D, int dt);
int64_t db1_relattn_dqr_workspace_bytes(int L, int H);   /* the transposed copy of R the stream kernel keeps in registers (+ the fused entry point's column-sum partials) */
int db1_relattn_dqr(const void* dT, const void* R, int64_t r_row_stride, void* out, int64_t out_row_stride, int64_t out_batch_stride,
                    int B, int L, int H, int D, void* ws, int64_t ws_bytes, void* stream);

/* The same stream with the rest of the query gradient in its epilogue (one pass less over two [B, L, H, 128] tensors):
 *   dq[b,i,h,:] = dq[b,i,h,:] + dq_r          (dq holds the (q+u).k branch from db1_relattn_flash_bwd on entry; one rounding)
 *   du_acc[h*128 + c] += sum_{b,i} dq_k[b,i,h,c]   (r_w_bias gradient),   dv_acc[h*128 + c] += sum_{b,i} dq_r[b,i,h,c]   (r_r_bias gradient)
 * float32 accumulators; the per-workgroup partial sums are added in a fixed order (deterministic). */
int db1_relattn_dqr_fused(const void* dT, const void* R, int64_t r_row_stride, void* dq, int64_t dq_row_stride, int64_t dq_batch_stride,
                          float* du_acc, float* dv_acc, int B, int L, int H, int D, void* ws, int64_t ws_bytes, void* stream);
/* ... WITHOUT its two reduces: parts [2][db1_relattn_dqr_parts_rows(H)][H * 128] float32 receives the per-workgroup column sums (dq_k -> du
 * half, dq_r -> dv half) for the caller to add up later (db1_colsum_acc per half) */
int db1_relattn_dqr_parts_rows(int H);
int db1_relattn_dqr_fused_parts(const void* dT, const void* R, int64_t r_row_stride, void* dq, int64_t dq_row_stride, int64_t dq_batch_stride,
                                float* parts, int B, int L, int H, int D, void* ws, int64_t ws_bytes, void* stream);
/* ... over a batch made of `ngroups` blocks of B / ngroups sequences, block g with its OWN R at R + g * r_group_stride elements: a whole
 * gradient-accumulation window run through ONE backward (TransformerXL's deferred backward; every micro-step drew its own position-table
 * dropout, transformer_xl.py:575, hence its own R = r_net(table), :138).  ngroups must divide B and the 256 / H workgroups of a head. */
int db1_relattn_dqr_groups_supported(int B, int L, int H, int D, int dt, int ngroups);
int64_t db1_relattn_dqr_groups_workspace_bytes(int L, int H, int ngroups);
int db1_relattn_dqr_fused_groups(const void* dT, const void* R, int64_t r_row_stride, int64_t r_group_stride, int ngroups, void* dq,
                                 int64_t dq_row_stride, int64_t dq_batch_stride, float* du_acc, float* dv_acc, int B, int L, int H, int D,
                                 void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ image-patch embedder pieces
 * (src/tokenizer/vision_embedding.py:65-86).  pixels [N_img, C, Himg, Wimg] -> normalised patches
 * [(n h w), C, p, p]: (x-mean)/(1e-6+std_unbiased)/sqrt(p) per (patch, channel). */
int db1_patch_normalize(const void* pixels, void* patches, int n_img, int C, int Himg, int Wimg, int p,
                        int dtIn, int dtOut, void* stream);
/* im2col for 3x3/pad 1 on p x p patches: x [N, C, p, p] -> cols [N*p*p, kpad] (kpad >= C*9 is the row stride; padding
 * columns are zero-filled so that K is a multiple of 8 for the MFMA tile GEMM); col2im gathers the reverse. */
int db1_im2col3x3(const void* x, void* cols, int64_t N, int C, int p, int kpad, int dt, void* stream);
int db1_col2im3x3(const void* dcols, void* dx, int64_t N, int C, int p, int kpad, int dt, void* stream);
/* Channels-last variants (the bf16 path): activations [N, p*p, C], column matrix tap-major cols[(n,y,x)][(ky*3+kx)*C + c], so every
 * access is a 16-byte vector of consecutive channels.  db1_conv_weight_permute builds the matching GEMM operand [Cout, kpad] from
 * the reference's [Cout, Cin, 3, 3] weight (vision_embedding.py:44-63), db1_conv_wgrad_unpermute adds a gradient computed in that
 * order back to the parameter layout.  The GroupNorm+GELU pair is specialised to the embedder's C = 64, hw = 256. */
int db1_patch_normalize_nhwc(const void* pixels, void* patches, int n_img, int C, int Himg, int Wimg, int p,
                             int dtIn, int dtOut, void* stream);
int db1_im2col3x3_nhwc(const void* x, void* cols, int64_t N, int C, int p, int kpad, int dt, void* stream);
int db1_col2im3x3_nhwc(const void* dcols, void* dx, int64_t N, int C, int p, int kpad, int dt, void* stream);
int db1_conv_weight_permute(const void* w, void* wp, int Cout, int Cin, int kpad, int dtIn, int dtOut, void* stream);
int db1_conv_wgrad_unpermute(const float* gp, float* g_acc, int Cout, int Cin, int kpad, void* stream);
/* Implicit-GEMM 3x3 convolutions for the 64 -> 64 channel layers on 16x16 patches (channels-last bf16): no column matrix.
 *   fwd (sign = +1): y[pix, o] = sum_{tap,c} x[pix + s(tap), c] * w_op[o, tap*64 + c] + bias[o], w_op from db1_conv_weight_permute;
 *   data gradient (sign = -1): x := dy, w_op := db1_conv_weight_permute_t(weight)  ([c, tap*64 + o]), bias = NULL;
 *   wgrad: gp_acc[o, tap*64 + c] += sum_pix dy[pix, o] * x[pix + s(tap), c]   (float32 [64, 576]; the pixel ranges' partial sums are added
 *          in a fixed order through the workspace: db1_conv3x3_implicit_wgrad_workspace_bytes, required). */
int db1_conv_weight_permute_t(const void* w, void* wp, int Cout, int Cin, int dtIn, int dtOut, void* stream);
int db1_conv3x3_implicit_fwd(const void* x, const void* w_op, const void* bias, void* y, int64_t n_patches, int sign, int dtBias,
                             void* stream);
/* same with a residual [n_patches * 256, 64] (bf16) added in the epilogue: y = conv + bias + res (the residual block's closing sum) */
int db1_conv3x3_implicit_fwd_res(const void* x, const void* w_op, const void* bias, const void* res, void* y, int64_t n_patches, int sign,
                                 int dtBias, void* stream);
/* The 3 -> 64 channel convolution of 16x16 patches (channels-last bf16 [n_patches*256, 3]) in one streaming kernel: y [n_patches*256, 64] =
 * conv + bias and, beside it, the column matrix cols [n_patches*256, 32] (tap-major, k = tap*3 + c, zero for k >= 27) that the weight
 * gradient contracts over; w_op [64, 32] as db1_conv_weight_permute leaves it. */
int db1_conv1_fused_fwd(const void* x, const void* w_op, const void* bias, void* cols, void* y, int64_t n_patches, int dtBias, void* stream);
int64_t db1_conv3x3_implicit_wgrad_workspace_bytes(int64_t n_patches);   /* per-pixel-range partial sums: with them the result is bit-reproducible */
int db1_conv3x3_implicit_wgrad(const void* dy, const void* x, float* gp_acc, float* gbias_acc /* optional [64]: += column sums of dy (the bias gradient) */,
                               int64_t n_patches, void* ws, int64_t ws_bytes, void* stream);
int db1_groupnorm_gelu_nhwc_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                int64_t N, int C, int hw, int groups, float eps, int dt, int dtParam, void* stream);
int64_t db1_groupnorm_gelu_nhwc_bwd_workspace_bytes(int64_t N);   /* per-sample parameter-gradient rows, summed in a fixed order (required) */
/* res (nullable): a tensor of dx's shape added to dx in fp32 before its one rounding -- the gradient that reaches x past the block
   (PatchEmbeddings' residual sum, vision_embedding.py:65-86), so that no separate add pass runs; res must not overlap dx */
int db1_groupnorm_gelu_nhwc_bwd(const void* dy, const void* x, const void* gamma, const void* beta, const float* mean,
                                const float* rstd, void* dx, const void* res, float* dgamma_acc, float* dbeta_acc,
                                int64_t N, int C, int hw, int groups, int dt, int dtParam, void* ws, int64_t ws_bytes, void* stream);
/* layout shuffles between GEMM output [N*p*p, C] ("NHWC") and [N, C, p, p] ("NCHW") */
int db1_nhwc_to_nchw(const void* x, void* y, int64_t N, int C, int hw, int dt, void* stream);
int db1_nchw_to_nhwc(const void* x, void* y, int64_t N, int C, int hw, int dt, void* stream);
/* GroupNorm(groups) + erf-GELU on [N, C, hw]; saves mean/rstd [N*groups]. bwd returns dx and accumulates dgamma/dbeta. */
int db1_groupnorm_gelu_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                           int64_t N, int C, int hw, int groups, float eps, int dt, int dtParam, void* stream);
int db1_groupnorm_gelu_bwd(const void* dy, const void* x, const void* gamma, const void* beta, const float* mean,
                           const float* rstd, void* dx, float* dgamma_acc, float* dbeta_acc,
                           int64_t N, int C, int hw, int groups, int dt, int dtParam, void* stream);

/* ------------------------------------------------------------------ optimizer
 * acc[0] += sum(x^2)  (global-norm clipping, train_config.py:211-215).  Workspace-free form: ONE workgroup, so the sum does not depend on
 * arrival order -- meant for small vectors (n <= 2^24, DB1_ERR_UNSUPPORTED above: one workgroup is 1 / 256 of the chip); the whole gradient
 * arena goes through db1_sumsq_det / db1_grad_norm_sq. */
int db1_sumsq_acc(const void* x, float* acc, int64_t n, int dt, void* stream);
/* the same sum over the whole chip (per-workgroup partials in the workspace, fixed-order final add); overwrite != 0: acc[0] = sum, else += */
int64_t db1_sumsq_det_workspace_bytes(int64_t n);
int db1_sumsq_det(const void* x, float* acc, int64_t n, int dt, int overwrite, void* ws, int64_t ws_bytes, void* stream);
/* One fused Adam/AdamW step over a flat segment (DeepSpeed FusedAdam stand-in; torch.optim semantics).
 * g: gradients, float32 or (dtGrad = DB1_BF16) the bf16 copy the data-parallel engine all-reduces (train.py:231-232);
 * p32/m/v: float32 state; p_work (nullable): bf16 working copy written alongside.
 * grad scale = gscale * min(1, clip / (sqrt(*norm_sq) * gscale + 1e-6)) when clip > 0 and norm_sq != NULL. */
int db1_adam_step(float* p32, const void* g, float* m, float* v, void* p_work, int64_t n,
                  double lr, double beta1, double beta2, double eps, double wd, int adamw, int step,
                  float gscale, float clip, const float* norm_sq, int dtGrad, int dtWork, void* stream);

/* ------------------------------------------------------------------ scalar tokenizer
 * ContinuousScalarTokenizer.discretize (src/tokenizer/scalar_tokenizer.py:28-45), bit-exact ids. */
int db1_mulaw_discretize(const float* x, int32_t* ids, int64_t n, int is_action, int num_bins, float mu, float M,
                         void* stream);
/* ContinuousScalarTokenizer.decode (scalar_tokenizer.py:47-63; caller evaluate_rl.py:263): ids (int32, or int64 when ids_are_int64)
 * clipped to [0, num_bins-1] -> x = id/num_bins*2 - 1; observations additionally sign(x) * ((1 + M*mu)^|x| - 1) / mu.
 * *oob_flag (device int, nullable) is OR-ed with 1 when an id was out of range (the reference warns and clips). */
int db1_mulaw_decode(const void* ids, float* out, int64_t n, int ids_are_int64, int is_action, int num_bins, float mu, float M,
                     int* oob_flag, void* stream);

/* ------------------------------------------------------------------ composite entry points: one call per block of the
 * reference's forward / backward, sequencing the launches above (csrc/composite.hip), so that a host in any language drives the hot
 * path without re-implementing the Python orchestration.  bf16 activations; same conventions (caller-owned device pointers, (ws, ws_bytes)
 * scratch with a size query, asynchronous on `stream`, int status). */
/* acc[0] = sum(g^2) over a flat gradient segment: the global-norm clip's reduction (train_config.py:211-215).  Deterministic: per-workgroup
 * partial sums through the workspace, added in a fixed order. */
int64_t db1_grad_norm_sq_workspace_bytes(int64_t n);
int db1_grad_norm_sq(const void* g, float* acc, int64_t n, int dt, void* ws, int64_t ws_bytes, void* stream);
/* backward of the tied head + masked CE (transformer_xl.py:593-613) from the (lse, sums) a previous db1_lmhead_ce_fwd left: the logits are
 * recomputed 16 384 rows at a time; dh [T, d] = d(loss * gscale) / dh, dW_acc [n_w_rows, d] (float32) = beta_dw * dW_acc + d(loss * gscale) / dW */
int64_t db1_lmhead_ce_bwd_workspace_bytes(int64_t T, int n_w_rows, int d, int chunk_rows, int dt);
int db1_lmhead_ce_bwd(const void* h, const void* W, const int64_t* labels, const float* mask, const float* lse, const float* sums, void* dh,
                      float* dW_acc, float beta_dw, float gscale, int64_t T, int V, int n_w_rows, int d, int chunk_rows, int dt, void* ws,
                      int64_t ws_bytes, void* stream);
/* RelPartialLearnableMultiHeadAttn score / softmax / P.V (transformer_xl.py:160-225, _rel_shift :98-110, mask :551-567) on the packed
 * projections qkv [B, L, 3, H, D] (bf16, D = 128, L % 128 == 0), u / vb [H, D], R [L, H, D] = r_net(position table):
 *   fwd: qu = q + u, qv = q + vb (outputs, [B, L, H, D], kept by the caller for the backward), out [B, L, H, D], lse [B, H, L];
 *        probs [B*H, L/32, L/16, 512] bf16 + mblk [B*H, L/32, L] float32 (both or neither): the forward keeps its probabilities;
 *   bwd: dqkv [B, L, 3, H, D] and dR [L, H, D] written, du_acc / dvb_acc [H, D] float32 += ; dT [H, B, L, L] bf16 is caller scratch that must
 *        be ZERO above the causal diagonal on entry (every other entry is rewritten: one zero-initialised buffer serves all calls);
 *        plain causal window only (shift >= L). */
int db1_relattn_fwd(const void* qkv, const void* u, const void* vb, const void* R, void* qu, void* qv, void* out, float* lse, void* probs,
                    float* mblk, int B, int L, int H, int D, int shift, float scale, void* stream);
int64_t db1_relattn_bwd_workspace_bytes(int B, int L, int H, int D, int have_probs);
int db1_relattn_bwd(const void* qkv, const void* qu, const void* qv, const void* R, const void* out, const void* dout, const float* lse,
                    const void* probs, const float* mblk, void* dqkv, void* dR, float* du_acc, float* dvb_acc, void* dT, int B, int L, int H,
                    int D, int shift, float scale, void* ws, int64_t ws_bytes, void* stream);
/* PatchEmbeddings.forward / its backward (vision_embedding.py:65-86) on 16 x 16 patches: pixels [n_img, C, Himg, Wimg] float32 ->
 * emb [N, d] bf16, N = n_img * (Himg / 16) * (Wimg / 16) (position embeddings are the caller's gather).  weights[12] (bf16, the reference's
 * layouts) / grads[12] (float32 accumulators, +=) in state-dict order: conv1.{weight,bias}, residual_path.0.{weight,bias},
 * residual_path.2.{weight,bias}, residual_path.3.{weight,bias}, residual_path.5.{weight,bias}, projection.{weight,bias}.
 * `save` (db1_patch_embed_save_bytes) carries the forward's activations to the backward. */
int64_t db1_patch_embed_save_bytes(int n_img, int C, int Himg, int Wimg, int p);
int64_t db1_patch_embed_workspace_bytes(int n_img, int C, int Himg, int Wimg, int p, int d, int backward);
int db1_patch_embed_fwd(const float* pixels, const void* const* weights, void* emb, void* save, int n_img, int C, int Himg, int Wimg, int p,
                        int d, void* ws, int64_t ws_bytes, void* stream);
int db1_patch_embed_bwd(const void* demb, const void* const* weights, const void* save, float* const* grads, int n_img, int C, int Himg,
                        int Wimg, int p, int d, void* ws, int64_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DB1_HIP_H */
