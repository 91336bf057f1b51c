// Deterministic embedding-table gradients (transformer_xl.py:621-672 backward: word, RL local-position and patch-position tables).
//     dtable[ids[t], :] += dout[t, :]
// without floating-point atomics: the tokens are ordered by table row with a STABLE device radix sort (rocPRIM) of (row, token index)
// pairs -- so the tokens of one row appear in token order -- and one wave per run of equal rows adds that run's dout rows in that order
// and is the only writer of its table row.  Same inputs -> same bits, whatever the scheduling (the atomic version it replaces was the
// last order-dependent reduction of the step).  The sort's scratch comes from the caller like every other workspace.
#include <cstring>   // (rocPRIM's headers call memset from host code)
#include <rocprim/device/device_radix_sort.hpp>
#include "db1_common.h"

#define SC_INVALID 0xFFFFFFFFu

__global__ __launch_bounds__(256) void scatter_keys_kernel(const int64_t* __restrict__ ids, unsigned* __restrict__ keys, unsigned* __restrict__ idx,
                                                           int64_t n, int64_t n_rows) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int64_t id = ids[t];
    keys[t] = (id >= 0 && id < n_rows) ? (unsigned)id : SC_INVALID;   // ids outside the table sort to the end and are skipped
    idx[t] = (unsigned)t;
}

// one workgroup per sorted position; only the first position of a run works: it walks the run (tokens in ascending order) and owns the
// table row.  The four waves split the columns (64 * V each per sweep) and the walk is batched eight tokens at a time -- eight index
// loads, then their row loads, then the adds in token order -- so that a long run (the 22 local-position rows of an RL batch take
// ~3 000 tokens each) is a stream, not a chain of dependent round trips.
template <typename T>
__global__ __launch_bounds__(256) void scatter_runs_kernel(const T* __restrict__ dout, int64_t ld, const unsigned* __restrict__ keys,
                                                           const unsigned* __restrict__ idx, float* __restrict__ dtable, int64_t n, int d) {
    constexpr int V = Vec16<T>::N, BATCH = 8;
    const int64_t p = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned key = keys[p];
    if (key == SC_INVALID || (p > 0 && keys[p - 1] == key)) return;
    float* trow = dtable + (int64_t)key * d;
    for (int c = (wave * 64 + lane) * V; c < d; c += 4 * 64 * V) {
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; j++) acc[j] = 0.f;
        int64_t q = p;
        bool more = true;
        while (more) {
            unsigned rows[BATCH];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < BATCH; u++) {          // (wave-uniform: every lane reads the same keys)
                const bool in = q + u < n && keys[q + u] == key;
                rows[u] = in ? idx[q + u] : 0u;
                cnt += in ? 1 : 0;
            }
            Vec16<T> v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; u++)
                if (u < cnt) v[u].load(dout + (int64_t)rows[u] * ld + c);
#pragma unroll
            for (int u = 0; u < BATCH; u++)
                if (u < cnt) {
#pragma unroll
                    for (int j = 0; j < V; j++) acc[j] += v[u].v[j];
                }
            q += cnt;
            more = cnt == BATCH;
        }
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            float4 o = *reinterpret_cast<float4*>(trow + c + j);
            o.x += acc[j]; o.y += acc[j + 1]; o.z += acc[j + 2]; o.w += acc[j + 3];
            *reinterpret_cast<float4*>(trow + c + j) = o;
        }
    }
}

static inline int64_t sc_al(int64_t x) { return (x + 255) & ~(int64_t)255; }
static int64_t sc_sort_temp_bytes(int64_t n) {
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tb, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (size_t)n, 0, 32, (hipStream_t)0, false);
    return (int64_t)tb;
}
extern "C" int64_t db1_embed_scatter_add_workspace_bytes(int64_t n_tokens) {
    if (n_tokens <= 0) return 0;
    return 4 * sc_al(n_tokens * 4) + sc_al(sc_sort_temp_bytes(n_tokens));
}

int db1_scatter_add_impl(const void* dout, const int64_t* ids, float* dtable_acc, int64_t n_tokens, int d, int64_t ld_dout, int64_t n_table_rows, int dt,
                         void* ws, int64_t ws_bytes, hipStream_t st, const char* who) {
    if (!db1_dt_ok(dt)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "%s: dtype", who);
    const int V = dt == DB1_F32 ? 4 : 8;
    if (n_tokens <= 0 || n_tokens > 0x7fffffffLL || d <= 0 || (d % V) || ld_dout < d || (ld_dout % V) || n_table_rows <= 0 || n_table_rows >= (int64_t)SC_INVALID)
        DB1_FAIL(DB1_ERR_BAD_SHAPE, "%s: n=%lld d=%d (a multiple of %d) ld=%lld rows=%lld", who, (long long)n_tokens, d, V, (long long)ld_dout, (long long)n_table_rows);
    if (!db1_aligned16(dout) || !db1_aligned16(dtable_acc)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "%s: alignment", who);
    DB1_NEED_WS(ws, ws_bytes, db1_embed_scatter_add_workspace_bytes(n_tokens), who);
    const int64_t seg = sc_al(n_tokens * 4);
    unsigned* keys_in = (unsigned*)ws;
    unsigned* keys_out = (unsigned*)((char*)ws + seg);
    unsigned* idx_in = (unsigned*)((char*)ws + 2 * seg);
    unsigned* idx_out = (unsigned*)((char*)ws + 3 * seg);
    void* temp = (char*)ws + 4 * seg;
    size_t tb = (size_t)(ws_bytes - 4 * seg);
    scatter_keys_kernel<<<(unsigned)((n_tokens + 255) / 256), 256, 0, st>>>(ids, keys_in, idx_in, n_tokens, n_table_rows);
    DB1_CHECK_LAUNCH(who);
    if (rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, idx_in, idx_out, (size_t)n_tokens, 0, 32, st, false) != hipSuccess)
        DB1_FAIL(DB1_ERR_HIP, "%s: radix sort", who);
    DB1_DISPATCH_DT(dt, T, (scatter_runs_kernel<T><<<(unsigned)n_tokens, 256, 0, st>>>((const T*)dout, ld_dout, keys_out, idx_out, dtable_acc, n_tokens, d)));
    DB1_CHECK_LAUNCH(who);
    return DB1_OK;
}

extern "C" int db1_embed_scatter_add_bwd(const void* dout, const int64_t* ids, float* dtable_acc, int64_t n_tokens, int d, int64_t ld_dout,
                                         int64_t n_table_rows, int dt, void* ws, int64_t ws_bytes, void* stream) {
    return db1_scatter_add_impl(dout, ids, dtable_acc, n_tokens, d, ld_dout, n_table_rows, dt, ws, ws_bytes, (hipStream_t)stream, "embed_scatter_add");
}
