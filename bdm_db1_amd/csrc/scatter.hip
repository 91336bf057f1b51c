// Deterministic embedding-table gradients (transformer_xl.py:621-672 backward: word, RL local-position and patch-position tables).
//     dtable[ids[t], :] += dout[t, :]
// without floating-point atomics: the tokens are ordered by table row with a STABLE radix sort of (row, token index) pairs -- so the
// tokens of one row appear in token order -- and one wave per run of equal rows adds that run's dout rows in that order and is the only
// writer of its table row.  Same inputs -> same bits, whatever the scheduling (the atomic version it replaces was the last
// order-dependent reduction of the step).  The sort is the library's own (rs_* below: least-significant-digit passes of 8 bits over as many
// bits as the table has rows -- two passes for the 33 025-row vocabulary, one for the 128-row position tables); its scratch comes from the
// caller like every other workspace.  (Rounds 2-4 called rocprim::radix_sort_pairs here: the last vendor primitive on the path.)
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

// one workgroup per sorted position; only the first position of a run works: it owns the table row.  The four waves split the columns
// (64 * V each per sweep) and the walk is batched sixteen tokens at a time -- the index loads, then their row loads, then the adds in token
// order -- so that a run is a stream, not a chain of dependent round trips.  Long runs (the 22 local-position rows of an RL batch take
// ~3 000 tokens each: one workgroup per row streamed 12 MB at 10 GB/s, 1.3 ms) are cut at the multiples of SC_CHUNK sorted positions: the
// part of a run inside a later chunk is summed by that chunk's own workgroup into a partial row (scatter_chunk_kernel, launched first),
// and the owner adds its own head part, then the partial rows in chunk order.  The cut points depend on the sorted order only: same
// inputs -> same bits.
#define SC_CHUNK 256
#define SC_PART_MAX_D 8192      // the workspace holds partial rows for d up to this (larger d: the owner walks its whole run)
template <typename T, int BATCH>
__device__ __forceinline__ void sc_sum_range(const T* __restrict__ dout, int64_t ld, const unsigned* __restrict__ keys, const unsigned* __restrict__ idx,
                                             unsigned key, int64_t q, int64_t qend, int c, float (&acc)[Vec16<T>::N]) {
    constexpr int V = Vec16<T>::N;
    bool more = true;
    while (more) {
        unsigned rows[BATCH];
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < BATCH; u++) {          // (wave-uniform: every lane reads the same keys)
            const bool in = q + u < qend && keys[q + u] == key;
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
}
// chunk ch >= 1 whose first position continues a run: partial[ch][:] = sum of that run's tokens inside the chunk, in token order
template <typename T>
__global__ __launch_bounds__(256) void scatter_chunk_kernel(const T* __restrict__ dout, int64_t ld, const unsigned* __restrict__ keys,
                                                            const unsigned* __restrict__ idx, float* __restrict__ part, int64_t n, int d) {
    constexpr int V = Vec16<T>::N;
    const int64_t ch = (int64_t)blockIdx.x + 1, p = ch * SC_CHUNK;
    const unsigned key = keys[p];
    if (key == SC_INVALID || keys[p - 1] != key) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t qend = p + SC_CHUNK < n ? p + SC_CHUNK : n;
    for (int c = (wave * 64 + lane) * V; c < d; c += 4 * 64 * V) {
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; j++) acc[j] = 0.f;
        sc_sum_range<T, 16>(dout, ld, keys, idx, key, p, qend, c, acc);
#pragma unroll
        for (int j = 0; j < V; j += 4) *reinterpret_cast<float4*>(part + ch * d + c + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void scatter_runs_kernel(const T* __restrict__ dout, int64_t ld, const unsigned* __restrict__ keys,
                                                           const unsigned* __restrict__ idx, float* __restrict__ dtable, const float* __restrict__ part,
                                                           int64_t n, int d) {
    constexpr int V = Vec16<T>::N;
    const int64_t p = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned key = keys[p];
    if (key == SC_INVALID || (p > 0 && keys[p - 1] == key)) return;
    float* trow = dtable + (int64_t)key * d;
    const int64_t head_end = part ? ((p / SC_CHUNK + 1) * SC_CHUNK < n ? (p / SC_CHUNK + 1) * SC_CHUNK : n) : n;   // (no partials: the whole run)
    for (int c = (wave * 64 + lane) * V; c < d; c += 4 * 64 * V) {
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; j++) acc[j] = 0.f;
        sc_sum_range<T, 16>(dout, ld, keys, idx, key, p, head_end, c, acc);
        if (part) {
            for (int64_t ch = p / SC_CHUNK + 1; ch * SC_CHUNK < n && keys[ch * SC_CHUNK] == key; ch++) {   // the run's later chunks, in order
#pragma unroll
                for (int j = 0; j < V; j += 4) {
                    const float4 o = *reinterpret_cast<const float4*>(part + ch * d + c + j);
                    acc[j] += o.x; acc[j + 1] += o.y; acc[j + 2] += o.z; acc[j + 3] += o.w;
                }
            }
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

// ---- stable LSD radix sort of (key, value) pairs, 8 bits per pass.  A pass = per-tile digit histograms (one wave per RS_TILE keys) ->
// exclusive scan over (digit, tile) in digit-major order (one workgroup) -> stable scatter: the wave of a tile walks its keys 64 at a
// time; lanes holding the same digit find each other with eight ballots (rank = lanes of the group below me), the group's running
// offset lives in LDS.  Integer arithmetic only: one correct output.  Keys above `cap` (SC_INVALID) sort as cap, i.e. behind every row.
#define RS_TILE 512
__device__ __forceinline__ unsigned rs_digit(unsigned k, unsigned cap, int shift) { return ((k > cap ? cap : k) >> shift) & 255u; }
__global__ __launch_bounds__(64) void rs_hist_kernel(const unsigned* __restrict__ keys, unsigned* __restrict__ hist, int64_t n, unsigned cap, int shift, int nb) {
    __shared__ unsigned h[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) h[i] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int s = 0; s < RS_TILE / 64; s++) {
        const int64_t i = base + s * 64 + lane;
        if (i < n) atomicAdd(&h[rs_digit(keys[i], cap, shift)], 1u);     // (integer counts in LDS: order-independent)
    }
    __syncthreads();
    for (int i = lane; i < 256; i += 64) hist[(int64_t)i * nb + blockIdx.x] = h[i];
}
__global__ __launch_bounds__(1024) void rs_scan_kernel(unsigned* __restrict__ hist, int64_t total) {   // exclusive prefix sums in place, one workgroup
    __shared__ unsigned sm[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (int64_t base = 0; base < total; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const unsigned v = i < total ? hist[i] : 0u;
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned add = threadIdx.x >= off ? sm[threadIdx.x - off] : 0u;
            __syncthreads();
            sm[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < total) hist[i] = carry + sm[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sm[1023];
        __syncthreads();
    }
}
__global__ __launch_bounds__(64) void rs_scatter_kernel(const unsigned* __restrict__ kin, const unsigned* __restrict__ vin, unsigned* __restrict__ kout,
                                                        unsigned* __restrict__ vout, const unsigned* __restrict__ hist, int64_t n, unsigned cap, int shift, int nb) {
    __shared__ unsigned off[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) off[i] = hist[(int64_t)i * nb + blockIdx.x];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int s = 0; s < RS_TILE / 64; s++) {
        const int64_t i = base + s * 64 + lane;
        const bool act = i < n;
        const unsigned k = act ? kin[i] : 0u, v = act ? vin[i] : 0u;
        const unsigned dg = rs_digit(k, cap, shift);
        unsigned long long grp = __ballot(act);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bal = __ballot(act && ((dg >> b) & 1u));
            grp &= ((dg >> b) & 1u) ? bal : ~bal;
        }
        const unsigned rank = (unsigned)__popcll(grp & ((1ull << lane) - 1ull));
        const unsigned pos = off[dg] + rank;
        __syncthreads();                                   // (every lane has read its group's offset)
        if (act) {
            kout[pos] = k;
            vout[pos] = v;
            if (rank == 0u) off[dg] += (unsigned)__popcll(grp);   // one lane per group: the lowest
        }
        __syncthreads();
    }
}
static int64_t rs_temp_bytes(int64_t n) { return sc_al(256 * ((n + RS_TILE - 1) / RS_TILE) * (int64_t)sizeof(unsigned)); }
static int64_t sc_sort_temp_bytes(int64_t n) { return rs_temp_bytes(n); }
static int64_t sc_part_bytes(int64_t n) { return sc_al(((n + SC_CHUNK - 1) / SC_CHUNK) * (int64_t)SC_PART_MAX_D * (int64_t)sizeof(float)); }
extern "C" int64_t db1_embed_scatter_add_workspace_bytes(int64_t n_tokens) {
    if (n_tokens <= 0) return 0;
    return 4 * sc_al(n_tokens * 4) + sc_al(sc_sort_temp_bytes(n_tokens)) + sc_part_bytes(n_tokens);
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
    float* part = d <= SC_PART_MAX_D ? (float*)((char*)ws + 4 * seg) : nullptr;      // [chunk][d] partial rows of the runs that cross chunk boundaries
    void* temp = (char*)ws + 4 * seg + sc_part_bytes(n_tokens);
    size_t tb = (size_t)(ws_bytes - 4 * seg - sc_part_bytes(n_tokens));
    scatter_keys_kernel<<<(unsigned)((n_tokens + 255) / 256), 256, 0, st>>>(ids, keys_in, idx_in, n_tokens, n_table_rows);
    DB1_CHECK_LAUNCH(who);
    (void)tb;
    {   // stable sort of (row, token): ceil(bits(n_table_rows) / 8) passes, ping-pong between the two buffer pairs
        const unsigned cap = (unsigned)n_table_rows;
        int bits = 0;
        while (bits < 32 && (cap >> bits)) bits++;
        const int nb = (int)((n_tokens + RS_TILE - 1) / RS_TILE);
        unsigned* hist = (unsigned*)temp;
        for (int shift = 0; shift < bits; shift += 8) {
            rs_hist_kernel<<<(unsigned)nb, 64, 0, st>>>(keys_in, hist, n_tokens, cap, shift, nb);
            rs_scan_kernel<<<1, 1024, 0, st>>>(hist, (int64_t)256 * nb);
            rs_scatter_kernel<<<(unsigned)nb, 64, 0, st>>>(keys_in, idx_in, keys_out, idx_out, hist, n_tokens, cap, shift, nb);
            DB1_CHECK_LAUNCH(who);
            unsigned* t0 = keys_in; keys_in = keys_out; keys_out = t0;
            unsigned* t1 = idx_in; idx_in = idx_out; idx_out = t1;
        }
        keys_out = keys_in;    // (after the last swap the sorted pairs are in the "in" pair)
        idx_out = idx_in;
    }
    const int64_t nchunk = (n_tokens + SC_CHUNK - 1) / SC_CHUNK;
    if (part && nchunk > 1) {
        DB1_DISPATCH_DT(dt, T, (scatter_chunk_kernel<T><<<(unsigned)(nchunk - 1), 256, 0, st>>>((const T*)dout, ld_dout, keys_out, idx_out, part, n_tokens, d)));
        DB1_CHECK_LAUNCH(who);
    }
    DB1_DISPATCH_DT(dt, T, (scatter_runs_kernel<T><<<(unsigned)n_tokens, 256, 0, st>>>((const T*)dout, ld_dout, keys_out, idx_out, dtable_acc, part, n_tokens, d)));
    DB1_CHECK_LAUNCH(who);
    return DB1_OK;
}

extern "C" int db1_embed_scatter_add_bwd(const void* dout, const int64_t* ids, float* dtable_acc, int64_t n_tokens, int d, int64_t ld_dout,
                                         int64_t n_table_rows, int dt, void* ws, int64_t ws_bytes, void* stream) {
    return db1_scatter_add_impl(dout, ids, dtable_acc, n_tokens, d, ld_dout, n_table_rows, dt, ws, ws_bytes, (hipStream_t)stream, "embed_scatter_add");
}
