// Relative-position attention for inference with Transformer-XL memory (evaluate_rl.py:157-266; transformer_xl.py:124-133,
// 160-225): 1 .. 64 new queries against klen = mlen + q cached keys / values (bf16, d_head = 128).
//     s[i,j] = ((q_i+u).k_j + (q_i+v).R[mlen+i-j]) / sqrt(d),   visible iff  i - shift < j <= i + mlen,   j in [0, klen)
// The materialised path ran two batched GEMMs, a softmax and another batched GEMM per layer on the generic strided kernel
// (135 us for P.V alone at q = 1); here one launch covers them, split over the keys ("flash decoding"):
//   grid = (key chunks of 64, heads, batch); the 4 waves of a workgroup are the 4 possible 16-query tiles (a wave whose tile
//   is empty exits); every wave is self-contained (own LDS: a 32-key V tile for the transposed reads + the skew scratch).
//   Per 32-key step and wave: S^T = K.Qu^T (2 x 4 v_mfma_f32_16x16x32_bf16, K fragments straight from global), the relative
//   term for the 47 distances the 16 x 32 block touches (3 x 4 MFMA against R rows) written to LDS [48 dist][16 q] and read
//   back skewed (element (a, a - b + 31)), online softmax (statistics lane-local + two cross-group shuffles), O^T += V^T.P^T
//   (8 MFMA; V^T via ds_read_b64_tr_b16 from the staged rows, with the key permutation that makes the S registers directly
//   the B operand: slots 0-3 = keys 4g..4g+3 of the first 16-key block, slots 4-7 = the same keys of the second).
//   Partials (m, l, O) per chunk go to a workspace; relattn_decode_merge_kernel combines them in chunk order.
#include "db1_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

#define DEC_D 128
#define DEC_KC 64                        // keys per workgroup (two 32-key steps: the chunks, not the loop, provide the parallelism)
#define DEC_WAVE_LDS (32 * 256 + 48 * 16 * 4)  // V tile [32][128] bf16 + T scratch [48][16] f32

struct DecodeArgs {
    const bf16_t* qu; const bf16_t* qv; const bf16_t* k; const bf16_t* v; const bf16_t* R;
    float* part;   // [B][H][nchunk][64 queries][D + 2]: O (unnormalised), m, l
    bf16_t* out;
    int64_t kv_rs, kv_bs;
    int B, q, klen, mlen, H, shift, nd, nchunk;
    float scale;
};

typedef __attribute__((ext_vector_type(2))) float dec_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 dec_bf16x2;
__device__ __forceinline__ unsigned dec_pk(float lo, float hi) {
    dec_f32x2 v = {lo, hi};
    dec_bf16x2 r = __builtin_convertvector(v, dec_bf16x2);
    return *reinterpret_cast<unsigned*>(&r);
}

__global__ __launch_bounds__(256) void relattn_decode_kernel(DecodeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = wave * 16;
    if (i0 >= p.q) return;  // waves are independent: no workgroup barrier below
    const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int a = lane & 15, g = lane >> 4;
    const int HD = p.H * DEC_D;
    char* Vs = smem + wave * DEC_WAVE_LDS;
    float* Tw = reinterpret_cast<float*>(Vs + 32 * 256);
    const int qi = i0 + a < p.q ? i0 + a : p.q - 1;  // rows beyond q repeat the last query (never stored)
    const bf16_t* qu = p.qu + ((int64_t)b * p.q + qi) * HD + h * DEC_D + g * 8;
    const bf16_t* qv = p.qv + ((int64_t)b * p.q + qi) * HD + h * DEC_D + g * 8;
    bf16x8_t fqu[4], fqv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        fqu[ks] = *reinterpret_cast<const bf16x8_t*>(qu + ks * 32);
        fqv[ks] = *reinterpret_cast<const bf16x8_t*>(qv + ks * 32);
    }
    const bf16_t* kg = p.k + (int64_t)b * p.kv_bs + h * DEC_D;
    const bf16_t* vg = p.v + (int64_t)b * p.kv_bs + h * DEC_D;
    const bf16_t* Rg = p.R + h * DEC_D;
    f32x4 acc_o[8];
#pragma unroll
    for (int db = 0; db < 8; db++) acc_o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_i = -1.0e30f, l_i = 0.f;
    const int i = i0 + a;  // this lane's query
    const int jc0 = chunk * DEC_KC;
    for (int st = 0; st < DEC_KC / 32; st++) {
        const int j0 = jc0 + st * 32;
        if (j0 >= p.klen) break;
        // whole step outside the window of every query of the tile?  (wave-uniform)
        if (j0 > i0 + 15 + p.mlen || j0 + 31 <= i0 - p.shift) continue;
        // ---- stage V rows j0 .. j0+31 (rows past klen are clamped: their probabilities are exactly zero)
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4), ch = lane & 15;
            const int jr = j0 + row < p.klen ? j0 + row : p.klen - 1;
            const uint4 val = *reinterpret_cast<const uint4*>(vg + (int64_t)jr * p.kv_rs + ch * 8);
            *reinterpret_cast<uint4*>(Vs + row * 256 + ch * 16) = val;
        }
        // ---- S^T[key][query] for the two 16-key blocks
        f32x4 acc_s[2];
#pragma unroll
        for (int blk = 0; blk < 2; blk++) {
            acc_s[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int jr = j0 + 16 * blk + a < p.klen ? j0 + 16 * blk + a : p.klen - 1;
            const bf16_t* kr = kg + (int64_t)jr * p.kv_rs + g * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                acc_s[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(kr + ks * 32), fqu[ks], acc_s[blk], 0, 0, 0);
        }
        // ---- relative term: distances d_lo .. d_lo+47 with d_lo = mlen + i0 - j0 - 31  ->  Tw[dist][query]
        const int d_lo = p.mlen + i0 - j0 - 31;
#pragma unroll
        for (int tb = 0; tb < 3; tb++) {
            int dr = d_lo + 16 * tb + a;
            dr = dr < 0 ? 0 : (dr > p.nd - 1 ? p.nd - 1 : dr);  // out-of-range distances belong to masked pairs
            const bf16_t* rr = Rg + (int64_t)dr * HD + g * 8;
            f32x4 acc_t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                acc_t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(rr + ks * 32), fqv[ks], acc_t, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) Tw[(16 * tb + 4 * g + r) * 16 + a] = acc_t[r];
        }
        // (same wave wrote and reads the scratch; LDS operations of a wave complete in order)
        float s[8];
#pragma unroll
        for (int blk = 0; blk < 2; blk++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int bcol = 16 * blk + 4 * g + r;  // key j0 + bcol
                const int j = j0 + bcol;
                const float v = (acc_s[blk][r] + Tw[(a - bcol + 31) * 16 + a]) * p.scale;
                const bool vis = (j < p.klen) && (j <= i + p.mlen) && (j > i - p.shift) && (i < p.q);
                s[blk * 4 + r] = vis ? v : -1.0e30f;
            }
        float mb = s[0];
#pragma unroll
        for (int t = 1; t < 8; t++) mb = fmaxf(mb, s[t]);
        mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float m_new = fmaxf(m_i, mb);
        const float alpha = __expf(m_i - m_new);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < 8; t++) { s[t] = s[t] > -1.0e29f ? __expf(s[t] - m_new) : 0.f; rs += s[t]; }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_i = l_i * alpha + rs;
        m_i = m_new;
        union { unsigned u[4]; bf16x8_t v; } pb;
#pragma unroll
        for (int t = 0; t < 4; t++) pb.u[t] = dec_pk(s[2 * t], s[2 * t + 1]);
        // ---- O^T[d][query] += V^T . P^T over the 32 keys (key permutation as described in the header)
#pragma unroll
        for (int db = 0; db < 8; db++) {
            const int t16 = lane & 15;
            const char* base = Vs + (t16 >> 2) * 256 + (db * 16 + (t16 & 3) * 4) * 2;
            const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4_t, const_cast<char*>(base) + (4 * g) * 256));
            const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4_t, const_cast<char*>(base) + (16 + 4 * g) * 256));
            const bf16x8_t vt = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int r = 0; r < 4; r++) acc_o[db][r] *= alpha;
            acc_o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pb.v, acc_o[db], 0, 0, 0);
        }
    }
    // ---- partial results of this chunk: lane holds query a, d = db*16 + 4g + r
    if (i < p.q) {
        float* dst = p.part + ((((int64_t)b * p.H + h) * p.nchunk + chunk) * 64 + i) * (DEC_D + 2);
#pragma unroll
        for (int db = 0; db < 8; db++)
            *reinterpret_cast<float2*>(dst + db * 16 + 4 * g) = make_float2(acc_o[db][0], acc_o[db][1]),
            *reinterpret_cast<float2*>(dst + db * 16 + 4 * g + 2) = make_float2(acc_o[db][2], acc_o[db][3]);
        if (g == 0) { dst[DEC_D] = m_i; dst[DEC_D + 1] = l_i; }
    }
}

// one 128-thread block per (query, head, batch): combine the chunk partials in chunk order
__global__ __launch_bounds__(128) void relattn_decode_merge_kernel(DecodeArgs p) {
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
    const float* src = p.part + ((((int64_t)b * p.H + h) * p.nchunk) * 64 + i) * (DEC_D + 2);
    const int64_t cs = (int64_t)64 * (DEC_D + 2);
    float m = -1.0e30f;
    for (int c = 0; c < p.nchunk; c++) m = fmaxf(m, src[c * cs + DEC_D]);
    float l = 0.f, o = 0.f;
    for (int c = 0; c < p.nchunk; c++) {
        const float lc = src[c * cs + DEC_D + 1];
        if (lc > 0.f) {
            const float w = __expf(src[c * cs + DEC_D] - m);
            l += lc * w;
            o += src[c * cs + d] * w;
        }
    }
    p.out[((int64_t)b * p.q + i) * p.H * DEC_D + h * DEC_D + d] = f2bf(l > 0.f ? o / l : 0.f);
}

// ======================================================================================= ring variant (hipGraph-friendly inference)
// The K / V of the memory live in a ring [B, cap, 2, H, D] (cap >= mlen + q): logical key j < mlen is ring row (start + j) % cap, the
// q new keys / values are read from the packed projections of this call, qkv_new [B, q, 3, H, D], and copied into ring rows
// (start + mlen + i) % cap by the workgroups that visit them -- so a call appends in place instead of concatenating 8 MB per layer, and
// `start` is a DEVICE scalar (advanced by db1_ring_advance after the last layer): the same captured graph serves every call.
// q + u and q + v are formed here from qkv_new (rounded to bf16 like db1_relattn_add_head_bias does).
// SPLIT (what the host launches): a workgroup = one 16-query tile x one 128-key chunk, its four waves are the chunk's four 32-key steps,
// merged through LDS into one partial per (tile, chunk).  (The other form -- a wave is one of four 16-query tiles and walks the chunk's
// steps itself -- took 38.6 us per launch at q = 22 against 8.8 us at q = 1: four dependent rounds of loads per wave.)
#define DECR_KC 128
struct DecodeRingArgs {
    const bf16_t* qkv; const bf16_t* u; const bf16_t* vb; bf16_t* ring; const int* state; const bf16_t* R;
    float* part; bf16_t* out; unsigned* tickets;
    int B, q, klen, mlen, H, shift, nd, nunit, cap;
    float scale;
};
__device__ __forceinline__ bf16x8_t dec_add_bias(const bf16x8_t a, const bf16x8_t b) {
    union { unsigned u[4]; bf16x8_t v; } o;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float x0 = __uint_as_float((unsigned)(unsigned short)a[2 * t] << 16) + __uint_as_float((unsigned)(unsigned short)b[2 * t] << 16);
        const float x1 = __uint_as_float((unsigned)(unsigned short)a[2 * t + 1] << 16) + __uint_as_float((unsigned)(unsigned short)b[2 * t + 1] << 16);
        o.u[t] = dec_pk(x0, x1);
    }
    return o.v;
}
__device__ __forceinline__ void dec_st_agent(float* p, float a, float b) {   // 8 bytes, write-through (sc1)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dec_ld_agent2(const float* p, float& a, float& b) {   // 8 bytes, L1-bypassing (sc1)
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __uint_as_float((unsigned)v); b = __uint_as_float((unsigned)(v >> 32));
}
template <bool SPLIT>
__global__ __launch_bounds__(256) void relattn_decode_ring_kernel(DecodeRingArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // SPLIT: blockIdx.x = chunk * (number of 16-query tiles) + tile -- every workgroup is ONE query tile against ONE 128-key chunk
    const int nqt = SPLIT ? (p.q + 15) / 16 : 1;
    const int i0 = SPLIT ? ((int)blockIdx.x % nqt) * 16 : wave * 16;
    const int chunk = SPLIT ? (int)blockIdx.x / nqt : (int)blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int a = lane & 15, g = lane >> 4;
    const int HD = p.H * DEC_D;
    char* Vs = smem + wave * DEC_WAVE_LDS;
    float* Tw = reinterpret_cast<float*>(Vs + 32 * 256);
    const int qi = i0 + a < p.q ? i0 + a : p.q - 1;  // rows beyond q repeat the last query (never stored)
    const bf16_t* qrow = p.qkv + ((int64_t)b * p.q + qi) * 3 * HD + h * DEC_D + g * 8;
    // (requested before the append so that they travel while the ring origin arrives and the new rows are copied)
    bf16x8_t qraw[4], uraw[4], vraw[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        qraw[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 32);
        uraw[ks] = *reinterpret_cast<const bf16x8_t*>(p.u + h * DEC_D + g * 8 + ks * 32);
        vraw[ks] = *reinterpret_cast<const bf16x8_t*>(p.vb + h * DEC_D + g * 8 + ks * 32);
    }
    {   // append: the new keys / values of this chunk enter the ring (all 256 threads; 32 sixteen-byte pieces per row and head)
        const int start_ = p.state[0];
        const int ja = chunk * DECR_KC > p.mlen ? chunk * DECR_KC : p.mlen, jb = (chunk + 1) * DECR_KC < p.klen ? (chunk + 1) * DECR_KC : p.klen;
        const bf16_t* nk = p.qkv + (int64_t)b * p.q * 3 * HD + HD + h * DEC_D;
        bf16_t* rg = p.ring + (int64_t)b * p.cap * 2 * HD + h * DEC_D;
        for (int idx = threadIdx.x; idx < ((!SPLIT || i0 == 0) ? (jb - ja) * 32 : 0); idx += 256) {   // (one query tile of the chunk copies)
            const int j = ja + (idx >> 5), pc = idx & 31;          // piece 0..15: K, 16..31: V
            int r = start_ + j;
            r = r >= p.cap ? r - p.cap : r;
            *reinterpret_cast<uint4*>(rg + (int64_t)r * 2 * HD + (pc >> 4) * HD + (pc & 15) * 8) =
                *reinterpret_cast<const uint4*>(nk + (int64_t)(j - p.mlen) * 3 * HD + (pc >> 4) * HD + (pc & 15) * 8);
        }
    }
    if (i0 >= p.q) return;  // waves are independent: no workgroup barrier below
    bf16x8_t fqu[4], fqv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        fqu[ks] = dec_add_bias(qraw[ks], uraw[ks]);
        fqv[ks] = dec_add_bias(qraw[ks], vraw[ks]);
    }
    const int start = p.state[0];
    bf16_t* ring = p.ring + (int64_t)b * p.cap * 2 * HD + h * DEC_D;        // row r: K at r * 2 HD, V at r * 2 HD + HD
    const bf16_t* newkv = p.qkv + (int64_t)b * p.q * 3 * HD + HD + h * DEC_D;  // new row i: K at i * 3 HD, V at i * 3 HD + HD
    // K row of logical key j (V: + HD); j clamped by the caller.  A select, not a branch: with a divergent if / else around them the
    // compiler issued the 28 loads of a step one round trip at a time (14.6 us per launch at q = 1, almost all of it memory latency)
    auto krow = [&](int j) -> const bf16_t* {
        int r = start + j;
        r = r >= p.cap ? r - p.cap : r;
        const int64_t off_ring = (int64_t)r * 2 * HD, off_new = (int64_t)(j - p.mlen) * 3 * HD;
        const bf16_t* base = j >= p.mlen ? newkv : ring;
        return base + (j >= p.mlen ? off_new : off_ring);
    };
    const bf16_t* Rg = p.R + h * DEC_D;
    f32x4 acc_o[8];
#pragma unroll
    for (int db = 0; db < 8; db++) acc_o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_i = -1.0e30f, l_i = 0.f;
    const int i = i0 + a;  // this lane's query
    const int jc0 = chunk * DECR_KC;
    for (int st = SPLIT ? wave : 0; st < (SPLIT ? wave + 1 : DECR_KC / 32); st++) {
        const int j0 = jc0 + st * 32;
        if (j0 >= p.klen) break;
        if (j0 > i0 + 15 + p.mlen || j0 + 31 <= i0 - p.shift) continue;   // whole step outside the window of every query of the tile (wave-uniform)
        // ---- every global load of the step is requested before the first use: V rows j0 .. j0+31 (rows past klen are clamped: their
        // probabilities are exactly zero), the two 16-key K blocks and the three 16-distance R blocks (d_lo = mlen + i0 - j0 - 31)
        uint4 vreg[8];
        bf16x8_t kreg[2][4], rreg[3][4];
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4), ch = lane & 15;
            const int jr = j0 + row < p.klen ? j0 + row : p.klen - 1;
            vreg[it] = *reinterpret_cast<const uint4*>(krow(jr) + HD + ch * 8);
        }
#pragma unroll
        for (int blk = 0; blk < 2; blk++) {
            const int jr = j0 + 16 * blk + a < p.klen ? j0 + 16 * blk + a : p.klen - 1;
            const bf16_t* kr = krow(jr) + g * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) kreg[blk][ks] = *reinterpret_cast<const bf16x8_t*>(kr + ks * 32);
        }
        const int d_lo = p.mlen + i0 - j0 - 31;
#pragma unroll
        for (int tb = 0; tb < 3; tb++) {
            int dr = d_lo + 16 * tb + a;
            dr = dr < 0 ? 0 : (dr > p.nd - 1 ? p.nd - 1 : dr);  // out-of-range distances belong to masked pairs
            const bf16_t* rr = Rg + (int64_t)dr * HD + g * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) rreg[tb][ks] = *reinterpret_cast<const bf16x8_t*>(rr + ks * 32);
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4), ch = lane & 15;
            *reinterpret_cast<uint4*>(Vs + row * 256 + ch * 16) = vreg[it];
        }
        // ---- S^T[key][query] for the two 16-key blocks
        f32x4 acc_s[2];
#pragma unroll
        for (int blk = 0; blk < 2; blk++) {
            acc_s[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ks++) acc_s[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kreg[blk][ks], fqu[ks], acc_s[blk], 0, 0, 0);
        }
        // ---- relative term  ->  Tw[dist][query]
#pragma unroll
        for (int tb = 0; tb < 3; tb++) {
            f32x4 acc_t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ks++) acc_t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rreg[tb][ks], fqv[ks], acc_t, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) Tw[(16 * tb + 4 * g + r) * 16 + a] = acc_t[r];
        }
        float s[8];
#pragma unroll
        for (int blk = 0; blk < 2; blk++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int bcol = 16 * blk + 4 * g + r;  // key j0 + bcol
                const int j = j0 + bcol;
                const float v = (acc_s[blk][r] + Tw[(a - bcol + 31) * 16 + a]) * p.scale;
                const bool vis = (j < p.klen) && (j <= i + p.mlen) && (j > i - p.shift) && (i < p.q);
                s[blk * 4 + r] = vis ? v : -1.0e30f;
            }
        float mb = s[0];
#pragma unroll
        for (int t = 1; t < 8; t++) mb = fmaxf(mb, s[t]);
        mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float m_new = fmaxf(m_i, mb);
        const float alpha = __expf(m_i - m_new);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < 8; t++) { s[t] = s[t] > -1.0e29f ? __expf(s[t] - m_new) : 0.f; rs += s[t]; }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_i = l_i * alpha + rs;
        m_i = m_new;
        union { unsigned u[4]; bf16x8_t v; } pb;
#pragma unroll
        for (int t = 0; t < 4; t++) pb.u[t] = dec_pk(s[2 * t], s[2 * t + 1]);
#pragma unroll
        for (int db = 0; db < 8; db++) {
            const int t16 = lane & 15;
            const char* base = Vs + (t16 >> 2) * 256 + (db * 16 + (t16 & 3) * 4) * 2;
            const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4_t, const_cast<char*>(base) + (4 * g) * 256));
            const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4_t, const_cast<char*>(base) + (16 + 4 * g) * 256));
            const bf16x8_t vt = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int r = 0; r < 4; r++) acc_o[db][r] *= alpha;
            acc_o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pb.v, acc_o[db], 0, 0, 0);
        }
    }
    if (!SPLIT) {
        if (i < p.q) {
            float* dst = p.part + ((((int64_t)b * p.H + h) * p.nunit + chunk) * 64 + i) * (DEC_D + 2);
#pragma unroll
            for (int db = 0; db < 8; db++)
                *reinterpret_cast<float2*>(dst + db * 16 + 4 * g) = make_float2(acc_o[db][0], acc_o[db][1]),
                *reinterpret_cast<float2*>(dst + db * 16 + 4 * g + 2) = make_float2(acc_o[db][2], acc_o[db][3]);
            if (g == 0) { dst[DEC_D] = m_i; dst[DEC_D + 1] = l_i; }
        }
        return;
    }
    // ---- SPLIT: the four waves' partial results (one 32-key step each) are merged here, through LDS, into ONE unit per chunk (every wave
    // reaches this point: i0 = 0 < q, and break / continue above only skip the step)
    constexpr int OLD = DEC_D + 4;                       // floats per query row in LDS: O[128], m, l, pad
    float* mine = reinterpret_cast<float*>(Vs);          // (the wave's own V stage / T scratch, 11 KB >= 16 rows x 132 floats: it is done with them)
#pragma unroll
    for (int db = 0; db < 8; db++) *reinterpret_cast<f32x4*>(mine + a * OLD + db * 16 + 4 * g) = acc_o[db];
    if (g == 0) { mine[a * OLD + DEC_D] = m_i; mine[a * OLD + DEC_D + 1] = l_i; }
    __syncthreads();
    const bool inlaunch = p.tickets != nullptr;   // the merge over the chunks happens in this launch too: partials as write-through stores
    {
        const int qi = threadIdx.x & 15, dg = (threadIdx.x >> 4) * 8;
        const float* w0p = reinterpret_cast<const float*>(smem) + qi * OLD;
        float mw[4], lw[4], mx = -1.0e30f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            mw[w] = w0p[w * (DEC_WAVE_LDS / 4) + DEC_D]; lw[w] = w0p[w * (DEC_WAVE_LDS / 4) + DEC_D + 1];
            mx = fmaxf(mx, mw[w]);
        }
        float lsum_ = 0.f, o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; w++) {   // (waves in order: deterministic)
            const float wt = lw[w] > 0.f ? __expf(mw[w] - mx) : 0.f;
            lsum_ += lw[w] * wt;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(w0p + w * (DEC_WAVE_LDS / 4) + dg), hi = *reinterpret_cast<const f32x4*>(w0p + w * (DEC_WAVE_LDS / 4) + dg + 4);
#pragma unroll
            for (int j = 0; j < 4; j++) { o8[j] += lo[j] * wt; o8[4 + j] += hi[j] * wt; }
        }
        if (i0 + qi < p.q) {
            float* dst = p.part + ((((int64_t)b * p.H + h) * p.nunit + chunk) * 64 + i0 + qi) * (DEC_D + 2);
            if (inlaunch) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) dec_st_agent(dst + dg + j, o8[j], o8[j + 1]);
                if (dg == 0) dec_st_agent(dst + DEC_D, mx, lsum_);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j += 2) *reinterpret_cast<float2*>(dst + dg + j) = make_float2(o8[j], o8[j + 1]);
                if (dg == 0) { dst[DEC_D] = mx; dst[DEC_D + 1] = lsum_; }
            }
        }
    }
    if (!inlaunch) return;
    // ---- the chunk that finishes last for this (batch, head) merges the units (ticket counter, zero on entry and left zero): one launch less
    // per layer.  Same arithmetic as relattn_decode_merge2_kernel, two queries at a time (128 threads each).
    // Hand-off (per-XCD L2s are not coherent, L1s never refreshed): write-through 8-byte stores above, every wave drains them, one relaxed
    // agent-scope ticket per workgroup, and the merging workgroup reads the partials with L1-bypassing (sc1) loads.
    __shared__ unsigned ticket_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* tk = p.tickets + ((int64_t)b * p.H + h) * nqt + i0 / 16;
    if (threadIdx.x == 0) ticket_s = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket_s != gridDim.x / nqt - 1) return;
    if (threadIdx.x == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {   // thread = (query qi of the tile, 8 output columns): all units' pieces are requested at once (<= 16 chunks: klen <= 2048), weighted in
        // chunk order -- the arithmetic of relattn_decode_merge2_kernel, one round trip instead of one per row pair
        constexpr int NU = 16;
        const int qi = threadIdx.x & 15, dg = (threadIdx.x >> 4) * 8;
        const bool valid = i0 + qi < p.q;
        const float* src = p.part + ((((int64_t)b * p.H + h) * p.nunit) * 64 + (valid ? i0 + qi : i0)) * (DEC_D + 2);
        const int64_t cs = (int64_t)64 * (DEC_D + 2);
        float mlm[NU], mll[NU], ov[NU][8];
#pragma unroll
        for (int c = 0; c < NU; c++) {
            const float* u = src + (c < p.nunit ? c : p.nunit - 1) * cs;
            dec_ld_agent2(u + DEC_D, mlm[c], mll[c]);
#pragma unroll
            for (int j = 0; j < 8; j += 2) dec_ld_agent2(u + dg + j, ov[c][j], ov[c][j + 1]);
        }
        float mx = -1.0e30f;
#pragma unroll
        for (int c = 0; c < NU; c++) mx = fmaxf(mx, c < p.nunit ? mlm[c] : -1.0e30f);
        float l = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NU; c++) {
            const float lc = c < p.nunit ? mll[c] : 0.f;
            const float wt = lc > 0.f ? __expf(mlm[c] - mx) : 0.f;
            l += lc * wt;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] += ov[c][j] * wt;
        }
        if (valid) {
            uint4 w;
            w.x = f2bf_pk(l > 0.f ? o[0] / l : 0.f, l > 0.f ? o[1] / l : 0.f); w.y = f2bf_pk(l > 0.f ? o[2] / l : 0.f, l > 0.f ? o[3] / l : 0.f);
            w.z = f2bf_pk(l > 0.f ? o[4] / l : 0.f, l > 0.f ? o[5] / l : 0.f); w.w = f2bf_pk(l > 0.f ? o[6] / l : 0.f, l > 0.f ? o[7] / l : 0.f);
            *reinterpret_cast<uint4*>(p.out + ((int64_t)b * p.q + i0 + qi) * p.H * DEC_D + h * DEC_D + dg) = w;
        }
    }
}
// one 128-thread block per (query, head, batch): the units' (m, l) first (independent loads, weights through LDS), then the weighted sum
__global__ __launch_bounds__(128) void relattn_decode_merge2_kernel(const float* __restrict__ part, bf16_t* __restrict__ out, int q, int H, int nunit) {
    __shared__ float wgt[64];
    __shared__ float red[2];
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
    const float* src = part + ((((int64_t)b * H + h) * nunit) * 64 + i) * (DEC_D + 2);
    const int64_t cs = (int64_t)64 * (DEC_D + 2);
    if (d < 64) {   // first wave: nunit <= 64 units
        const float m = d < nunit ? src[d * cs + DEC_D] : -1.0e30f;
        const float l = d < nunit ? src[d * cs + DEC_D + 1] : 0.f;
        float mx = m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float w = l > 0.f ? __expf(m - mx) : 0.f;
        wgt[d] = w;
        float ls = l * w;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o, 64);
        if (d == 0) red[0] = ls;
    }
    __syncthreads();
    float o = 0.f;
#pragma unroll 8
    for (int c = 0; c < nunit; c++) o += src[c * cs + d] * wgt[c];   // (units in order: deterministic)
    const float l = red[0];
    out[((int64_t)b * q + i) * H * DEC_D + h * DEC_D + d] = f2bf(l > 0.f ? o / l : 0.f);
}
__global__ void ring_advance_kernel(int* state, int q, int cap) { int s = state[0] + q; state[0] = s >= cap ? s - cap : s; }

extern "C" int db1_ring_advance(int* state, int q, int cap, void* stream) {
    if (!state || q < 0 || cap <= 0) DB1_FAIL(DB1_ERR_BAD_SHAPE, "ring_advance");
    ring_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>(state, q, cap);
    DB1_CHECK_LAUNCH("ring_advance");
    return DB1_OK;
}
extern "C" int64_t db1_relattn_decode_ring_workspace_bytes(int B, int q, int klen, int H) {
    (void)q;
    return (int64_t)B * H * ((klen + 31) / 32) * 64 * (DEC_D + 2) * (int64_t)sizeof(float);
}
extern "C" int db1_relattn_decode_ring_fwd(const void* qkv_new, const void* u, const void* vb, void* kv_ring, const int* ring_state, int cap, const void* R,
                                           int nd, void* out, int B, int q, int mlen, int H, int D, int shift, float scale, void* ws, int64_t ws_bytes,
                                           void* tickets, void* stream) {
    const int klen = mlen + q;
    if (!db1_relattn_decode_supported(B, q, klen, H, D, DB1_BF16) || klen > 64 * 32)
        DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_decode_ring: needs bf16, d_head = 128, 1 <= q <= 64, klen <= 2048 (got q=%d klen=%d D=%d)", q, klen, D);
    if (cap < klen || nd < 1 || !R || !ring_state || !kv_ring || !u || !vb) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_decode_ring: ring capacity %d < klen %d, or null buffer", cap, klen);
    if (!db1_aligned16(qkv_new) || !db1_aligned16(kv_ring) || !db1_aligned16(R) || (out && !db1_aligned16(out)) || !db1_aligned16(u) || !db1_aligned16(vb))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_decode_ring: operands must be 16-byte aligned");
    DecodeRingArgs a;
    a.qkv = (const bf16_t*)qkv_new; a.u = (const bf16_t*)u; a.vb = (const bf16_t*)vb; a.ring = (bf16_t*)kv_ring; a.state = ring_state; a.R = (const bf16_t*)R;
    a.out = (bf16_t*)out; a.B = B; a.q = q; a.klen = klen; a.mlen = mlen; a.H = H; a.shift = shift; a.nd = nd; a.cap = cap; a.scale = scale;
    const bool split = true;   // (one workgroup per (query tile, chunk) for every q: the tile-per-wave form walked a chunk's four steps in sequence)
    const int nchunk = (klen + DECR_KC - 1) / DECR_KC, nqt = (q + 15) / 16;
    a.tickets = ((int64_t)B * H * nqt <= 2048 && out) ? (unsigned*)tickets : nullptr;   // (the ticket buffer of db1_linear_decode_tickets_bytes)
    a.nunit = nchunk;   // (the four waves of a workgroup are merged inside it)
    DB1_NEED_WS(ws, ws_bytes, db1_relattn_decode_ring_workspace_bytes(B, q, klen, H), "relattn_decode_ring");
    a.part = (float*)ws;
    hipStream_t st = (hipStream_t)stream;
    relattn_decode_ring_kernel<true><<<dim3((unsigned)(nchunk * nqt), (unsigned)H, (unsigned)B), 256, 4 * DEC_WAVE_LDS, st>>>(a);
    DB1_CHECK_LAUNCH("relattn_decode_ring");
    if (out && !(split && a.tickets)) {
        relattn_decode_merge2_kernel<<<dim3((unsigned)q, (unsigned)H, (unsigned)B), 128, 0, st>>>(a.part, a.out, q, H, a.nunit);
        DB1_CHECK_LAUNCH("relattn_decode_merge2");
    }
    return DB1_OK;
}

extern "C" int db1_relattn_decode_supported(int B, int q, int klen, int H, int D, int dt) {
    return (dt == DB1_BF16 && D == DEC_D && B > 0 && H > 0 && q >= 1 && q <= 64 && klen >= q && B <= 65535 && H <= 65535) ? 1 : 0;
}

extern "C" int64_t db1_relattn_decode_workspace_bytes(int B, int q, int klen, int H) {  // per (b, h, key chunk): 64 rows x (out, max, sum)
    (void)q;
    return (int64_t)B * H * ((klen + DEC_KC - 1) / DEC_KC) * 64 * (DEC_D + 2) * (int64_t)sizeof(float);
}

extern "C" int db1_relattn_decode_fwd(const void* qu, const void* qv, const void* k, const void* v, int64_t kv_row_stride,
                                      int64_t kv_batch_stride, const void* R, int nd, void* out, int B, int q, int klen, int mlen, int H,
                                      int D, int shift, float scale, void* ws, int64_t ws_bytes, void* stream) {
    if (!db1_relattn_decode_supported(B, q, klen, H, D, DB1_BF16))
        DB1_FAIL(DB1_ERR_UNSUPPORTED, "relattn_decode: needs bf16, d_head = 128, 1 <= q <= 64 (got q=%d klen=%d D=%d)", q, klen, D);
    if (mlen != klen - q) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_decode: klen (%d) must be mlen (%d) + q (%d)", klen, mlen, q);
    if (nd < 1 || !R) DB1_FAIL(DB1_ERR_BAD_SHAPE, "relattn_decode: R rows");
    if ((kv_row_stride % 8) || (kv_batch_stride % 8) || !db1_aligned16(qu) || !db1_aligned16(qv) || !db1_aligned16(k) || !db1_aligned16(v) ||
        !db1_aligned16(R) || !db1_aligned16(out))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "relattn_decode: operands must be 16-byte aligned, strides multiples of 8 elements");
    DecodeArgs a;
    a.qu = (const bf16_t*)qu; a.qv = (const bf16_t*)qv; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.R = (const bf16_t*)R;
    a.out = (bf16_t*)out; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride;
    a.B = B; a.q = q; a.klen = klen; a.mlen = mlen; a.H = H; a.shift = shift; a.nd = nd; a.scale = scale;
    a.nchunk = (klen + DEC_KC - 1) / DEC_KC;
    DB1_NEED_WS(ws, ws_bytes, db1_relattn_decode_workspace_bytes(B, q, klen, H), "relattn_decode");
    a.part = (float*)ws;
    hipStream_t st = (hipStream_t)stream;
    relattn_decode_kernel<<<dim3((unsigned)a.nchunk, (unsigned)H, (unsigned)B), 256, 4 * DEC_WAVE_LDS, st>>>(a);
    DB1_CHECK_LAUNCH("relattn_decode");
    relattn_decode_merge_kernel<<<dim3((unsigned)q, (unsigned)H, (unsigned)B), 128, 0, st>>>(a);
    DB1_CHECK_LAUNCH("relattn_decode_merge");
    return DB1_OK;
}
