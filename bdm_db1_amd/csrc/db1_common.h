// Shared device/host helpers for libdb1_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include "../../include/db1_hip.h"
#include "../../include/db1_hip_test.h"

// The only process-level state of the library: "this kernel's dynamic-LDS attribute has been set on this device" -- an immutable
// per-device cache, set once under std::call_once (kernel attributes are per device: a process that drives several GPUs sets
// them on each).
struct Db1PerDeviceOnce {
    std::once_flag flags[64];
    template <class F> void run(F&& fn) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::call_once(flags[dev & 63], fn);
    }
};

typedef unsigned short bf16_t;  // raw bfloat16 bits

void db1_set_error(const char* fmt, ...);

// A/B knobs for measurements and parity tests (include/db1_hip_test.h: db1_test_set_knob): THREAD-LOCAL, unset by default, never read
// from the environment -- the library has no process-level mutable state.  db1_knob(id, dflt) = the calling thread's value.
enum Db1Knob { DB1_KNOB_GEMM_TILE = 0, DB1_KNOB_GEMM_SPLITK, DB1_KNOB_PP32_STAGES, DB1_KNOB_LINEAR_DECODE_SPLITK, DB1_KNOB_W4,
               DB1_KNOB_FLASH_FWD2, DB1_KNOB_FLASH_KV3, DB1_KNOB_CONV_WGRAD_KS, DB1_KNOB_GEGLU_EPI, DB1_KNOB_GEMM_HALFWAVE, DB1_KNOB_W4N, DB1_KNOB_CONV_PATCH, DB1_KNOB_TRI_SPLIT, DB1_KNOB_COUNT };
int db1_knob(int id, int dflt);

#define DB1_FAIL(code, ...)           \
    do {                              \
        db1_set_error(__VA_ARGS__);   \
        return (code);                \
    } while (0)

#define DB1_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) DB1_FAIL(DB1_ERR_HIP, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// Scratch memory comes from the CALLER (include/db1_hip.h, "Conventions"): an entry point that needs scratch takes (ws, ws_bytes) and has a
// db1_<op>_workspace_bytes(...) query; the library never calls hipMalloc / hipFree / hipDeviceSynchronize.
#define DB1_NEED_WS(ws, ws_bytes, need, name)                                                                      \
    do {                                                                                                           \
        if ((need) > 0 && (!(ws) || (int64_t)(ws_bytes) < (int64_t)(need) || (((uintptr_t)(ws)) & 15)))             \
            DB1_FAIL(DB1_ERR_WORKSPACE_TOO_SMALL, "%s: needs %lld bytes of 16-byte aligned workspace, got %lld",  \
                     name, (long long)(need), (long long)((ws) ? (ws_bytes) : 0));                               \
    } while (0)

static inline int db1_elt_size(int dt) { return dt == DB1_F32 ? 4 : 2; }
static inline bool db1_dt_ok(int dt) { return dt == DB1_F32 || dt == DB1_BF16; }
static inline bool db1_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// fp32 -> bf16, round to nearest even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 (the integer sequence it replaces was ~10 VALU
// instructions and a divergent NaN branch per element: the bf16 activation kernels were VALU-bound on it)
typedef __attribute__((ext_vector_type(2))) float db1_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 db1_bf16x2_t;
__device__ __forceinline__ unsigned f2bf_pk(float lo, float hi) {  // {bf16(lo), bf16(hi)} in one dword
    db1_f32x2_t v = {lo, hi};
    db1_bf16x2_t r = __builtin_convertvector(v, db1_bf16x2_t);
    return *reinterpret_cast<unsigned*>(&r);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(f2bf_pk(f, 0.f) & 0xffffu); }

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// 16-byte vector access: VEC = 4 floats or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float v[4];
    __device__ __forceinline__ void load(const float* p) { float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const bf16_t* p) {
        uint4 t = *reinterpret_cast<const uint4*>(p);
        unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; i++) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    __device__ __forceinline__ void store(bf16_t* p) const {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = f2bf_pk(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}
// block-wide sum for blockDim.x == 256 (4 waves); sm must hold >= 4 floats
__device__ __forceinline__ float block_sum256(float x, float* sm) {
    x = wave_sum(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = x;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ float block_max256(float x, float* sm) {
    x = wave_max(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = x;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * __expf(-0.5f * x * x) * 0.39894228040143267794f;
}

// bf16 storage path: Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and e = exp(-x^2 / 2) from ONE v_exp and one v_rcp (Abramowitz-Stegun
// 7.1.26, |error| <= 1.5e-7 absolute: three orders below the bf16 rounding of the result).  libm's erff costs ~3x the VALU work
// and made the activation kernels VALU-bound (act_bwd: ~100 us of VALU for 124 us).  The fp32 path keeps erff (parity gate).
__device__ __forceinline__ void gelu_phi_fast(float x, float& phi, float& e) {
    const float t = fabsf(x) * 0.70710678118654752440f;
    e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2/2) = 2^(-x^2/2 * log2 e)
    const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, t, 1.0f));
    float poly = fmaf(1.061405429f, k, -1.453152027f);
    poly = fmaf(poly, k, 1.421413741f);
    poly = fmaf(poly, k, -0.284496736f);
    poly = fmaf(poly, k, 0.254829592f);
    const float erf_abs = fmaf(-poly * k, e, 1.0f);
    phi = 0.5f + copysignf(0.5f * erf_abs, x);
}
template <typename T> __device__ __forceinline__ float gelu_fwd_t(float x) {
    if (sizeof(T) == 4) return gelu_erf(x);
    float phi, e;
    gelu_phi_fast(x, phi, e);
    return x * phi;
}
// gelu(x) and d gelu / dx together (they share erf and the exponential)
template <typename T> __device__ __forceinline__ void gelu_both_t(float x, float& y, float& dy) {
    if (sizeof(T) == 4) { y = gelu_erf(x); dy = gelu_erf_grad(x); return; }
    float phi, e;
    gelu_phi_fast(x, phi, e);
    y = x * phi;
    dy = fmaf(x * 0.39894228040143267794f, e, phi);
}

// ---- dropout (transformer_xl.py:229,262-269,409,545,575): counter-based, no mask tensors.  The keep decision of element e of a tensor is a
// pure function of (seed, step, site, e): Philox4x32-10 on the counter (e / 8 [64 bit], site, step) under the key (seed), whose 128
// output bits are eight 16-bit uniforms u; element e keeps its value (scaled by 65536 / (65536 - thr)) iff u[e % 8] >= thr, with
// thr = round(p * 65536) -- so E[output] = input exactly.  The backward regenerates the same decisions.  site = which tensor of the
// step (layer * 4 + {0: attention output, 1: feed-forward output}, 0xE0000000: embeddings, 0xE0000001: position table), step = the
// engine's micro-step counter.  oracle/db1_oracle.py holds the same function in NumPy, which is how parity at p > 0 is checked.
struct Db1Drop {
    unsigned thr;   // 0 = dropout off
    float scale;
    unsigned k0, k1, site, step;
    const unsigned* step_dev;   // nullable: the step used is step + *step_dev (a device counter: the value a captured hipGraph reads at REPLAY time)
    // > 0: the tensor holds SEVERAL micro-steps' rows, rows_per_step each (a whole accumulation window run through one backward): row r belongs to
    // step + r / rows_per_step and its elements are counted from the first row of that block -- the decisions its own forward drew
    long long rows_per_step;
};
static inline Db1Drop db1_drop_make(float p, uint64_t seed, uint32_t site, uint32_t step, const uint32_t* step_dev = nullptr) {
    Db1Drop d;
    long t = p > 0.f ? lrintf(p * 65536.f) : 0;
    d.thr = (unsigned)(t < 0 ? 0 : (t > 65535 ? 65535 : t));
    d.scale = 65536.f / (float)(65536 - (long)d.thr);
    d.k0 = (unsigned)(seed & 0xffffffffu); d.k1 = (unsigned)(seed >> 32); d.site = site; d.step = step;
    d.step_dev = step_dev;
    d.rows_per_step = 0;
    return d;
}
// the element index of (row, column 0) inside the row's micro-step block; advances dr.step to that micro-step (per-row copy of the descriptor)
__device__ __forceinline__ int64_t db1_drop_row_base(Db1Drop& dr, int64_t row, int d) {
    if (dr.rows_per_step > 0) {
        const unsigned m = (unsigned)row / (unsigned)dr.rows_per_step;
        dr.step += m;
        row -= (int64_t)m * dr.rows_per_step;
    }
    return row * d;
}
__device__ __forceinline__ void db1_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned o[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// multiplies the V consecutive elements starting at element index e0 (a multiple of V; V = 4 or 8) by their keep-scales
template <int V>
__device__ __forceinline__ void db1_drop_apply(const Db1Drop& dr, int64_t e0, float* v) {
    unsigned o[4];
    const unsigned long long blk = (unsigned long long)e0 >> 3;
    const unsigned step = dr.step_dev ? dr.step + *dr.step_dev : dr.step;   // (uniform address: a scalar load)
    db1_philox4x32_10((unsigned)blk, (unsigned)(blk >> 32), dr.site, step, dr.k0, dr.k1, o);
    const int w0 = V == 8 ? 0 : (int)((e0 >> 2) & 1) * 2;
#pragma unroll
    for (int j = 0; j < V; j += 2) {
        const unsigned w = o[w0 + (j >> 1)];
        v[j] *= (w & 0xffffu) >= dr.thr ? dr.scale : 0.f;
        v[j + 1] *= (w >> 16) >= dr.thr ? dr.scale : 0.f;
    }
}

// dtype dispatch helpers (host)
#define DB1_DISPATCH_DT(dt, T, ...)                          \
    do {                                                     \
        if ((dt) == DB1_F32) { using T = float; __VA_ARGS__; } \
        else { using T = bf16_t; __VA_ARGS__; }               \
    } while (0)
