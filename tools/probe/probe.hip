// Hardware-semantics probe for gfx950 (MI355X).  Not part of the product: it
// pins down the lane<->element maps the kernels in bdm_db1_amd/csrc rely on
// (MFMA A/B/C fragment layouts, ds_read_b64_tr_b16, global_load_lds,
// ds_bpermute) so that they can be written without trial-and-error on the box.
// Build: hipcc --offload-arch=gfx950 -O2 probe.hip -o probe ; run: ./probe > probe.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
#define LDSP(T, p) ((__attribute__((address_space(3))) T*)(p))

static __host__ __device__ inline unsigned short f2bf(float f) {
    union { float f; unsigned u; } x; x.f = f;
    unsigned r = x.u + 0x7fffu + ((x.u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}

// A: [M][K] row-major float (small ints), B: [K][N] row-major
__global__ void k_mfma16(const float* A, const float* B, float* C) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int t = 0; t < 8; t++) {
        int k = (l >> 4) * 8 + t;
        a[t] = (short)f2bf(A[(l & 15) * 32 + k]);
        b[t] = (short)f2bf(B[k * 16 + (l & 15)]);
    }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[l * 4 + r] = c[r];
}
__global__ void k_mfma32(const float* A, const float* B, float* C) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int t = 0; t < 8; t++) {
        int k = (l >> 5) * 8 + t;
        a[t] = (short)f2bf(A[(l & 31) * 16 + k]);
        b[t] = (short)f2bf(B[k * 32 + (l & 31)]);
    }
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[l * 16 + r] = c[r];
}
__global__ void k_mfma16f(const float* A, const float* B, float* C) {  // 16x16x4 f32
    int l = threadIdx.x;
    float a = A[(l & 15) * 4 + (l >> 4)];
    float b = B[(l >> 4) * 16 + (l & 15)];
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[l * 4 + r] = c[r];
}
__global__ void k_mfma32f(const float* A, const float* B, float* C) {  // 32x32x2 f32
    int l = threadIdx.x;
    float a = A[(l & 31) * 2 + (l >> 5)];
    float b = B[(l >> 5) * 32 + (l & 31)];
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[l * 16 + r] = c[r];
}

// tr16_b64: LDS element e holds value e. mode 0: lane address = lane*8 bytes.
// mode 1: address = group*(4*stride) + (t>>2)*stride + (t&3)*8 with stride = 80 bytes (row of 40 shorts)
__global__ void k_tr(int mode, int* out) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    int l = threadIdx.x;
    for (int i = l; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    int byteoff;
    if (mode == 0) byteoff = l * 8;
    else { int g = l >> 4, t = l & 15; byteoff = g * (4 * 80) + (t >> 2) * 80 + (t & 3) * 8; }
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(s4, (char*)lds + byteoff));
    for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}

// global_load_lds 16B: 2 waves; wave w writes to lds + w*1024 bytes (+512 extra for w==1 to see base handling)
__global__ void k_glds(const short* g, int* out) {
    __shared__ __attribute__((aligned(16))) short lds[2048];
    int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    for (int i = tid; i < 2048; i += 128) lds[i] = (short)-1;
    __syncthreads();
    // per-lane source: reversed chunk order so we can see which lane lands where
    const short* src = g + (w * 64 + (63 - l)) * 8;
    __builtin_amdgcn_global_load_lds(src, LDSP(void, (char*)lds + w * 2048), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 2048; i += 128) out[i] = lds[i];
}

__global__ void k_bperm(int* out) {
    int l = threadIdx.x;
    int v = 1000 + l;
    int src = (l * 7 + 3) & 63;
    out[l] = __builtin_amdgcn_ds_bpermute(src * 4, v);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)


int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s arch %s CUs %d clock %d kHz mem %zu\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.totalGlobalMem);
    float *dA, *dB, *dC; CK(hipMalloc(&dA, 4096 * 4)); CK(hipMalloc(&dB, 4096 * 4)); CK(hipMalloc(&dC, 4096 * 4));
    std::vector<float> A(4096), B(4096), C(4096);
    auto fill = [&](int na, int nb) { for (int i = 0; i < na; i++) A[i] = (float)((i * 7 + 3) % 11 - 5); for (int i = 0; i < nb; i++) B[i] = (float)((i * 5 + 1) % 13 - 6);
        CK(hipMemcpy(dA, A.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), nb * 4, hipMemcpyHostToDevice)); };
    // ---- 16x16x32 bf16
    {
        fill(16 * 32, 32 * 16);
        k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 64 * 4 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) { int col = l & 15, row = (l >> 4) * 4 + r; float e = 0; for (int k = 0; k < 32; k++) e += A[row * 32 + k] * B[k * 16 + col]; if (e != C[l * 4 + r]) bad++; }
        printf("mfma_16x16x32_bf16 hypothesis A[l&15][(l>>4)*8+t] B[(l>>4)*8+t][l&15] C col=l&15,row=(l>>4)*4+r : mismatches %d\n", bad);
    }
    {
        fill(32 * 16, 16 * 32);
        k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 64 * 16 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) { int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); float e = 0; for (int k = 0; k < 16; k++) e += A[row * 16 + k] * B[k * 32 + col]; if (e != C[l * 16 + r]) bad++; }
        printf("mfma_32x32x16_bf16 hypothesis A[l&31][(l>>5)*8+t] B[..][l&31] C col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5) : mismatches %d\n", bad);
    }
    {
        fill(16 * 4, 4 * 16);
        k_mfma16f<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 64 * 4 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) { int col = l & 15, row = (l >> 4) * 4 + r; float e = 0; for (int k = 0; k < 4; k++) e += A[row * 4 + k] * B[k * 16 + col]; if (e != C[l * 4 + r]) bad++; }
        printf("mfma_16x16x4_f32 hypothesis A[l&15][l>>4] B[l>>4][l&15] C col=l&15,row=(l>>4)*4+r : mismatches %d\n", bad);
    }
    {
        fill(32 * 2, 2 * 32);
        k_mfma32f<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 64 * 16 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) { int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); float e = 0; for (int k = 0; k < 2; k++) e += A[row * 2 + k] * B[k * 32 + col]; if (e != C[l * 16 + r]) bad++; }
        printf("mfma_32x32x2_f32 hypothesis A[l&31][l>>5] : mismatches %d\n", bad);
    }
    // ---- tr16
    int* dI; CK(hipMalloc(&dI, 8192 * 4)); std::vector<int> I(8192);
    for (int mode = 0; mode < 2; mode++) {
        k_tr<<<1, 64>>>(mode, dI); CK(hipDeviceSynchronize()); CK(hipMemcpy(I.data(), dI, 256 * 4, hipMemcpyDeviceToHost));
        printf("tr16_b64 mode %d (value = LDS short index):\n", mode);
        int bad = 0;
        for (int l = 0; l < 64; l++) {
            printf("  lane %2d: %5d %5d %5d %5d\n", l, I[l * 4], I[l * 4 + 1], I[l * 4 + 2], I[l * 4 + 3]);
            for (int j = 0; j < 4; j++) {
                int e = (mode == 0) ? ((l & 15) + j * 16 + (l >> 4) * 64) : ((l >> 4) * 160 + j * 40 + (l & 15));
                if (I[l * 4 + j] != e) bad++;
            }
        }
        printf("tr16_b64 mode %d hypothesis (lane gets column l&15 of its group's 4x16 block, elem j = row j): mismatches %d\n", mode, bad);
    }
    // ---- glds
    {
        short* dG; CK(hipMalloc(&dG, 4096 * 2)); std::vector<short> G(4096); for (int i = 0; i < 4096; i++) G[i] = (short)i; CK(hipMemcpy(dG, G.data(), 4096 * 2, hipMemcpyHostToDevice));
        k_glds<<<1, 128>>>(dG, dI); CK(hipDeviceSynchronize()); CK(hipMemcpy(I.data(), dI, 2048 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int w = 0; w < 2; w++) for (int l = 0; l < 64; l++) for (int t = 0; t < 8; t++) { int e = (w * 64 + (63 - l)) * 8 + t; if (I[w * 1024 + l * 8 + t] != e) bad++; }
        printf("global_load_lds x4 hypothesis (lds[base + lane*16B] <- per-lane src 16B): mismatches %d\n", bad);
        if (bad) { for (int i = 0; i < 2048; i += 8) { printf("  lds[%4d]:", i); for (int t = 0; t < 8; t++) printf(" %5d", I[i + t]); printf("\n"); } }
    }
    {
        k_bperm<<<1, 64>>>(dI); CK(hipDeviceSynchronize()); CK(hipMemcpy(I.data(), dI, 64 * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int l = 0; l < 64; l++) if (I[l] != 1000 + ((l * 7 + 3) & 63)) bad++;
        printf("ds_bpermute hypothesis out[l]=v[addr/4] : mismatches %d\n", bad);
    }
    // ---- bandwidth + clock sanity: float4 copy 1 GiB
    {
        size_t n = (size_t)1 << 28; float *x, *y; CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMemset(x, 1, n * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int it = 0; it < 3; it++) { CK(hipEventRecord(e0)); CK(hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("d2d copy 1GiB: %.3f ms -> %.1f GB/s (r+w)\n", ms, 2.0 * n * 4 / ms / 1e6); }
    }
    printf("probe done\n");
    return 0;
}
