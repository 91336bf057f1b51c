// Implicit-GEMM 3x3 convolutions of the image-patch embedder (vision_embedding.py:44-63: 64 -> 64 channels on 16x16 patches, zero
// padding at the PATCH border), channels-last bf16.  The column matrix of im2col (295 KB per patch and conv) is never written:
// the LDS-DMA that stages a GEMM operand tile simply takes a different SOURCE address per lane -- the pixel shifted by the tap --
// and lanes whose shifted pixel falls outside the patch read a 128-byte page of zeros instead (LDS-DMA cannot write constants,
// but it can read them).  Everything after the staging is the 128-row K-major / M-major tile machinery of gemm_tile.h.
//   conv_implicit_kernel  (forward, and data gradient with sign = -1 and the transposed weight operand):
//       Y[pix, o] = sum_{tap, c} X[pix + sign * s(tap), c] * Wop[o, tap*64 + c] (+ bias)
//       one workgroup = one patch (256 pixels) x 64 outputs, 4 waves x (64 pixels x 64 outputs), 9 k-steps = the 9 taps.
//   conv_wgrad_implicit_kernel:  gp[o, tap*64 + c] += sum_pix dY[pix, o] * X[pix + s(tap), c]
//       a TN product with M = 64, N = 576, K = all pixels; the B operand tile (64 pixels x 128 columns = 2 taps) is gathered;
//       the pixel range is split over blockIdx.z; the partial sums go through the caller's workspace and are added in a fixed order
//       (without a workspace: fp32 atomics, as the explicit path did).
#include "gemm_tile.h"

#define CI_C 64
#define CI_P 16
#define CI_HW 256

// 256 bytes of zeros in device memory (module data, never written): the source of the LDS-DMA lanes that fall outside the patch
__device__ __attribute__((aligned(256))) bf16_t ci_zero_page[128] = {};

struct ConvArgs {
    const bf16_t* x;      // gathered activations [n_patches * 256, 64]
    const bf16_t* w;      // fwd / dgrad: weight operand [64, 576] (K-major);  wgrad: dY [n_patches * 256, 64]
    void* y;              // fwd / dgrad: [n_patches * 256, 64] bf16;  wgrad: gp [64, 576] float32 (accumulated)
    const void* bias;
    int64_t n_patches;
    int sign, ksplit;
    float* part;          // wgrad: per-pixel-range partial sums [ksplit][64][576] (deterministic mode), or null (fp32 atomics onto y)
    const bf16_t* res;    // fwd: optional residual [n_patches * 256, 64] added in the epilogue (y = conv + bias + res)
    float* gbias;         // wgrad: optional bias gradient [64] (accumulated): column sums of dY from the A fragments the kernel holds anyway
    float* part_bias;     // ... its partial sums [ksplit][64] in the deterministic mode
};

// source of the 16-byte chunk `c` (8 channels) of pixel `pix` shifted by tap `tap`, or the zero page
__device__ __forceinline__ const bf16_t* ci_src(const ConvArgs& p, int64_t pix, int tap, int sign, int c) {
    const int yx = (int)(pix & (CI_HW - 1));
    const int yy = (yx >> 4) + sign * (tap / 3 - 1), xx = (yx & 15) + sign * (tap % 3 - 1);
    const bool ok = yy >= 0 && yy < CI_P && xx >= 0 && xx < CI_P;
    return ok ? p.x + ((pix - yx) + yy * CI_P + xx) * CI_C + c * 8 : ci_zero_page + c * 8;
}

// ---------------------------------------------------------------------------------------------------------------------- fwd / dgrad
#define CI_STAGE_BYTES (2 * TILE_BYTES + TILE_BYTES / 2)   // A: 256 pixels x 64 ch (two 128-row images) | B: 64 outputs x 64 k
template <typename TBIAS>
__global__ __launch_bounds__(256, 2) void conv_implicit_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t pix0 = (int64_t)blockIdx.x * CI_HW;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int tap, char* s) {
        // A: 32 pieces of 8 rows (K-major image, chunk swizzle c ^ (r & 7)); wave w takes pieces 8w .. 8w+7
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int q = wave * 8 + it;                 // piece 0..31 -> rows 8q .. 8q+7 of the 256
            const int r = q * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            __builtin_amdgcn_global_load_lds(ci_src(p, pix0 + r, tap, p.sign, c), LDS_PTR(void, s + (q >> 4) * TILE_BYTES + (q & 15) * 1024), 16, 0, 0);
        }
        // B: weight rows o = 0..63, k = tap*64 .. +63: 8 pieces, 2 per wave
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int q = wave * 2 + it;
            const int r = q * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            __builtin_amdgcn_global_load_lds(p.w + (int64_t)r * (9 * CI_C) + tap * CI_C + c * 8, LDS_PTR(void, s + 2 * TILE_BYTES + q * 1024), 16, 0, 0);
        }
    };
    stage(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int tap = 0; tap < 9; tap++) {
        char* sa = smem + cur * CI_STAGE_BYTES;
        char* sb = sa + 2 * TILE_BYTES;
        if (tap + 1 < 9) stage(tap + 1, smem + (cur ^ 1) * CI_STAGE_BYTES);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = wave * 64 + i * 16;
                af[i] = load_frag<true>(sa + (row >> 7) * TILE_BYTES, row & 127, ks, lane);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) bfr[j] = load_frag<true>(sb, j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // swapped: D[n][m]
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    bf16_t* Y = (bf16_t*)p.y + pix0 * CI_C;
    if (p.res) {   // (the residual sum of the block, vision_embedding.py:84: one pass over two [pixels, 64] tensors less than a separate add)
        const bf16_t* Rr = p.res + pix0 * CI_C;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int m = wave * 64 + i * 16 + (lane & 15), n = j * 16 + (lane >> 4) * 4;
                const uint2 rv = *reinterpret_cast<const uint2*>(Rr + (int64_t)m * CI_C + n);
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (p.bias) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] += ldf((const TBIAS*)p.bias + n + r);
                }
                v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
                v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
                uint2 o;
                o.x = f2bf_pk(v[0], v[1]); o.y = f2bf_pk(v[2], v[3]);
                *reinterpret_cast<uint2*>(Y + (int64_t)m * CI_C + n) = o;
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            store_frag<bf16_t, TBIAS>(acc[i][j], Y, CI_C, wave * 64 + i * 16 + (lane & 15), j * 16 + (lane >> 4) * 4, 1.f, 0.f, p.bias);
}

// ---------------------------------------------------------------------------------------------------------------------- fwd / dgrad, patch-resident
// Round 6.  The kernel above walks the nine taps as nine k-steps of a tile GEMM: per patch it stages 9 x (32 KB of tap-shifted pixels + 8 KB
// of weights) from L2 and pays nine request -> vmcnt(0) -> barrier round trips -- 19 us per patch for 2.2 us of MFMAs (RL step, 60 160
// patches: 2.26 ms per convolution, 1.7 TB/s of its 3.9 GB, 0.2 of the MFMA peak; profiles/r06c_rl_step_table.txt).  A patch is its own
// zero-padded image (vision_embedding.py:44-63: the padding is at the PATCH border), so nothing outside its 32 KB is ever needed:
//   * ONE workgroup per CU, persistent over patches; the whole weight operand [64][576] (72 KB) is staged once per workgroup;
//   * a patch's [256 pixels][64 channels] image is staged ONCE (32 KB, two buffers: patch k + 1 lands while patch k is computed) and the nine
//     taps are nine different ROW addresses into it: the A fragment of (pixel tile i, tap) is the same ds_read_b128 at pixel + shift, and a
//     lane whose shifted pixel is outside the patch reads a row of zeros that sits in front of each buffer (one lane-constant address
//     per (i, tap), the second k-step is address ^ 64);
//   * one barrier per patch; the LDS-DMA is issued from inline asm (hipcc drains vmcnt(0) before the first LDS read it cannot disambiguate
//     from a pending builtin DMA, which is what serialised the kernel above).
// Bytes per patch: 32 KB read + 32 KB written (the kernel is HBM-bound at ~0.8 ms per convolution at 60 160 patches), 288 MFMAs and 144
// fragment reads per wave.  Same arithmetic and output as conv_implicit_kernel (fp32 accumulation over the same 576 products per output;
// the order of the k-steps is the same: tap-major).
#define CP_W_BYTES (9 * 8192)                 // nine tap slabs [64 outputs][64 k] (K-major rows of 128 B, chunk swizzle c ^ (row & 7))
#define CP_X_BYTES (CI_HW * 128)              // one patch [256 pixels][128 B]
#define CP_X_STRIDE (CP_X_BYTES + 256)        // a 256-byte row of zeros in front of each patch buffer
#define CP_LDS (CP_W_BYTES + 2 * CP_X_STRIDE)
__device__ __forceinline__ void cp_glds16(const void* src, unsigned dst_lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst_lds) : "memory");
}
// (Measured and dropped, same box, 60 160 patches: the weight fragments in REGISTERS for the workgroup's whole walk instead of in LDS -- half the
// fragment reads per MFMA -- 1322 -> 1580 us: hipcc parks the 288 registers in AGPRs and copies each one back before its MFMA, 235 moves per
// patch; profiles/r06e_conv_patch_ab.txt.  What bounds the kernel is the one patch in flight per CU: 32 KB at a CU's share of HBM is ~1.7 us
// plus the request latency, against 2.3 us of MFMAs -- 3.0 TB/s of the 5 a copy reaches; a third image buffer does not fit beside the weights.)
template <typename TBIAS>
__global__ __launch_bounds__(256, 1) void conv_patch_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xm = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
    const unsigned xbase0 = lds0 + CP_W_BYTES + 256;          // buffer b at xbase0 + b * CP_X_STRIDE, its zero row 256 bytes below
    if (tid < 128) {                                          // the two zero rows
        *reinterpret_cast<unsigned*>(smem + CP_W_BYTES + (tid >> 6) * CP_X_STRIDE + (tid & 63) * 4) = 0u;
    }
    // weights, once: 72 pieces of 8 rows; wave w takes pieces 18 w .. 18 w + 17
#pragma unroll
    for (int it = 0; it < 18; it++) {
        const int q = wave * 18 + it, tap = q >> 3;
        const int r = (q & 7) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        cp_glds16(p.w + (int64_t)r * (9 * CI_C) + tap * CI_C + c * 8, lds0 + q * 1024);
    }
    auto stage = [&](int64_t patch, int buf) {                // 32 pieces of 8 pixels; wave w takes pieces 8 w .. 8 w + 7
        const bf16_t* xp = p.x + patch * (CI_HW * CI_C);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int q = wave * 8 + it;
            const int r = q * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            cp_glds16(xp + r * CI_C + c * 8, xbase0 + buf * CP_X_STRIDE + q * 1024);
        }
    };
    // lane-constant fragment addresses, relative to a patch buffer: (pixel tile i, tap) -> the shifted pixel's row, or the zero row below the buffer
    int aoff[4][9];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int y = wave * 4 + i + p.sign * (tap / 3 - 1), x = xm + p.sign * (tap % 3 - 1);
            const bool ok = y >= 0 && y < CI_P && x >= 0 && x < CI_P;
            const int ps = y * CI_P + x;
            aoff[i][tap] = ok ? ps * 128 + ((g ^ (ps & 7)) << 4) : -256 + (g << 4);
        }
    unsigned boff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = j * 16 + xm;
        boff[j] = lds0 + row * 128 + ((g ^ (row & 7)) << 4);
    }
    typedef __attribute__((address_space(3))) const bf16x8_t* lds_frag_ptr;
    // this lane's 16 bias values (output n = 16 j + 4 g + r), once
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) bv[j][r] = p.bias ? ldf((const TBIAS*)p.bias + j * 16 + g * 4 + r) : 0.f;
    const int64_t first = blockIdx.x, stride = gridDim.x;
    if (first < p.n_patches) stage(first, 0);
    int buf = 0;
    bool first_iter = true;
    for (int64_t patch = first; patch < p.n_patches; patch += stride) {
        // this patch's image has landed (the first time: and the weights).  The 16 output stores of the previous patch were issued AFTER its
        // requests and may stay in flight (vmcnt retires in issue order): waiting for their acknowledgements here cost ~2 us per patch.
        if (first_iter) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        first_iter = false;
        const int64_t pix0 = patch * CI_HW;
        // the residual rows of this patch (vision_embedding.py:84), requested BEFORE the next patch's image so that they are the older requests
        uint2 rv[4][4];
        if (p.res) {
            const bf16_t* Rr = p.res + pix0 * CI_C;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) rv[i][j] = *reinterpret_cast<const uint2*>(Rr + (int64_t)(wave * 64 + i * 16 + xm) * CI_C + j * 16 + g * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (patch + stride < p.n_patches) stage(patch + stride, buf ^ 1);      // (that buffer was last read before the barrier)
        else {   // keep the count of the wait above: eight requests per wave and iteration (the last patch re-reads itself into the idle buffer)
            stage(patch, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned xb = xbase0 + buf * CP_X_STRIDE;
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                bf16x8_t af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; i++) af[i] = *(lds_frag_ptr)(size_t)((xb + (unsigned)aoff[i][tap]) ^ (ks << 6));
#pragma unroll
                for (int j = 0; j < 4; j++) bfr[j] = *(lds_frag_ptr)(size_t)((boff[j] + tap * 8192) ^ (ks << 6));
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // swapped: D[n][m]
            }
        }
        bf16_t* Y = (bf16_t*)p.y + pix0 * CI_C;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v[4] = {acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1], acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3]};
                if (p.res) {
                    v[0] += __uint_as_float(rv[i][j].x << 16); v[1] += __uint_as_float(rv[i][j].x & 0xffff0000u);
                    v[2] += __uint_as_float(rv[i][j].y << 16); v[3] += __uint_as_float(rv[i][j].y & 0xffff0000u);
                }
                uint2 o;
                o.x = f2bf_pk(v[0], v[1]); o.y = f2bf_pk(v[2], v[3]);
                *reinterpret_cast<uint2*>(Y + (int64_t)(wave * 64 + i * 16 + xm) * CI_C + j * 16 + g * 4) = o;
            }
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------- wgrad
// gp[o, n] += sum_pix dY[pix, o] * G[pix, n],  G[pix, tap*64 + c] = X[pix + s(tap), c].  Tile: M = 64 outputs (one M-major sub-tile,
// half used), N = 128 columns = taps 2 tn, 2 tn + 1 (the fifth tile holds tap 8 only), k-tiles of 64 pixels (a quarter patch).
// 4 waves as 2 x 2 over (64 outputs) x (128 columns): wave tile 32 x 64.
__global__ __launch_bounds__(256, 2) void conv_wgrad_implicit_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tn = blockIdx.x;                           // column tile: taps 2 tn (and 2 tn + 1 if < 9)
    const int64_t nk_total = p.n_patches * (CI_HW / TBK);  // k-tiles of 64 pixels
    const int64_t per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int64_t kt0 = (int64_t)blockIdx.z * per;
    const int64_t kt1 = kt0 + per < nk_total ? kt0 + per : nk_total;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of dY over the pixels: the waves (tn = 0, wn = 0) that hold the dY fragments multiply them by a ones
    // operand as well (2 of 18 MFMAs per k-step): the separate pass over dY (2 GB at 15.4 M pixels, 0.64 ms) is gone
    const bool do_bias = p.gbias != nullptr && tn == 0 && wn == 0;
    f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const bf16x8_t ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16_t* dY = p.w;
    auto stage = [&](int64_t kt, char* s) {
        const int64_t pixk = kt * TBK;
        // A = dY^T, M-major image [64 k][256 B]: only the first 64 of the 128 "rows" exist (chunks 0..7); piece q = k-rows 4q..4q+3
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int q = wave * 4 + it;
            const int kr = q * 4 + (lane >> 4);
            const int f = ((kr & 3) << 1) | (kr & 8);
            int c = (lane & 15) ^ f;
            if (c > 7) c = 7;                            // columns 64..127 of the tile are never stored: any valid address will do
            __builtin_amdgcn_global_load_lds(dY + (pixk + kr) * CI_C + c * 8, LDS_PTR(void, s + q * 1024), 16, 0, 0);
        }
        // B = gathered activations, M-major image [64 k = pixels][128 n]: chunk c (8 columns) -> tap 2 tn + c / 8, channels (c % 8) * 8 ..
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int q = wave * 4 + it;
            const int kr = q * 4 + (lane >> 4);
            const int f = ((kr & 3) << 1) | (kr & 8);
            const int c = (lane & 15) ^ f;
            int tap = 2 * tn + (c >> 3);
            if (tap > 8) tap = 8;                        // second half of the last tile: masked in the epilogue
            __builtin_amdgcn_global_load_lds(ci_src(p, pixk + kr, tap, 1, c & 7), LDS_PTR(void, s + TILE_BYTES + q * 1024), 16, 0, 0);
        }
    };
    if (kt0 < kt1) stage(kt0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int64_t kt = kt0; kt < kt1; kt++) {
        char* sa = smem + cur * 2 * TILE_BYTES;
        char* sb = sa + TILE_BYTES;
        if (kt + 1 < kt1) stage(kt + 1, smem + (cur ^ 1) * 2 * TILE_BYTES);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8_t af[2], bfr[4];
#pragma unroll
            for (int i = 0; i < 2; i++) af[i] = load_frag<false>(sa, wm * 32 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; j++) bfr[j] = load_frag<false>(sb, wn * 64 + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // swapped: D[n][m]
            if (do_bias) {   // (wave-uniform)
#pragma unroll
                for (int i = 0; i < 2; i++) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accb[i], 0, 0, 0);   // D[*][m] = sum_k dY[k][m]
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    if (do_bias && (lane >> 4) == 0) {   // every n row of accb holds the same sums: lanes g = 0 write output m = wm * 32 + i * 16 + (lane & 15)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int m = wm * 32 + i * 16 + (lane & 15);
            p.part_bias[(int64_t)blockIdx.z * CI_C + m] = accb[i][0];
        }
    }
    float* G = p.part + (int64_t)blockIdx.z * (CI_C * 9 * CI_C);   // this pixel range's partial sums (conv_wgrad_reduce_kernel adds the ranges in order)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int m = wm * 32 + i * 16 + (lane & 15);
            const int n = tn * 128 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= 9 * CI_C) continue;
            *reinterpret_cast<float4*>(G + (int64_t)m * (9 * CI_C) + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
}
// ---------------------------------------------------------------------------------------------------------------------- wgrad, patch-resident
// Round 6, the weight gradient in the same form: gp[o, tap*64 + c] += sum_pix dY[pix, o] * X[pix + s(tap), c] with k = the PIXEL index, so both
// operands are read "against the grain" (ds_read_b64_tr_b16).  One persistent workgroup per CU keeps ALL of gp -- 64 x 576 fp32 = 144
// accumulators per lane -- in registers over every patch it walks (wave w owns input channels 16 w .. 16 w + 15 of all nine taps and all 64
// outputs: 36 MFMA tiles), and writes one partial slab at the end (added in workgroup order by conv_wgrad_reduce_kernel: deterministic).
//   * X is staged into an 18 x 18 image whose border is zero (written once; the LDS-DMA only ever writes the 16 x 16 interior), so a tap is a
//     constant address offset and no lane ever tests a bound; dY [256 pixels][64 outputs] as it is.  Two buffers each: patch k + 1 lands while
//     patch k is contracted.  32-byte column blocks are XOR-swizzled with bit 1 of the pixel slot (source side, in the DMA's lane -> chunk
//     map), which makes the four pixel rows of a 16-lane transpose read hit eight distinct bank groups, shifted or not.
//   * per 32-pixel k-step a wave reads 4 dY fragments and 9 X fragments (26 transpose reads) for 36 + 1 MFMAs (the + 1: its share of the
//     bias gradient, a ones operand against its dY fragment).
// Bytes per patch: 64 KB read, nothing written.  The tile form above re-reads X nine times through L2 and pays a vmcnt(0) + barrier per
// 64 pixels (2.7 ms per convolution at 60 160 patches).
__device__ __forceinline__ bf16x8_t ot_select(const bf16x8_t (&yf)[4], int w) {   // (w is wave-uniform: scalar branches, no register indexing)
    return w == 0 ? yf[0] : (w == 1 ? yf[1] : (w == 2 ? yf[2] : yf[3]));
}
#define WP_XB (18 * 18 * 128)                 // padded patch image: 41 472 B
#define WP_DB (CI_HW * 128)
#define WP_LDS (2 * WP_XB + 2 * WP_DB)
__global__ __launch_bounds__(256, 1) void conv_wgrad_patch_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);
    const unsigned xb0 = lds0, db0 = lds0 + 2 * WP_XB;
    for (int o = tid * 16; o < 2 * WP_XB; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0u, 0u, 0u, 0u);   // the zero borders (and interiors, once)
    const bf16_t* dY = p.w;
    auto stage = [&](int64_t patch, int buf) {
        const bf16_t* xp = p.x + patch * (CI_HW * CI_C);
        const bf16_t* yp = dY + patch * (CI_HW * CI_C);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int q = wave * 8 + it;                         // piece: 8 pixels of patch row q >> 1
            const int y = q >> 1, x = (q & 1) * 8 + (lane >> 3);
            const int slot = (y + 1) * 18 + 1 + x;               // pixel slot in the padded image
            const int cx = (lane & 7) ^ (((slot >> 1) & 1) << 1);
            cp_glds16(xp + (y * CI_P + x) * CI_C + cx * 8, xb0 + buf * WP_XB + ((y + 1) * 18 + 1 + (q & 1) * 8) * 128);
            const int r = q * 8 + (lane >> 3);
            const int cy = (lane & 7) ^ (((r >> 1) & 1) << 1);
            cp_glds16(yp + r * CI_C + cy * 8, db0 + buf * WP_DB + q * 1024);
        }
    };
    // lane-constant transpose-read addresses (k-step 0, buffer 0): a 16-lane group passes (pixel row kb + i16 / 4, 4 elements at column block + (i16 % 4) * 4)
    const int prow = g * 8 + (i16 >> 2);                          // + 4 h: the lane's pixel inside a 32-pixel k-step
    unsigned xoff[2][9], yoff[4][2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int pm = prow + 4 * h, py = pm >> 4, px = pm & 15;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int slot = (py + tap / 3) * 18 + px + tap % 3;  // (y + 1 + dy) * 18 + (x + 1 + dx), dy = tap / 3 - 1, dx = tap % 3 - 1
            xoff[h][tap] = xb0 + slot * 128 + ((wave ^ ((slot >> 1) & 1)) << 5) + (i16 & 3) * 8;
        }
#pragma unroll
        for (int ot = 0; ot < 4; ot++) yoff[ot][h] = db0 + pm * 128 + ((ot ^ ((pm >> 1) & 1)) << 5) + (i16 & 3) * 8;
    }
    f32x4 acc[9][4], accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int ot = 0; ot < 4; ot++) acc[t][ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    auto trfrag = [&](unsigned a0, unsigned a1) __attribute__((always_inline)) {
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(size_t)a0);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(size_t)a1);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    const int64_t first = blockIdx.x, stride = gridDim.x;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the zeros are in place before the first image lands on top of them
    if (first < p.n_patches) stage(first, 0);
    int buf = 0;
    for (int64_t patch = first; patch < p.n_patches; patch += stride) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // this patch's two images have landed; everybody is done with the other pair
        if (patch + stride < p.n_patches) stage(patch + stride, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned xo = buf * WP_XB, yo = buf * WP_DB;
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            bf16x8_t yf[4];
#pragma unroll
            for (int ot = 0; ot < 4; ot++) yf[ot] = trfrag(yoff[ot][0] + yo + ks * 4096, yoff[ot][1] + yo + ks * 4096);
            accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ot_select(yf, wave), accb, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const bf16x8_t xf = trfrag(xoff[0][t] + xo + ks * (36 * 128), xoff[1][t] + xo + ks * (36 * 128));
#pragma unroll
                for (int ot = 0; ot < 4; ot++) acc[t][ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf[ot], acc[t][ot], 0, 0, 0);   // D[c][o]
            }
        }
        buf ^= 1;
    }
    // this workgroup's partial slab: lane (i16 = output inside its tile, g) holds input channels 16 wave + 4 g .. + 3 of tap t
    float* G = p.part + (int64_t)blockIdx.x * (CI_C * 9 * CI_C);
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int ot = 0; ot < 4; ot++)
            *reinterpret_cast<float4*>(G + (int64_t)(ot * 16 + i16) * (9 * CI_C) + t * CI_C + wave * 16 + g * 4) =
                make_float4(acc[t][ot][0], acc[t][ot][1], acc[t][ot][2], acc[t][ot][3]);
    if (p.part_bias && g == 0) p.part_bias[(int64_t)blockIdx.x * CI_C + wave * 16 + i16] = accb[0];   // (every row of the ones product holds the same sums)
}

// gp[i] += sum over the pixel ranges of part[z][i], z in increasing order (deterministic)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gp, int nz,
                                                                const float* __restrict__ part_bias, float* __restrict__ gbias) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < CI_C * 9 * CI_C) {
        float s = 0.f;
        for (int z = 0; z < nz; z++) s += part[(int64_t)z * (CI_C * 9 * CI_C) + i];
        gp[i] += s;
    } else if (gbias && i < CI_C * 9 * CI_C + CI_C) {
        const int m = i - CI_C * 9 * CI_C;
        float s = 0.f;
        for (int z = 0; z < nz; z++) s += part_bias[(int64_t)z * CI_C + m];
        gbias[m] += s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------- conv1 (3 -> 64)
// The first convolution of the patch embedder (vision_embedding.py:44-50: 3 input channels, K = 27 padded to 32) as ONE streaming kernel per
// patch: the 1.5 KB of pixels go to LDS, every lane gathers the 8 column-matrix entries of its (pixel, k-group) -- that 16-byte piece IS the
// MFMA A fragment and is also stored as the column matrix the weight gradient contracts over -- and one v_mfma_f32_16x16x32_bf16 per
// (16 pixels x 16 outputs) finishes the contraction.  Reads 1.5 KB, writes 16 + 32 KB per patch; it replaces a scalar im2col (27 two-byte
// stores per pixel), a zero-padding pass and a K = 32 GEMM on the generic strided kernel: 3.3 ms -> ~0.7 ms at 60 160 patches.
template <typename TBIAS>
__global__ __launch_bounds__(256) void conv1_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const void* bias,
                                                          bf16_t* __restrict__ cols, bf16_t* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) bf16_t px[CI_HW * 3 + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t pix0 = (int64_t)blockIdx.x * CI_HW;
    if (tid < CI_HW * 3 / 8) *reinterpret_cast<uint4*>(px + tid * 8) = *reinterpret_cast<const uint4*>(x + pix0 * 3 + tid * 8);
    const int xm = lane & 15, g = lane >> 4;
    int off[8];          // LDS element offset of entry j relative to the pixel's own first channel, or a large negative marker for the zero padding k >= 27
    int dyj[8], dxj[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = 8 * g + j, tap = k / 3, c = k - 3 * tap;
        dyj[j] = k < 27 ? tap / 3 - 1 : 100;
        dxj[j] = tap % 3 - 1;
        off[j] = (dyj[j] * 16 + dxj[j]) * 3 + c;
    }
    bf16x8_t bfr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) bfr[j] = *reinterpret_cast<const bf16x8_t*>(wp + (j * 16 + xm) * 32 + 8 * g);
    __syncthreads();
    bf16_t* Y = y + pix0 * CI_C;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int yr = wave * 4 + i, pix = yr * 16 + xm;       // this lane's pixel of the tile = patch row yr
        bf16x8_t af;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int yy = yr + dyj[j], xx = xm + dxj[j];
            const bool in = yy >= 0 && yy < 16 && xx >= 0 && xx < 16;
            const short v = (short)px[in ? pix * 3 + off[j] : 0];
            af[j] = in ? v : (short)0;
        }
        *reinterpret_cast<bf16x8_t*>(cols + (pix0 + pix) * 32 + 8 * g) = af;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af, acc, 0, 0, 0);  // swapped: D[n][m]
            store_frag<bf16_t, TBIAS>(acc, Y, CI_C, pix, j * 16 + g * 4, 1.f, 0.f, bias);
        }
    }
}
extern "C" int db1_conv1_fused_fwd(const void* x, const void* w_op, const void* bias, void* cols, void* y, int64_t n_patches, int dtBias, void* stream) {
    if (n_patches <= 0 || n_patches > 8000000) DB1_FAIL(DB1_ERR_BAD_SHAPE, "conv1_fused_fwd: n_patches=%lld", (long long)n_patches);
    if (!x || !w_op || !cols || !y || !db1_aligned16(x) || !db1_aligned16(w_op) || !db1_aligned16(cols) || !db1_aligned16(y))
        DB1_FAIL(DB1_ERR_BAD_ALIGN, "conv1_fused_fwd: operands must be 16-byte aligned");
    if (bias && !db1_dt_ok(dtBias)) DB1_FAIL(DB1_ERR_UNSUPPORTED_DTYPE, "conv1_fused_fwd: bias dtype");
    hipStream_t st = (hipStream_t)stream;
    if (bias && dtBias == DB1_BF16) conv1_fused_kernel<bf16_t><<<(unsigned)n_patches, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w_op, bias, (bf16_t*)cols, (bf16_t*)y);
    else conv1_fused_kernel<float><<<(unsigned)n_patches, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w_op, bias, (bf16_t*)cols, (bf16_t*)y);
    DB1_CHECK_LAUNCH("conv1_fused_fwd");
    return DB1_OK;
}

// ---------------------------------------------------------------------------------------------------------------------- host side
extern "C" int db1_conv3x3_implicit_fwd_res(const void* x, const void* w_op, const void* bias, const void* res, void* y, int64_t n_patches, int sign,
                                            int dtBias, void* stream);
extern "C" int db1_conv3x3_implicit_fwd(const void* x, const void* w_op, const void* bias, void* y, int64_t n_patches, int sign, int dtBias,
                                        void* stream) {
    return db1_conv3x3_implicit_fwd_res(x, w_op, bias, nullptr, y, n_patches, sign, dtBias, stream);
}
extern "C" int db1_conv3x3_implicit_fwd_res(const void* x, const void* w_op, const void* bias, const void* res, void* y, int64_t n_patches, int sign,
                                            int dtBias, void* stream) {
    if (res && !db1_aligned16(res)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "conv3x3_implicit_fwd: residual must be 16-byte aligned");
    if (n_patches <= 0 || (sign != 1 && sign != -1)) DB1_FAIL(DB1_ERR_BAD_SHAPE, "conv3x3_implicit_fwd: n_patches=%lld sign=%d", (long long)n_patches, sign);
    if (!x || !w_op || !y || !db1_aligned16(x) || !db1_aligned16(w_op) || !db1_aligned16(y)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "conv3x3_implicit_fwd: operands must be 16-byte aligned");
    if (n_patches > 8000000) DB1_FAIL(DB1_ERR_BAD_SHAPE, "conv3x3_implicit_fwd: too many patches");
    ConvArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w_op; a.y = y; a.bias = bias; a.n_patches = n_patches; a.sign = sign; a.ksplit = 1; a.part = nullptr;
    a.gbias = nullptr; a.part_bias = nullptr; a.res = (const bf16_t*)res;
    static Db1PerDeviceOnce attr_once;
    attr_once.run([] {
        hipFuncSetAttribute((const void*)conv_implicit_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CI_STAGE_BYTES);
        hipFuncSetAttribute((const void*)conv_implicit_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CI_STAGE_BYTES);
    });
    hipStream_t st = (hipStream_t)stream;
    if (db1_knob(DB1_KNOB_CONV_PATCH, 1)) {     // (A/B knob "conv_patch": 0 = the nine-k-step tile form above)
        static Db1PerDeviceOnce attr_patch;
        attr_patch.run([] {
            hipFuncSetAttribute((const void*)conv_patch_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS);
            hipFuncSetAttribute((const void*)conv_patch_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS);
        });
        const unsigned grid = (unsigned)(n_patches < 256 ? n_patches : 256);     // one persistent workgroup per CU
        if (bias && dtBias == DB1_BF16) conv_patch_kernel<bf16_t><<<grid, 256, CP_LDS, st>>>(a);
        else conv_patch_kernel<float><<<grid, 256, CP_LDS, st>>>(a);
        DB1_CHECK_LAUNCH("conv3x3_implicit_fwd (patch-resident)");
        return DB1_OK;
    }
    if (bias && dtBias == DB1_BF16) conv_implicit_kernel<bf16_t><<<(unsigned)n_patches, 256, 2 * CI_STAGE_BYTES, st>>>(a);
    else conv_implicit_kernel<float><<<(unsigned)n_patches, 256, 2 * CI_STAGE_BYTES, st>>>(a);
    DB1_CHECK_LAUNCH("conv3x3_implicit_fwd");
    return DB1_OK;
}

static int ci_wgrad_patch_grid(int64_t n_patches) { return (int)(n_patches < 256 ? n_patches : 256); }   // one persistent workgroup per CU
static int ci_wgrad_ksplit(int64_t n_patches) {
    const int64_t nk = n_patches * (CI_HW / TBK);
    // 5 column tiles x ks pixel ranges workgroups, two per CU (64 KB of LDS each): 102 ranges = 510 workgroups are ONE round of the 512
    // slots (128 ranges = 640 left a second round a quarter full); DB1_CONV_WGRAD_KS overrides (A/B)
    const int env_ks = db1_knob(DB1_KNOB_CONV_WGRAD_KS, 0);   // A/B knob
    int ks = env_ks > 0 ? env_ks : 102;
    while (ks > 1 && nk / ks < 8) ks >>= 1;
    return ks;
}
extern "C" int64_t db1_conv3x3_implicit_wgrad_workspace_bytes(int64_t n_patches) {
    if (n_patches <= 0) return 0;
    const int ks = ci_wgrad_ksplit(n_patches), wg = ci_wgrad_patch_grid(n_patches);      // partial slabs of either form (the A/B knob picks at call time)
    return (int64_t)(ks > wg ? ks : wg) * (CI_C * 9 * CI_C + CI_C) * (int64_t)sizeof(float);
}
extern "C" int db1_conv3x3_implicit_wgrad(const void* dy, const void* x, float* gp_acc, float* gbias_acc, int64_t n_patches, void* ws, int64_t ws_bytes,
                                          void* stream) {
    if (n_patches <= 0 || n_patches > 8000000) DB1_FAIL(DB1_ERR_BAD_SHAPE, "conv3x3_implicit_wgrad: n_patches=%lld", (long long)n_patches);
    if (!dy || !x || !gp_acc || !db1_aligned16(dy) || !db1_aligned16(x)) DB1_FAIL(DB1_ERR_BAD_ALIGN, "conv3x3_implicit_wgrad: operands must be 16-byte aligned");
    ConvArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)dy; a.y = gp_acc; a.bias = nullptr; a.n_patches = n_patches; a.sign = 1;
    const bool patch_form = db1_knob(DB1_KNOB_CONV_PATCH, 1) != 0;
    const int ks = patch_form ? ci_wgrad_patch_grid(n_patches) : ci_wgrad_ksplit(n_patches);
    a.ksplit = ks;
    // per-range partial sums in the caller's workspace + a fixed-order reduce (bit-reproducible): there is no atomic form
    DB1_NEED_WS(ws, ws_bytes, db1_conv3x3_implicit_wgrad_workspace_bytes(n_patches), "conv3x3_implicit_wgrad");
    a.part = (float*)ws;
    a.gbias = gbias_acc; a.res = nullptr;
    a.part_bias = gbias_acc ? a.part + (int64_t)ks * (CI_C * 9 * CI_C) : nullptr;
    static Db1PerDeviceOnce attr_once;
    attr_once.run([] {
        hipFuncSetAttribute((const void*)conv_wgrad_implicit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
        hipFuncSetAttribute((const void*)conv_wgrad_patch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS);
    });
    if (patch_form) conv_wgrad_patch_kernel<<<dim3((unsigned)ks), 256, WP_LDS, (hipStream_t)stream>>>(a);
    else conv_wgrad_implicit_kernel<<<dim3(5, 1, (unsigned)ks), 256, 4 * TILE_BYTES, (hipStream_t)stream>>>(a);
    DB1_CHECK_LAUNCH("conv3x3_implicit_wgrad");
    if (a.part) {
        conv_wgrad_reduce_kernel<<<(CI_C * 9 * CI_C + CI_C + 255) / 256, 256, 0, (hipStream_t)stream>>>(a.part, gp_acc, ks, a.part_bias, gbias_acc);
        DB1_CHECK_LAUNCH("conv3x3_implicit_wgrad reduce");
    }
    return DB1_OK;
}
