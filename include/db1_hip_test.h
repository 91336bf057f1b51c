/*
 * db1_hip_test.h -- TEST-ONLY symbols of libdb1_hip.so.  Not part of the product ABI (include/db1_hip.h): production callers never
 * use them.  They steer which GEMM kernel the dispatcher takes so that the parity tests can cross-check every tile kernel at full
 * size against the strided fp32-MFMA kernel, and so that tuning scripts can time one kernel.  The state they set is THREAD-LOCAL
 * (it affects only GEMM calls made afterwards by the same host thread).
 */
#ifndef DB1_HIP_TEST_H
#define DB1_HIP_TEST_H

#ifdef __cplusplus
extern "C" {
#endif

/* route every GEMM of this thread to the strided fp32-MFMA kernel (on != 0) */
void db1_test_gemm_force_generic(int on);
/* pin one bf16 tile kernel where its shape constraints hold: 128 = 128x128, 256 = 256x128 3-stage, 512 = 256x256 8-wave ping-pong
 * (2 stages of k64), 1024 = the same with a 4-stage ring of k32; 0 = the measured heuristics (the default) */
void db1_test_gemm_tile_override(int tile);

/* the flash-attention forward that keeps its probabilities: 0 = the compiled key-block loop (relattn_flash.hip), 1 = the default
 * dispatch (the hand-scheduled 4-wave loop, relattn_flash_fwd3.hip), 2 = the hand-scheduled 8-wave loop (relattn_flash_fwd2.hip);
 * lets the parity tests compare them on the same inputs */
void db1_test_flash_fwd2(int on);

/* A/B knobs of the dispatchers for measurements: THREAD-LOCAL (they affect calls made afterwards by the same host thread), unset by
 * default; the library never reads the environment.  Names: "gemm_tile" (128 | 256 | 512 | 1024: pin one tile kernel), "gemm_splitk" (0: no
 * workspace split-K), "pp32_stages" (4 | 5), "linear_decode_splitk" (0 | 1: never split), "w4" (0: the 8-wave GEMM kernels, 2: 4-wave for NT
 * only), "flash_fwd2" (0: the compiled flash-forward loop), "flash_kv3" (0 / 1: force 16 / 32 keys per wave in the key-side backward; -1 or unset: 32 from 512 workgroups on), "conv_wgrad_ks",
 * "geglu_epi" (0: db1_gemm_nt_geglu / db1_gemm_nn_geglu_bwd run as separate GEMM + activation launches at every shape), "gemm_halfwave" (k-tiles
 * per slice from which half-wave outputs are split in two), "w4n" (0: no 256 x 128 tiles of the 4-wave GEMM; 1 / 2 / 3: see gemm.hip), "conv_patch" (0: the 3x3 convolutions' forward / data gradient as nine k-steps of a tile GEMM instead of the patch-resident kernel).
 * Returns 0, or DB1_ERR_BAD_SHAPE for an unknown name. */
int db1_test_set_knob(const char* name, int value);
void db1_test_clear_knobs(void);
/* 1 if the library was compiled with -DDB1_EXPERIMENT (timing ablations, wrong results by construction): loaders must refuse it */
int db1_is_experiment_build(void);
/* an empty kernel named db1_marker_kernel on the stream: lets a kernel trace be cut to the timed region of bench.py (tools/prof_table.py) */
int db1_test_marker(int tag, void* stream);
/* db1_decode_chain launches made afterwards write workgroup 0's stage times (wall_clock64 ticks of 10 ns) to buf[0 .. 15] and every
 * workgroup's first four to buf[16 + 4 workgroup ..] (device memory, 16 + 2048 int64: [1040 + 4 workgroup ..] worker wave 0's W0 issued / merge inputs requested / W1 issued / merged row written;
 * null switches it off): start, A0, B0, y_o complete, LN1 done, A1, B1, act complete, act in LDS, A2, B2, f complete, LN2 done, A3, B3 */
void db1_test_decode_chain_timestamps(void* buf, int slot);   /* slot < 0: every launch; else only launches with that slot */

/* Box calibration (bench.py `box`): `iters` rounds of 16 independent v_mfma_f32_16x16x32_bf16 per wave, operands in registers (no memory
 * traffic), one wave per SIMD of every CU, on random-mantissa operand bits (random_bits != 0) or zeros; asynchronous on `stream` -- the
 * caller times it.  sink: >= 65 536 floats, never written. */
int db1_test_mfma_calibration(float* sink, int iters, int random_bits, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DB1_HIP_TEST_H */
