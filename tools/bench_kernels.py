#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at DB1-1.3B sizes (HIP-event timing, random data).
    python tools/bench_kernels.py flash [B]     relative-position flash attention fwd / bwd
    python tools/bench_kernels.py gemm  [B]     the bf16 tile GEMM on the model's NT / NN / TN shapes
Used under rocprofv3 (--kernel-trace --stats, or --pmc ...) to attribute time and counters per kernel."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bdm_db1_amd import ops, lib  # noqa: E402
lib.apply_env_knobs()

DEV = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def flash(B=16, L=1024, H=16, D=128):
    torch.manual_seed(0)
    qkv = (torch.randn(B, L, 3, H, D, device=DEV) * 0.7).to(torch.bfloat16)
    R = torch.randn(L, H, D, device=DEV).to(torch.bfloat16)
    u = (torch.randn(H, D, device=DEV) * 0.3).to(torch.bfloat16)
    vb = (torch.randn(H, D, device=DEV) * 0.3).to(torch.bfloat16)
    qu, qv = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16), torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    ops.relattn_add_head_bias(qkv, u, vb, qu, qv, B, L, L, H, D)
    out = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device=DEV, dtype=torch.float32)
    dout = torch.randn(B, L, H, D, device=DEV).to(torch.bfloat16)
    dqkv = torch.empty(B, L, 3, H, D, device=DEV, dtype=torch.bfloat16)
    dT = torch.zeros(H, B, L, L, device=DEV, dtype=torch.bfloat16)
    delta = torch.empty(B, H, L, device=DEV, dtype=torch.float32)
    scale = 1.0 / math.sqrt(D)
    unit = 2.0 * B * H * (L * (L + 1) / 2) * D  # one causal-counted L x L x D contraction
    t = timeit(lambda: ops.relattn_flash_fwd(qu, qv, qkv, R, out, lse, B, L, H, D, L, scale))
    print(f"flash fwd  B={B}: {t * 1e3:8.1f} us   {3 * unit / t / 1e9:7.1f} TFLOP/s algorithmic (3 contractions)")
    probs = torch.empty(B * H, ops.relattn_flash_probs_tiles(L), 512, device=DEV, dtype=torch.bfloat16)
    mblk = torch.empty(B * H, L // 32, L, device=DEV, dtype=torch.float32)
    t = timeit(lambda: ops.relattn_flash_fwd(qu, qv, qkv, R, out, lse, B, L, H, D, L, scale, probs=probs, mblk=mblk))
    print(f"flash fwd  B={B} (+ stored p~): {t * 1e3:8.1f} us")
    t = timeit(lambda: ops.relattn_flash_bwd(qu, qv, qkv, R, out, dout, lse, delta, dqkv, dT, B, L, H, D, L, scale, probs=probs, mblk=mblk))
    print(f"flash bwd  B={B} (forward-stored p~): {t * 1e3:8.1f} us   {6 * unit / t / 1e9:7.1f} TFLOP/s algorithmic")
    for e in os.environ.get("FLASH_EXPS", "").split(","):
        if e:
            os.environ["DB1_FLASH_EXP"] = e
            t = timeit(lambda: ops.relattn_flash_bwd(qu, qv, qkv, R, out, dout, lse, delta, dqkv, dT, B, L, H, D, L, scale, probs=probs, mblk=mblk))
            print(f"flash bwd (fwd-stored) exp={e}: {t * 1e3:8.1f} us")
    os.environ.pop("DB1_FLASH_EXP", None)
    for sp in (True, False):
        t = timeit(lambda: ops.relattn_flash_bwd(qu, qv, qkv, R, out, dout, lse, delta, dqkv, dT, B, L, H, D, L, scale, store_probs=sp))
        print(f"flash bwd  B={B} ({'stored P/dS' if sp else 'recompute  '}): {t * 1e3:8.1f} us   {6 * unit / t / 1e9:7.1f} TFLOP/s algorithmic (6 contractions, excl. dq_r/dR GEMMs)")


def gemm(B=16, L=1024):
    T, d = B * L, 2048
    torch.manual_seed(0)
    shapes = [("qkv  NT", T, 3 * d, d), ("o    NT", T, d, d), ("ff1  NT", T, 4 * d, d), ("ff2  NT", T, d, 2 * d), ("head NT", T, 33280, d)]
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
        y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(x, w.t(), y))
        print(f"{name} M={M} N={N} K={K}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TFLOP/s")
        dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        dx = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(dy, w, dx))
        print(f"  dx NN M={M} N={K} K={N}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TFLOP/s")
        dw = torch.zeros(N, K, device=DEV, dtype=torch.float32)
        t = timeit(lambda: ops.gemm(dy.t(), x, dw, beta=1.0))
        print(f"  dW TN M={N} N={K} K={M}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TFLOP/s")


def small_out(B=16, L=1024):
    """the under-filled transposed-operand GEMMs of a layer's backward (split-K candidates)"""
    T, d, H, D = B * L, 2048, 16, 128
    torch.manual_seed(0)
    for name, M, N in [("o_net dW", d, d), ("ff2 dW", d, 2 * d), ("qkv dW", 3 * d, d)]:
        dy = torch.randn(T, M, device=DEV).to(torch.bfloat16)
        x = torch.randn(T, N, device=DEV).to(torch.bfloat16)
        dw = torch.zeros(M, N, device=DEV, dtype=torch.float32)
        t = timeit(lambda: ops.gemm(dy.t(), x, dw, beta=1.0))
        print(f"{name} TN M={M} N={N} K={T}: {t * 1e3:8.1f} us  {2.0 * M * N * T / t / 1e9:7.1f} TFLOP/s")
    dT = torch.randn(H, B, L, L, device=DEV).to(torch.bfloat16)
    qv = torch.randn(B, L, H, D, device=DEV).to(torch.bfloat16)
    dR = torch.empty(L, H * D, device=DEV, dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm_batched(dT.view(H, B * L, L).transpose(1, 2).unsqueeze(1), qv.view(B * L, H, D).permute(1, 0, 2).unsqueeze(1),
                                        dR.view(L, H, D).permute(1, 0, 2).unsqueeze(1)))
    print(f"dR per head M={L} N={D} K={T} x{H}: {t * 1e3:8.1f} us  {dT.numel() * 2 / t / 1e6:7.1f} GB/s of dT")
    t = timeit(lambda: ops.gemm_batched(dT.view(H, B * L, L).transpose(1, 2).unsqueeze(1), qv.view(B * L, H, D).permute(1, 0, 2).unsqueeze(1),
                                        dR.view(L, H, D).permute(1, 0, 2).unsqueeze(1), tri=(2, L)))
    print(f"dR with the zero hint: {t * 1e3:8.1f} us")
    R = torch.randn(L, H, D, device=DEV).to(torch.bfloat16)
    dqv = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    for tri in ((0, 0), (1, 0)):
        t = timeit(lambda: ops.gemm_batched(dT, R.permute(1, 0, 2).unsqueeze(1).expand(H, B, L, D), dqv.permute(2, 0, 1, 3), tri=tri))
        print(f"dq_r M={L} N={D} K={L} x{H * B} tri={tri[0]}: {t * 1e3:8.1f} us")
    if ops.relattn_dqr_supported(B, L, H, D, torch.bfloat16):
        dTc = dT * torch.tril(torch.ones(L, L, device=DEV)).to(torch.bfloat16)
        t = timeit(lambda: ops.relattn_dqr(dTc, R.view(L, H * D), dqv))
        print(f"dq_r stream kernel: {t * 1e3:8.1f} us  ({dT.numel() / t / 1e6:7.1f} GB/s of the causal half of dT)")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "flash"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    {"flash": flash, "gemm": gemm, "small_out": small_out}[which](B)
