"""Experiment: does a padded leading dimension (row stride not a multiple of 4 KiB) change the NT GEMM rate, and on which operand?"""
import os, sys
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))   # tools/ (bench_kernels) and the repository root
import torch
from bdm_db1_amd import ops
from bench_kernels import timeit
DEV = "cuda"
T, d = 16384, 2048
for name, M, N, K in [("qkv", T, 3 * d, d), ("o_net", T, d, d), ("ff1", T, 4 * d, d)]:
    for px, pw in ((0, 0), (128, 0), (0, 128), (128, 128)):
        x = torch.randn(M, K + px, device=DEV).to(torch.bfloat16)[:, :K]
        w = (torch.randn(N, K + pw, device=DEV) * 0.02).to(torch.bfloat16)[:, :K]
        y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(x, w.t(), y), iters=20)
        print(f"{name} NT pad x={px:3d} w={pw:3d}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TFLOP/s")
    # dx = dy W (NN): B = W [N x K] row-major
    dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    for pw in (0, 128):
        w = (torch.randn(N, K + pw, device=DEV) * 0.02).to(torch.bfloat16)[:, :K]
        dx = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(dy, w, dx), iters=20)
        print(f"{name} NN (dx) pad w={pw:3d}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TFLOP/s")
