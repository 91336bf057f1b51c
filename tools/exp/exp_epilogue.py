"""Experiment: epilogue cost of the bf16-output ping-pong GEMMs (beta = 0: bf16 staging; beta != 0: fp32 staging + C read)."""
import os, sys
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))   # tools/ (bench_kernels) and the repository root
import torch
from bdm_db1_amd import ops
from bench_kernels import timeit
DEV = "cuda"
M, N, K = 65536, 2048, 8192
dz = torch.randn(M, K, device=DEV).to(torch.bfloat16)
w = (torch.randn(K, N, device=DEV) * 0.02).to(torch.bfloat16)
y = torch.randn(M, N, device=DEV).to(torch.bfloat16)
for beta in (0.0, 0.9, 0.0, 0.9):
    t = timeit(lambda: ops.gemm(dz, w, y, beta=beta))
    print(f"NN M={M} N={N} K={K} beta={beta}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TFLOP/s")
