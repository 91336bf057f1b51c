"""Experiment: forward projections as NT (weights [N, K], the reference layout) vs NN (a transposed weight copy [K, N])."""
import os, sys
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))   # tools/ (bench_kernels) and the repository root
import torch
from bdm_db1_amd import ops
from bench_kernels import timeit
DEV = "cuda"
T, d = int(sys.argv[1]) * 1024 if len(sys.argv) > 1 else 16384, 2048
for name, M, N, K in [("qkv", T, 3 * d, d), ("o_net", T, d, d), ("ff1", T, 4 * d, d), ("ff2", T, d, 2 * d), ("head", T, 33280, d)]:
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    wt = w.t().contiguous()
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    t1 = timeit(lambda: ops.gemm(x, w.t(), y), iters=20)
    t2 = timeit(lambda: ops.gemm(x, wt, y), iters=20)
    print(f"{name} M={M} N={N} K={K}: NT {t1 * 1e3:8.1f} us {2.0 * M * N * K / t1 / 1e9:7.1f} TF | NN {t2 * 1e3:8.1f} us {2.0 * M * N * K / t2 / 1e9:7.1f} TF")
