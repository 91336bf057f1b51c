"""Experiment: sensitivity of the NT GEMM rate to the base-address offset of W relative to x (memory-channel alignment)."""
import os, sys
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))   # tools/ (bench_kernels) and the repository root
import torch
from bdm_db1_amd import ops
from bench_kernels import timeit
DEV = "cuda"
T, d = 16384, 2048
for name, M, N, K in [("qkv", T, 3 * d, d), ("ff1", T, 4 * d, d), ("o_net", T, d, d)]:
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    wbuf = (torch.randn(N * K + 8192, device=DEV) * 0.02).to(torch.bfloat16)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    res = []
    for off_bytes in (0, 128, 256, 512, 1024, 2048, 4096 + 256):
        w = wbuf[off_bytes // 2: off_bytes // 2 + N * K].view(N, K)
        t = timeit(lambda: ops.gemm(x, w.t(), y), iters=20)
        res.append(f"{off_bytes}:{2.0 * M * N * K / t / 1e9:6.0f}")
    print(f"{name} NT TFLOP/s by W base offset (bytes): " + "  ".join(res), " x.ptr%4096=", x.data_ptr() % 4096, " w.ptr%65536=", wbuf.data_ptr() % 65536)
