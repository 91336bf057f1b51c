"""same-box A/B: the feed-forward GEMMs with the GEGLU epilogues in the z form and in the saved-factor form (DB1-1.3B shapes, 64 sequences)"""
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))
import torch
from bdm_db1_amd import ops
from bench_kernels import timeit
DEV = "cuda"
M, d, dff = int(sys.argv[1]) * 1024 if len(sys.argv) > 1 else 65536, 2048, 4096
bf = torch.bfloat16
x = torch.randn(M, d, device=DEV).to(bf); W1 = (torch.randn(2 * dff, d, device=DEV) * 0.02).to(bf); b1 = torch.zeros(2 * dff, device=DEV, dtype=bf)
dy = torch.randn(M, d, device=DEV).to(bf); W2 = (torch.randn(d, dff, device=DEV) * 0.02).to(bf)
z = torch.empty(M, 2 * dff, device=DEV, dtype=bf); act = torch.empty(M, dff, device=DEV, dtype=bf); dz = torch.empty_like(z); gb = torch.zeros(2 * dff, device=DEV)
for rnd in range(3):
    t = {}
    t["ff1 z"] = timeit(lambda: ops.gemm_nt_geglu(x, W1, b1, z, act))
    t["dff2 z"] = timeit(lambda: ops.gemm_nn_geglu_bwd(dy, W2, z, dz, gb))
    t["ff1 saved"] = timeit(lambda: ops.gemm_nt_geglu_saved(x, W1, b1, z, act))
    t["dff2 saved"] = timeit(lambda: ops.gemm_nn_geglu_bwd_saved(dy, W2, z, dz, gb))
    t["ff1 plain"] = timeit(lambda: ops.gemm(x, W1.t(), z))
    da = torch.empty(M, dff, device=DEV, dtype=bf)
    t["dff2 plain"] = timeit(lambda: ops.gemm(dy, W2, da))
    print("  ".join(f"{k} {v * 1e3:.1f} us" for k, v in t.items()), flush=True)
