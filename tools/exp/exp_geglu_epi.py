"""the two feed-forward GEMMs with the activation in their epilogues at the 64-sequence step's size, timed alone (HIP events)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bdm_db1_amd import lib, ops
dev = torch.device("cuda", 0)
M, d, dff = 65536, 2048, 4096
torch.manual_seed(0)
x = torch.randn(M, d, device=dev).to(torch.bfloat16)
w1 = (torch.randn(2 * dff, d, device=dev) * 0.02).to(torch.bfloat16)
b1 = torch.randn(2 * dff, device=dev).to(torch.bfloat16)
w2 = (torch.randn(d, dff, device=dev) * 0.02).to(torch.bfloat16)
z = torch.empty(M, 2 * dff, device=dev, dtype=torch.bfloat16)
act = torch.empty(M, dff, device=dev, dtype=torch.bfloat16)
dy = torch.randn(M, d, device=dev).to(torch.bfloat16)
dz = torch.empty(M, 2 * dff, device=dev, dtype=torch.bfloat16)
parts = torch.empty(M // 128, 2 * dff, device=dev)
plain = torch.empty(M, dff, device=dev, dtype=torch.bfloat16)
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(2):
    t1 = timeit(lambda: ops.gemm_nt_geglu(x, w1, b1, z, act))
    t2 = timeit(lambda: ops.gemm_nn_geglu_bwd_parts(dy, w2, z, dz, parts))
    t3 = timeit(lambda: ops.gemm(dy, w2, plain))                       # the same product without the epilogue (dact stored)
    t4 = timeit(lambda: ops.gemm(x, w1.t(), z, bias=b1))               # ff1 without the activation
    print(f"round {rnd}: ff1 + GEGLU {t1:.0f} us (plain ff1 {t4:.0f})   dff2 + GEGLU' {t2:.0f} us (plain dact {t3:.0f})")
