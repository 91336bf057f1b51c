"""Experiment: the in-step GEMM shapes of DB1-1.3B at 64 sequences (all three operand layouts) through ops.gemm, for A/B runs of
the 4-wave hand-scheduled kernels (DB1_W4=0|1|2, or a rebuilt schedule variant: W4_VARIANT=... python tools/gen_gemm_w4.py).
Usage: python tools/exp/exp_w4.py [reps]"""
import os, sys
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))   # tools/ (bench_kernels) and the repository root
import torch
from bdm_db1_amd import ops
from bdm_db1_amd import lib; lib.apply_env_knobs()   # DB1_* A/B switches of this script -> the library's thread-local knobs
from bench_kernels import timeit

DEV = "cuda"
torch.manual_seed(0)


def mk(M, N, K, mode):
    if mode == "NT":
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16); b = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16).t()
    elif mode == "NN":
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16); b = (torch.randn(K, N, device=DEV) * 0.02).to(torch.bfloat16)
    else:
        a = torch.randn(K, M, device=DEV).to(torch.bfloat16).t(); b = (torch.randn(K, N, device=DEV) * 0.02).to(torch.bfloat16)
    return a, b


def tm(name, M, N, K, mode, odt=torch.bfloat16, beta=0.0):
    a, b = mk(M, N, K, mode)
    y = torch.zeros(M, N, device=DEV, dtype=odt)
    t = timeit(lambda: ops.gemm(a, b, y, beta=beta))
    print(f"{name:8s} {mode} {M}x{N}x{K} {str(odt)[6:]:8s} {t * 1e3:8.1f} us {2 * M * N * K / t / 1e9:7.0f} TFLOP/s", flush=True)
    return t


tot = 0.0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    tot = 0.0
    tot += tm("ff1", 65536, 8192, 2048, "NT")
    tot += tm("qkv", 65536, 6144, 2048, "NT")
    tot += tm("ff2", 65536, 2048, 4096, "NT")
    tot += tm("proj", 65536, 2048, 2048, "NT")
    tot += tm("dff1", 65536, 2048, 8192, "NN", beta=1.0)    # in the step (post-LN): accumulated in place onto the residual gradient
    tot += tm("dff2", 65536, 4096, 2048, "NN")
    tot += tm("dqkv", 65536, 2048, 6144, "NN", beta=1.0)
    tot += tm("dproj", 65536, 2048, 2048, "NN")
    tot += tm("wff1", 8192, 2048, 65536, "TN", torch.float32, 1.0)
    tot += tm("wqkv", 6144, 2048, 65536, "TN", torch.float32, 1.0)
    tot += tm("wff2", 2048, 4096, 65536, "TN", torch.float32, 1.0)
    tot += tm("wproj", 2048, 2048, 65536, "TN", torch.float32, 1.0)
    print(f"sum {tot * 1e3:.0f} us  (x 24 layers = {tot * 24:.1f} ms)", flush=True)
