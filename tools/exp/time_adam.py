"""same-box A/B of the two Adam kernels (knob adam_nt) on the DB1-1.3B arena size: 1 210 585 216 parameters, fp32 gradients, bf16 working copy"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bdm_db1_amd import lib, ops
n = 1_210_585_216
dev = torch.device("cuda", 0)
p = torch.randn(n, device=dev) * 0.02
g = torch.randn(n, device=dev) * 1e-3
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
w = torch.empty(n, device=dev, dtype=torch.bfloat16)
nsq = torch.ones(1, device=dev)
for rnd in range(3):
    for nt in (0, 1):
        lib.set_knob("adam_nt", nt)
        for _ in range(2):
            ops.adam_step(p, g, m, v, w, 1e-4, 0.9, 0.999, 1e-8, 0.01, False, 3, gscale=1.0, clip=1.0, norm_sq=nsq)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.adam_step(p, g, m, v, w, 1e-4, 0.9, 0.999, 1e-8, 0.01, False, 3, gscale=1.0, clip=1.0, norm_sq=nsq)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"round {rnd} adam_nt={nt}: {ms:.3f} ms  {30 * n / ms / 1e9:.2f} TB/s")
