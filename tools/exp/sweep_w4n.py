"""Experiment: time the 256 x 128-tile 4-wave GEMM on the shapes it is dispatched for at 4 sequences (run after a schedule variant was
generated: W4N_A / W4N_B / W4N_RD / W4N_SP=... python tools/gen_gemm_w4.py && python -m bdm_db1_amd.build)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bdm_db1_amd import ops
from bench_kernels import timeit
T, d = 4096, 2048
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
tot = 0.0
out = []
for name, lay, M, N, K in (("o NT", "nt", T, d, d), ("ff2 NT", "nt", T, d, 2 * d), ("qkv NT", "nt", T, 3 * d, d), ("do NN", "nn", T, d, d), ("dqkv NN", "nn", T, d, 3 * d),
                           ("dff1 NN", "nn", T, d, 4 * d), ("wff2 TN", "tn", d, 2 * d, T)):
    if lay == "nt":
        a, b = r(M, K), (r(N, K) * 0.02).t()
    elif lay == "nn":
        a, b = r(M, K), r(K, N) * 0.02
    else:
        a, b = r(K, M).t(), r(K, N)
    y = torch.empty(M, N, device="cuda", dtype=torch.float32 if lay == "tn" else torch.bfloat16)
    assert ops.gemm_kernel_choice(a, b, y)[0] == "w4n", name
    t = min(timeit(lambda: ops.gemm(a, b, y), iters=20) for _ in range(3))
    tot += t
    out.append(f"{name} {t * 1e3:.1f}")
print(f"sum {tot * 1e3:.1f} us | " + " | ".join(out), flush=True)
