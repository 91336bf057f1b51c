import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bdm_db1_amd import ops
T, V, ld = 16384, 33025, 33280
lg = torch.randn(T, ld, device="cuda").bfloat16(); labels = torch.randint(0, V, (T,), device="cuda"); mask = torch.ones(T, device="cuda")
lse = torch.empty(T, device="cuda"); sums = torch.zeros(2, device="cuda"); norm = torch.tensor([0.0, float(T)], device="cuda"); dl = torch.empty_like(lg)
def t(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("ce_fwd  %.0f us" % t(lambda: ops.masked_ce_fwd(lg, labels, mask, lse, sums, V)))
print("ce_bwd  %.0f us" % t(lambda: ops.masked_ce_bwd(lg, labels, mask, lse, norm, dl, V)))
print("one pass %.0f us" % t(lambda: ops.masked_ce_fwd_bwd(lg, labels, mask, lse, sums, norm, V)))
