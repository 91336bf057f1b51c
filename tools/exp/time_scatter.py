import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bdm_db1_amd import ops
n, d = 65536, 2048
dout = torch.randn(n, d, device="cuda").bfloat16()
for rows in (22, 512, 32000):
    ids = torch.randint(0, rows, (n,), device="cuda")
    tab = torch.zeros(rows, d, device="cuda")
    f = lambda: ops.embed_scatter_add(dout, ids, tab)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"scatter-add {n} tokens into {rows} rows: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
