import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bdm_db1_amd import ops
DEV = "cuda"; B, H, D, mlen, cap = 1, 16, 128, 1024, 1088
d = H * D
for q in (1, 16, 22, 48):
    qkv = torch.randn(B, q, 3, H, D, device=DEV).bfloat16(); u = torch.randn(H, D, device=DEV).bfloat16(); vb = torch.randn(H, D, device=DEV).bfloat16()
    ring = torch.randn(B, cap, 2, H, D, device=DEV).bfloat16(); R = torch.randn(cap, d, device=DEV).bfloat16()
    state = torch.zeros(1, dtype=torch.int32, device=DEV)
    av = torch.empty(B, q, H, D, device=DEV, dtype=torch.bfloat16)
    part = torch.empty(ops.relattn_decode_ring_part_numel(B, q, mlen + q, H), device=DEV, dtype=torch.float32)
    for name, kw in (("in-launch merge", dict(out=av)), ("separate merge", dict(out=av, fused_merge=False)), ("partials only", dict(out=None, part=part))):
        out = kw.pop("out")
        f = lambda: ops.relattn_decode_ring_fwd(qkv, u, vb, ring, state, R, out, B, q, mlen, H, D, mlen + q, 1 / math.sqrt(D), **kw)
        g = torch.cuda.CUDAGraph()
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20): f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        print(f"q={q:3d} {name:16s}: {e0.elapsed_time(e1) / 40 * 1e3:7.1f} us per call")
