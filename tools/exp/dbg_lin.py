import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bdm_db1_amd import ops
DEV = "cuda"
for M in (1, 2, 4, 5, 16, 22):
    d, eps, alpha = 2048, 1e-5, 0.81
    g = torch.Generator(device=DEV).manual_seed(M)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=g) * sc).to(torch.bfloat16)
    x, res = rnd(M, d), rnd(M, d)
    Wo = rnd(d, d, sc=0.03)
    gam, bet = rnd(d) + 1, rnd(d, sc=0.1)
    new = lambda *s: torch.empty(*s, device=DEV, dtype=torch.bfloat16)
    f32 = lambda *s: torch.empty(*s, device=DEV, dtype=torch.float32)
    y_ref, h_ref = new(M, d), new(M, d)
    ops.gemm(x, Wo.t(), y_ref)
    ops.layernorm_residual_fwd(res, y_ref, alpha, gam, bet, h_ref, None, f32(M), f32(M), eps)
    y, h = new(M, d), new(M, d)
    ops.linear_decode(x, Wo, None, y, ln=(res, alpha, gam, bet, eps, h))
    torch.cuda.synchronize()
    dy = (y.float() - y_ref.float()).abs(); dh = (h.float() - h_ref.float()).abs()
    print(M, "y diff rows", dy.amax(1).tolist(), "h diff rows", dh.amax(1).tolist(), "n bad h", (dh > 0).sum(1).tolist())
