"""same-box A/B of the per-head dR contraction (dR[d, h, :] = sum_{b, i} dT[h, b, i, d] qv[b, i, h, :], structural-zero hint 2) with 4 and with 8
k slices under the heavy-first walk of the 4-wave kernel (knob tri_split), at the 64-sequence step's size and in the window form (16 x 4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bdm_db1_amd import lib, ops
dev = torch.device("cuda", 0)
B, L, H, D = 64, 1024, 16, 128
torch.manual_seed(0)
dT = torch.randn(H, B, L, L, device=dev, dtype=torch.bfloat16) * 0.05
dT *= torch.tril(torch.ones(L, L, device=dev, dtype=torch.bfloat16))       # dT[h, b, i, d] == 0 for d > i
qv = torch.randn(B, L, H, D, device=dev, dtype=torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def one(dR):
    ops.gemm_batched(dT.view(H, B * L, L).transpose(1, 2).unsqueeze(1), qv.view(B * L, H, D).permute(1, 0, 2).unsqueeze(1),
                     dR.view(L, H, D).permute(1, 0, 2).unsqueeze(1), tri=(2, L))
def window(dR, ng):
    Bm = B // ng
    ops.gemm_batched(dT.view(H, ng, Bm * L, L).transpose(2, 3), qv.view(ng, Bm * L, H, D).permute(2, 0, 1, 3),
                     dR.view(ng, L, H, D).permute(2, 0, 1, 3), tri=(2, L))
ref = torch.einsum("hbid,bihe->dhe", dT[:, :8].float(), qv[:8].float())    # 8 sequences in fp32 (the full product is 2 x 16 GB of fp32 operands)
for rnd in range(2):
    for knob in (0, 1, 2):
        lib.set_knob("tri_split", knob); ops._ws_query_cache.clear()    # (the workspace query depends on the knob)
        dR = torch.full((L, H * D), float("nan"), device=dev)
        t = timeit(lambda: one(dR))
        dRw = torch.full((16 * L, H * D), float("nan"), device=dev)
        tw = timeit(lambda: window(dRw, 16))
        err = (dRw.view(16, L, H, D).sum(0) - dR.view(L, H, D)).abs().max().item()
        print(f"round {rnd} tri_split={knob}: dR {t:.0f} us   window form (16 x 4) {tw:.0f} us   |sum of window blocks - one product| max {err:.3e}  (|dR| max {dR.abs().max().item():.1f})")
        if rnd == 0:
            Bs = 8
            d8 = torch.full((L, H * D), float("nan"), device=dev)
            ops.gemm_batched(dT[:, :Bs].reshape(H, Bs * L, L).transpose(1, 2).unsqueeze(1), qv[:Bs].reshape(Bs * L, H, D).permute(1, 0, 2).unsqueeze(1),
                             d8.view(L, H, D).permute(1, 0, 2).unsqueeze(1), tri=(2, L))
            print(f"         8 sequences against fp32 einsum: max |diff| {(d8.view(L, H, D) - ref).abs().max().item():.3e} (|ref| max {ref.abs().max().item():.1f})")
if len(sys.argv) > 1:
    torch.save({"dR": dR.cpu()}, sys.argv[1])
