"""Hand-scheduled flash forward (relattn_flash_fwd2.hip) against the compiled one on the same inputs: out, lse, p~ images, block maxima.
    python tools/exp/check_fwd2.py [B L H] ..."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from bdm_db1_amd import ops

DEV = torch.device("cuda", 0)
MODE = int(os.environ.get("FWD_MODE", 1))   # 1: the 4-wave loop, 2: the 8-wave loop


def run(B, L, H, seed=0, timing=False):
    D = 128
    g = torch.Generator(device="cpu").manual_seed(seed)
    qkv = (torch.randn(B, L, 3, H, D, generator=g) * 0.8).to(torch.bfloat16).to(DEV)
    R = torch.randn(L, H, D, generator=g).to(torch.bfloat16).to(DEV)
    u = (torch.randn(H, D, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    vb = (torch.randn(H, D, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    qu, qv = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16), torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    ops.relattn_add_head_bias(qkv, u, vb, qu, qv, B, L, L, H, D)
    scale = 1.0 / math.sqrt(D)
    res = []
    for on in (0, MODE):
        ops.flash_fwd2(on)
        out = torch.full((B, L, H, D), 7.0, device=DEV, dtype=torch.bfloat16)
        lse = torch.full((B, H, L), 3.0, device=DEV, dtype=torch.float32)
        probs = torch.full((B * H, L // 32, L // 16, 512), float("nan"), device=DEV, dtype=torch.bfloat16)
        mblk = torch.full((B * H, L // 32, L), float("nan"), device=DEV, dtype=torch.float32)
        ops.relattn_flash_fwd(qu, qv, qkv, R, out, lse, B, L, H, D, L, scale, probs=probs, mblk=mblk)
        torch.cuda.synchronize()
        if timing:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                ops.relattn_flash_fwd(qu, qv, qkv, R, out, lse, B, L, H, D, L, scale, probs=probs, mblk=mblk)
            ev0.record()
            for _ in range(10):
                ops.relattn_flash_fwd(qu, qv, qkv, R, out, lse, B, L, H, D, L, scale, probs=probs, mblk=mblk)
            ev1.record()
            torch.cuda.synchronize()
            print(f"  fwd2={on}: {ev0.elapsed_time(ev1) / 10 * 1e3:.1f} us")
        res.append((out.float().cpu(), lse.cpu(), probs.float().cpu(), mblk.cpu()))
    ops.flash_fwd2(1)
    (o0, l0, p0, m0), (o1, l1, p1, m1) = res
    nan_same = bool((torch.isnan(p0) == torch.isnan(p1)).all()) and bool((torch.isnan(m0) == torch.isnan(m1)).all())

    def prec(p, m, l):   # P = p~ exp2(m_blk c2 - lse log2 e): what the backward rebuilds; image [bh][jb][qt][lane][8], lane & 15 = query in its tile
        BH, NJ, NQ = p.shape[0], p.shape[1], p.shape[2]
        f = torch.exp2(m.view(BH, NJ, NQ, 1, 16) - (l.reshape(BH, 1, NQ, 1, 16) * 1.4426950408889634))   # [bh][jb][qt][1][a]
        f = f.expand(BH, NJ, NQ, 4, 16).reshape(BH, NJ, NQ, 64, 1)
        return p.view(BH, NJ, NQ, 64, 8) * f
    p0, p1 = prec(p0, m0, l0).reshape(p0.shape), prec(p1, m1, l1).reshape(p1.shape)
    m0, m1 = torch.zeros_like(m0), torch.zeros_like(m1)     # (the maxima may differ: deferred maximum; only the product above is defined)
    p0z, p1z, m0z, m1z = torch.nan_to_num(p0), torch.nan_to_num(p1), torch.nan_to_num(m0), torch.nan_to_num(m1)
    eo = float((o0 - o1).abs().max()) / max(float(o0.abs().max()), 1e-9)
    print(f"B={B} L={L} H={H}: out rel {eo:.2e}  lse abs {float((l0 - l1).abs().max()):.2e}  p~ abs {float((p0z - p1z).abs().max()):.2e}  "
          f"mblk abs {float((m0z - m1z).abs().max()):.2e}  nan-pattern same {nan_same}  finite out {bool(torch.isfinite(o1).all())}")
    ok = eo < 2e-2 and float((l0 - l1).abs().max()) < 1e-2 and float((p0z - p1z).abs().max()) < 2e-2 and float((m0z - m1z).abs().max()) < 1e-3 and nan_same
    if not ok:
        # where: per (bh, query tile of 16) error of out
        d = (o0 - o1).abs().amax(dim=3)   # [B, L, H]
        bad = torch.nonzero(d > 0.05 * float(o0.abs().max()))
        print("  first bad (b, i, h):", bad[:10].tolist(), " count", len(bad))
        dm = (m0z - m1z).abs()
        badm = torch.nonzero(dm > 1e-3)
        print("  first bad mblk (bh, jb, i):", badm[:10].tolist(), " count", len(badm))
    return ok


if __name__ == "__main__":
    if sys.argv[1:2] == ["time"]:
        run(64, 1024, 16, timing=True)
        sys.exit(0)
    a = [int(x) for x in sys.argv[1:]]
    cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(1, 128, 1), (1, 256, 1), (2, 256, 2), (1, 384, 1), (1, 1024, 1), (2, 1024, 16)]
    ok = all([run(*c) for c in cases])
    if ok and os.environ.get("TIMING", "1") == "1":
        run(64, 1024, 16, timing=True)
    print("ALL OK" if ok else "MISMATCH")
