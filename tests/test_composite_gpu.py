"""The composite C-ABI entry points (include/db1_hip.h, csrc/composite.hip): db1_grad_norm_sq, db1_lmhead_ce_bwd,
db1_relattn_{fwd,bwd}, db1_patch_embed_{fwd,bwd} -- called through ctypes exactly as any non-Python host would call them -- against
the per-op path they sequence (same kernels: equal results) and against the CPU oracle; plus a plain-C program (tests/csrc/
composite_main.c, no Python in the process) that drives two of them."""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import db1_oracle as O  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def bf(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).to(DEV)


def rel(got, want):
    want = np.asarray(want, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - want).max() / (np.abs(want).max() + 1e-30))


def test_grad_norm_sq():
    from bdm_db1_amd import lib, ops
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1_000_003).astype(np.float32)
    for t in (torch.from_numpy(x).to(DEV), bf(x)):
        acc = torch.full((1,), 123.0, device=DEV)     # overwritten, not accumulated
        n = int(lib.load().db1_grad_norm_sq_workspace_bytes(t.numel()))
        ws = torch.empty(n, device=DEV, dtype=torch.uint8)
        vals = set()
        for _ in range(5):
            lib.call("db1_grad_norm_sq", ops.P(t), ops.P(acc), t.numel(), ops.dt_code(t), ops.P(ws), n, ops.stream())
            vals.add(float(acc))
        want = float((t.double().cpu().numpy() ** 2).sum())
        assert len(vals) == 1 and abs(float(acc) - want) < 1e-5 * want      # bit-reproducible (no atomics)


def test_lmhead_ce_bwd_equals_the_fused_sweep():
    """forward-only sweep (lse, sums) + db1_lmhead_ce_bwd == db1_lmhead_ce_fwd_bwd (the same kernels in the same order)"""
    from bdm_db1_amd import lib, ops
    rng = np.random.default_rng(1)
    T, d, V, rows = 700, 128, 1000, 1024
    h, W = bf(rng.standard_normal((T, d))), bf(np.concatenate([rng.standard_normal((V, d)) * 0.3, np.zeros((rows - V, d))]))
    lab = torch.from_numpy(rng.integers(0, V, T)).to(DEV)
    msk = torch.from_numpy((rng.random(T) > 0.2).astype(np.float32)).to(DEV)
    lse1, sums1 = torch.empty(T, device=DEV), torch.zeros(2, device=DEV)
    dh1, dW1 = torch.empty(T, d, device=DEV, dtype=torch.bfloat16), torch.full((rows, d), 0.5, device=DEV)
    ops.lmhead_ce(h, W, lab, msk, lse1, sums1, V, dh=dh1, dW_acc=dW1, beta_dw=1.0, gscale=0.5, chunk_rows=256)
    lse2, sums2 = torch.empty(T, device=DEV), torch.zeros(2, device=DEV)
    ops.lmhead_ce(h, W, lab, msk, lse2, sums2, V, chunk_rows=256)
    dh2, dW2 = torch.empty(T, d, device=DEV, dtype=torch.bfloat16), torch.full((rows, d), 0.5, device=DEV)
    n = int(lib.load().db1_lmhead_ce_bwd_workspace_bytes(T, rows, d, 256, ops.dt_code(h)))
    ws = torch.empty(n, device=DEV, dtype=torch.uint8)
    lib.call("db1_lmhead_ce_bwd", ops.P(h), ops.P(W), ops.P(lab), ops.P(msk), ops.P(lse2), ops.P(sums2), ops.P(dh2), ops.P(dW2), 1.0, 0.5, T, V, rows, d, 256,
             ops.dt_code(h), ops.P(ws), n, ops.stream())
    torch.cuda.synchronize()
    assert torch.equal(lse1, lse2) and torch.equal(sums1, sums2) and torch.equal(dh1, dh2) and torch.equal(dW1, dW2)
    assert float(dW2[V:].abs().max()) == 0.5     # padded vocabulary rows: gradient exactly zero (the accumulator keeps its content)


@pytest.mark.parametrize("keep_probs", [True, False])
def test_relattn_composites_match_the_per_op_path_and_the_oracle(keep_probs):
    from bdm_db1_amd import lib, ops
    B, L, H, D = 2, 256, 2, 128
    rng = np.random.default_rng(5)
    to16 = lambda a: np.asarray(torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy(), np.float64)
    qkv, R = to16(rng.standard_normal((B, L, 3, H, D)) * 0.8), to16(rng.standard_normal((L, H, D)))
    u, vb = to16(rng.standard_normal((H, D)) * 0.5), to16(rng.standard_normal((H, D)) * 0.5)
    dout = to16(rng.standard_normal((B, L, H, D)))
    scale = 1.0 / math.sqrt(D)
    QKV, Rd, U, VB, DO = bf(qkv), bf(R), bf(u), bf(vb), bf(dout)
    new = lambda *s, dt=torch.bfloat16: torch.empty(*s, device=DEV, dtype=dt)
    vp0 = ctypes.c_void_p(0)

    def composite():
        qu, qv, out, lse = new(B, L, H, D), new(B, L, H, D), new(B, L, H, D), new(B, H, L, dt=torch.float32)
        probs = torch.full((B * H, ops.relattn_flash_probs_tiles(L), 512), float("nan"), device=DEV, dtype=torch.bfloat16) if keep_probs else None
        mblk = torch.full((B * H, L // 32, L), float("nan"), device=DEV) if keep_probs else None
        lib.call("db1_relattn_fwd", ops.P(QKV), ops.P(U), ops.P(VB), ops.P(Rd), ops.P(qu), ops.P(qv), ops.P(out), ops.P(lse),
                 ops.P(probs) if keep_probs else vp0, ops.P(mblk) if keep_probs else vp0, B, L, H, D, L, scale, ops.stream())
        dqkv, dR = torch.full((B, L, 3, H, D), 3.0, device=DEV, dtype=torch.bfloat16), new(L, H, D)
        du, dv = torch.zeros(H, D, device=DEV), torch.zeros(H, D, device=DEV)
        dT = torch.zeros(H, B, L, L, device=DEV, dtype=torch.bfloat16)
        n = int(lib.load().db1_relattn_bwd_workspace_bytes(B, L, H, D, int(keep_probs)))
        ws = torch.empty(n, device=DEV, dtype=torch.uint8)
        lib.call("db1_relattn_bwd", ops.P(QKV), ops.P(qu), ops.P(qv), ops.P(Rd), ops.P(out), ops.P(DO), ops.P(lse), ops.P(probs) if keep_probs else vp0,
                 ops.P(mblk) if keep_probs else vp0, ops.P(dqkv), ops.P(dR), ops.P(du), ops.P(dv), ops.P(dT), B, L, H, D, L, scale, ops.P(ws), n, ops.stream())
        torch.cuda.synchronize()
        return out, lse, dqkv, dR, du, dv

    def per_op():
        qu, qv, out, lse = new(B, L, H, D), new(B, L, H, D), new(B, L, H, D), new(B, H, L, dt=torch.float32)
        ops.relattn_add_head_bias(QKV, U, VB, qu, qv, B, L, L, H, D)
        probs = torch.full((B * H, ops.relattn_flash_probs_tiles(L), 512), float("nan"), device=DEV, dtype=torch.bfloat16) if keep_probs else None
        mblk = torch.full((B * H, L // 32, L), float("nan"), device=DEV) if keep_probs else None
        ops.relattn_flash_fwd(qu, qv, QKV, Rd, out, lse, B, L, H, D, L, scale, probs=probs, mblk=mblk)
        dqkv, dR = torch.full((B, L, 3, H, D), 3.0, device=DEV, dtype=torch.bfloat16), new(L, H, D)
        du, dv = torch.zeros(H, D, device=DEV), torch.zeros(H, D, device=DEV)
        dT = torch.zeros(H, B, L, L, device=DEV, dtype=torch.bfloat16)
        delta = new(B, H, L, dt=torch.float32)
        ops.relattn_flash_bwd(qu, qv, QKV, Rd, out, DO, lse, delta, dqkv, dT, B, L, H, D, L, scale, store_probs=False, probs=probs, mblk=mblk)
        ops.relattn_dqr_fused(dT, Rd.view(L, H * D), dqkv[:, :, 0], du.view(-1), dv.view(-1))
        ops.gemm_batched(dT.view(H, B * L, L).transpose(1, 2).unsqueeze(1), qv.view(B * L, H, D).permute(1, 0, 2).unsqueeze(1),
                         dR.view(L, H, D).permute(1, 0, 2).unsqueeze(1), tri=(2, L))
        torch.cuda.synchronize()
        return out, lse, dqkv, dR, du, dv
    got, ref = composite(), per_op()
    for a, b_, name in zip(got, ref, ("out", "lse", "dqkv", "dR", "du", "dv")):
        assert torch.equal(a, b_), name
    masked = (~(np.arange(L)[None, :] <= np.arange(L)[:, None])).astype(np.uint8)
    out_ref, cache = O.relattn_core_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], R, u, vb, masked, scale)
    dq_ref, dk_ref, dv_ref, dR_ref, du_ref, dvb_ref = O.relattn_core_bwd(dout, qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], R, u, vb, scale, cache)
    out, lse, dqkv, dR, du, dv = [x.float().cpu().numpy() for x in got]
    assert rel(out, out_ref) < 2e-2
    for g, r, name in ((dqkv[:, :, 0], dq_ref, "dq"), (dqkv[:, :, 1], dk_ref, "dk"), (dqkv[:, :, 2], dv_ref, "dv"), (dR, dR_ref, "dR"),
                       (du, du_ref, "du"), (dv, dvb_ref, "dv_bias")):
        assert rel(g, r) < 3e-2, name


def _pe_params(rng, C, d):
    pe = "vision_encoder.patch_embeddings."
    q = lambda *s, sc=1.0: np.asarray(torch.from_numpy((rng.standard_normal(s) * sc).astype(np.float32)).to(torch.bfloat16).float().numpy(), np.float64)
    return {pe + "conv1.weight": q(64, C, 3, 3, sc=0.2), pe + "conv1.bias": q(64, sc=0.05),
            pe + "residual_path.0.weight": 1 + q(64, sc=0.1), pe + "residual_path.0.bias": q(64, sc=0.05),
            pe + "residual_path.2.weight": q(64, 64, 3, 3, sc=0.05), pe + "residual_path.2.bias": q(64, sc=0.05),
            pe + "residual_path.3.weight": 1 + q(64, sc=0.1), pe + "residual_path.3.bias": q(64, sc=0.05),
            pe + "residual_path.5.weight": q(64, 64, 3, 3, sc=0.05), pe + "residual_path.5.bias": q(64, sc=0.05),
            pe + "projection.weight": q(d, 64, 16, 16, sc=0.01), pe + "projection.bias": q(d, sc=0.05)}


PE_ORDER = ["conv1.weight", "conv1.bias", "residual_path.0.weight", "residual_path.0.bias", "residual_path.2.weight", "residual_path.2.bias",
            "residual_path.3.weight", "residual_path.3.bias", "residual_path.5.weight", "residual_path.5.bias", "projection.weight", "projection.bias"]


def test_patch_embed_composites_match_the_oracle():
    """db1_patch_embed_fwd / _bwd (vision_embedding.py:65-86) on 3 images of 3 x 32 x 48 (6 patches each) against the NumPy oracle"""
    from bdm_db1_amd import lib, ops
    rng = np.random.default_rng(9)
    n_img, C, Hh, Ww, p, d = 3, 3, 32, 48, 16, 256
    params = _pe_params(rng, C, d)
    pe = "vision_encoder.patch_embeddings."
    pixels = rng.random((n_img, C, Hh, Ww)).astype(np.float32) * 255.0
    N = n_img * (Hh // p) * (Ww // p)
    demb = np.asarray(bf(rng.standard_normal((N, d))).float().cpu().numpy(), np.float64)
    out_ref, cache = O.patch_embed_fwd(params, pixels.astype(np.float64), p)
    g_ref = O.patch_embed_bwd(params, demb.reshape(n_img, -1, d), cache)
    W = [bf(params[pe + n]) for n in PE_ORDER]
    G = [torch.full(tuple(params[pe + n].shape), 0.25, device=DEV, dtype=torch.float32) for n in PE_ORDER]   # accumulators: += on top of 0.25
    wptr = (ctypes.c_void_p * 12)(*[t.data_ptr() for t in W])
    gptr = (ctypes.c_void_p * 12)(*[t.data_ptr() for t in G])
    L_ = lib.load()
    save = torch.empty(int(L_.db1_patch_embed_save_bytes(n_img, C, Hh, Ww, p)), device=DEV, dtype=torch.uint8)
    n_f, n_b = int(L_.db1_patch_embed_workspace_bytes(n_img, C, Hh, Ww, p, d, 0)), int(L_.db1_patch_embed_workspace_bytes(n_img, C, Hh, Ww, p, d, 1))
    ws = torch.empty(max(n_f, n_b), device=DEV, dtype=torch.uint8)
    px = torch.from_numpy(pixels).to(DEV)
    emb = torch.empty(N, d, device=DEV, dtype=torch.bfloat16)
    lib.call("db1_patch_embed_fwd", ops.P(px), ctypes.cast(wptr, ctypes.c_void_p), ops.P(emb), ops.P(save), n_img, C, Hh, Ww, p, d, ops.P(ws), n_f, ops.stream())
    lib.call("db1_patch_embed_bwd", ops.P(bf(demb)), ctypes.cast(wptr, ctypes.c_void_p), ops.P(save), ctypes.cast(gptr, ctypes.c_void_p), n_img, C, Hh, Ww, p, d,
             ops.P(ws), n_b, ops.stream())
    torch.cuda.synchronize()
    assert rel(emb.float().cpu().numpy(), out_ref.reshape(N, d)) < 3e-2
    for n, g in zip(PE_ORDER, G):
        got = g.cpu().numpy().astype(np.float64) - 0.25
        assert rel(got, g_ref[pe + n]) < 6e-2, n
    # too small a workspace is an error, not a crash
    with pytest.raises(lib.Db1Error, match="workspace"):
        lib.call("db1_patch_embed_fwd", ops.P(px), ctypes.cast(wptr, ctypes.c_void_p), ops.P(emb), ops.P(save), n_img, C, Hh, Ww, p, d, ops.P(ws), 1024, ops.stream())


def test_plain_c_host_drives_the_composites(tmp_path):
    """tests/csrc/composite_main.c: a C program (hipMalloc / hipMemcpy + the C ABI, no Python, no torch in the process) runs
    db1_grad_norm_sq and db1_patch_embed_fwd and checks their results against values it computes itself"""
    exe = str(tmp_path / "composite_main")
    src = os.path.join(ROOT, "tests", "csrc", "composite_main.c")
    libdir = os.path.join(ROOT, "bdm_db1_amd")
    cc = subprocess.run(["gcc", "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I/opt/rocm/include", src, "-o", exe,
                         "-L" + libdir, "-ldb1_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir + ":/opt/rocm/lib"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "grad_norm_sq ok" in r.stdout and "patch_embed_fwd ok" in r.stdout
