"""Kernel-level parity: every C-ABI entry point against the CPU oracle on seeded inputs.
Integer / index results are bit-exact; fp32 results to fp32 round-off; bf16 results are compared
with the oracle evaluated on the bf16-rounded inputs (tolerance = bf16 output rounding)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import db1_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bdm_db1_amd import ops as _ops, lib
    assert lib.load().db1_device_is_gfx950() == 1
    return _ops


DEV = "cuda"


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def dev16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(torch.bfloat16)


def bf(a):
    """round a float array to bf16 (returns float64 values that are exactly representable)"""
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


def close(got, ref, rtol, atol_scale=None, name=""):
    got = got.detach().to(torch.float64).cpu().numpy() if hasattr(got, "detach") else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(got - ref).max() / scale
    assert err <= rtol, f"{name}: max err {err:.3e} of scale {scale:.3e} > {rtol}"


# ------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_strided_odd_shapes(ops, form, dtype):
    rng = np.random.default_rng(1)
    M, N, K = 70, 130, 37
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N))
    bias = rng.standard_normal(N)
    C0 = rng.standard_normal((M, N))
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    if dtype == "bf16":
        A, B, bias = bf(A), bf(B), bf(bias)
    a_store = dev(A if form != "tn" else A.T.copy(), td)
    b_store = dev(B.T.copy() if form == "nt" else B, td)
    a = a_store if form != "tn" else a_store.t()
    b = b_store.t() if form == "nt" else b_store
    out = dev(C0, torch.float32)
    ops.gemm(a, b, out, bias=dev(bias, td), alpha=0.5, beta=2.0)
    close(out, 0.5 * A @ B + 2.0 * C0 + bias, 2e-6, name=f"gemm {form} {dtype}")


def test_gemm_batched_strided(ops):
    rng = np.random.default_rng(2)
    Z0, Z1, M, N, K = 2, 3, 33, 20, 50
    A = rng.standard_normal((Z0, Z1, M, K))
    B = rng.standard_normal((Z1, K, N))  # broadcast over z0
    a = dev(A)
    b = dev(B).unsqueeze(0).expand(Z0, Z1, K, N)
    out = torch.zeros(Z0, Z1, M, N, device=DEV)
    ops.gemm_batched(a, b, out)
    close(out, np.einsum("xymk,ykn->xymn", A, B), 2e-6, name="gemm batched")


def test_gemm_strided_millions_of_rows(ops):
    """the first conv of the patch embedder at an RL batch of 64 sequences is a GEMM with 15 M rows and K = 32:
    the row tiles must not sit on a 65535-limited grid dimension"""
    g = torch.Generator(device="cpu").manual_seed(3)
    M, N, K = 65536 * 64 + 192, 64, 32
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = torch.randn(N, K, generator=g).to(torch.bfloat16).to(DEV)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(a, w.t(), out)
    for sl in (slice(0, 4096), slice(M // 2, M // 2 + 4096), slice(M - 4096, M)):
        ref = a[sl].double().cpu().numpy() @ w.double().cpu().numpy().T
        close(out[sl], ref, 1e-2, name=f"gemm rows {sl.start}")


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
@pytest.mark.parametrize("out_dtype", ["f32", "bf16"])
def test_gemm_bf16_tile_kernels(ops, form, out_dtype):
    """the MFMA tile kernels (asymmetric operands; transposes would be caught)"""
    from bdm_db1_amd import lib
    rng = np.random.default_rng(3)
    M, N, K = 256, 384, 192
    A = bf(rng.standard_normal((M, K)))
    B = bf(rng.standard_normal((K, N)))
    bias = bf(rng.standard_normal(N))
    C0 = bf(rng.standard_normal((M, N)))
    a_store = dev(A if form != "tn" else A.T.copy(), torch.bfloat16)
    b_store = dev(B.T.copy() if form == "nt" else B, torch.bfloat16)
    a = a_store if form != "tn" else a_store.t()
    b = b_store.t() if form == "nt" else b_store
    od = torch.float32 if out_dtype == "f32" else torch.bfloat16
    out = dev(C0, od)
    assert lib.load().db1_gemm_would_use_fast(M, N, K, 1, 1, 0 if od == torch.float32 else 1, a.stride(0), a.stride(1),
                                              b.stride(0), b.stride(1), out.stride(0), out.stride(1)) == 1
    ops.gemm(a, b, out, bias=dev(bias, torch.bfloat16), alpha=0.25, beta=1.0)
    ref = 0.25 * A @ B + C0 + bias
    close(out, ref, 2e-6 if out_dtype == "f32" else 6e-3, name=f"tile gemm {form} {out_dtype}")
    # same call through the strided kernel must agree (accumulation order differs only)
    out2 = dev(C0, od)
    ops.gemm_force_generic(True)
    try:
        ops.gemm(a, b, out2, bias=dev(bias, torch.bfloat16), alpha=0.25, beta=1.0)
    finally:
        ops.gemm_force_generic(False)
    close(out2, ref, 2e-6 if out_dtype == "f32" else 6e-3, name=f"strided gemm {form} {out_dtype}")


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
@pytest.mark.parametrize("tile", [256, 512, 1024])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 320), (768, 512, 1024), (512, 768, 256), (768, 512, 384)])
def test_gemm_bf16_large_tile_kernels(ops, form, tile, M, N, K):
    """the 256x128 3-stage and the 256x256 kernels, pinned through the ABI hook: 1, odd and even k-tile counts (tiles 512 / 1024 with K a
    multiple of 128 from 256 on take the hand-scheduled 4-wave loops: 4, 6 and 16 k-tiles, all three operand layouts, fp32 and bf16
    outputs accumulated onto C with bias)"""
    rng = np.random.default_rng(7)
    A = bf(rng.standard_normal((M, K)))
    B = bf(rng.standard_normal((K, N)))
    a_store = dev(A if form != "tn" else A.T.copy(), torch.bfloat16)
    b_store = dev(B.T.copy() if form == "nt" else B, torch.bfloat16)
    a = a_store if form != "tn" else a_store.t()
    b = b_store.t() if form == "nt" else b_store
    bias = bf(rng.standard_normal(N))
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    ops.gemm_tile_override(tile)
    try:
        for od, tol in ((torch.float32, 2e-6), (torch.bfloat16, 6e-3)):
            c0 = bf(C0) if od == torch.bfloat16 else C0
            out = dev(c0, od)
            ops.gemm(a, b, out, bias=dev(bias, torch.bfloat16), alpha=0.5, beta=1.0)
            close(out, 0.5 * A @ B + c0 + bias, tol, name=f"tile{tile} gemm {form} {od}")
    finally:
        ops.gemm_tile_override(0)


@pytest.mark.parametrize("M", [1, 5, 16, 22, 33, 64])
@pytest.mark.parametrize("N,K,od", [(2048, 2048, "bf16"), (8192, 2048, "bf16"), (2048, 4096, "f32"), (33280, 2048, "bf16")])
def test_gemm_skinny_few_rows(ops, M, N, K, od):
    """y = x W^T (+ bias, accumulate) with 1..64 rows: the W-streaming kernel used by inference with memory"""
    rng = np.random.default_rng(M * 7 + N)
    x = bf(rng.standard_normal((M, K)))
    w = bf(rng.standard_normal((N, K)) * 0.05)
    bias = bf(rng.standard_normal(N))
    td = torch.float32 if od == "f32" else torch.bfloat16
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    c0 = c0 if od == "f32" else bf(c0)
    out = dev(c0, td)
    ops.gemm(dev(x, torch.bfloat16), dev(w, torch.bfloat16).t(), out, bias=dev(bias, torch.bfloat16), alpha=0.5, beta=1.0)
    close(out, 0.5 * x @ w.T + c0 + bias, 2e-6 if od == "f32" else 6e-3, name=f"skinny gemm M={M}")


def test_gemm_structural_zero_hints(ops):
    """the two contractions over dT (zero above the causal diagonal) with the k-tile skipping hints: same results as the plain
    calls; the skipped region really is skipped (it holds NaNs wherever a whole 64-wide k-tile lies above the diagonal)"""
    rng = np.random.default_rng(17)
    H, B, L, D = 2, 3, 512, 128
    i = np.arange(L)[:, None]; dd = np.arange(L)[None, :]
    dT = bf(rng.standard_normal((H, B, L, L))) * (dd <= i)
    R = bf(rng.standard_normal((L, H, D)))
    qv = bf(rng.standard_normal((B, L, H, D)))
    poisoned = dT.copy()
    tile_first_row = (i // 256) * 256
    poisoned[..., (dd // 64) * 64 > tile_first_row + 255] = np.nan      # mode 1 never touches k-tiles right of the 256-row tile
    dTd, dTp = dev16(dT), dev16(poisoned)
    Rd, qvd = dev16(R), dev16(qv)
    ref_dq = np.einsum("hbik,khd->bihd", dT, R)
    ref_dR = np.einsum("hbik,bihd->khd", dT, qv)
    for src, name in ((dTd, "zeros"), (dTp, "poisoned")):
        dqv = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
        ops.gemm_batched(src, Rd.view(L, H, D).permute(1, 0, 2).unsqueeze(1).expand(H, B, L, D), dqv.permute(2, 0, 1, 3), tri=(1, 0))
        close(dqv, ref_dq, 6e-3, name=f"dq_r tri ({name})")
    poisoned2 = dT.copy()
    dist_tile_first = (dd // 256) * 256
    poisoned2[..., (i // 64) * 64 + 63 < dist_tile_first] = np.nan        # mode 2 never touches query k-tiles above the 256-distance tile
    for src, name in ((dTd, "zeros"), (dev16(poisoned2), "poisoned")):
        dR = torch.empty(L, H * D, device=DEV, dtype=torch.bfloat16)
        ops.gemm_batched(src.view(H, B * L, L).transpose(1, 2).unsqueeze(1), qvd.view(B * L, H, D).permute(1, 0, 2).unsqueeze(1),
                         dR.view(L, H, D).permute(1, 0, 2).unsqueeze(1), tri=(2, L))
        close(dR.view(L, H, D), ref_dR, 6e-3, name=f"dR tri ({name})")


def test_gemm_structural_zero_hint_by_period_at_model_length(ops):
    """dR = dT^T . (q + v) per head at L = 1024 (the model's length): 8 batches = 8 periods of k, split-K over the periods, on the
    256 x 128 form of the 4-wave kernel, whose loop walks each period from the tile's first row on and steps over the rest (the k-tiles
    above the diagonal hold NaNs here: they must never be read); two runs agree bit for bit"""
    rng = np.random.default_rng(23)
    H, B, L, D = 2, 8, 1024, 128
    i = np.arange(L)[:, None]; dd = np.arange(L)[None, :]
    dT = (bf(rng.standard_normal((H, B, L, L))) * (dd <= i)).astype(np.float32)
    qv = bf(rng.standard_normal((B, L, H, D))).astype(np.float32)
    ref = np.einsum("hbik,bihd->khd", dT, qv, optimize=True)
    poisoned = dT.copy()
    poisoned[..., (i // 64) * 64 + 63 < (dd // 256) * 256] = np.nan
    qvd = dev16(qv)
    outs = []
    for src in (dev16(dT), dev16(poisoned), dev16(poisoned)):
        dR = torch.full((L, H * D), float("nan"), device=DEV, dtype=torch.bfloat16)
        ops.gemm_batched(src.view(H, B * L, L).transpose(1, 2).unsqueeze(1), qvd.view(B * L, H, D).permute(1, 0, 2).unsqueeze(1),
                         dR.view(L, H, D).permute(1, 0, 2).unsqueeze(1), tri=(2, L))
        close(dR.view(L, H, D), ref, 6e-3, name="dR by period")
        outs.append(dR)
    assert torch.equal(outs[1], outs[2])


def test_gemm_structural_zero_hint_heavy_rows_first(ops):
    """the per-head dR with its tile rows handed out heaviest first (the 4-wave kernel's walk for hint 2 on one column of tiles: the (row,
    batch x slice) pair of a workgroup comes from its linear id, rows first): a bijection, so the same partial sums in the same slices as
    the plain walk -- bit-identical outputs with the walk on (knob tri_split 1 / 2) and off (0) at equal slice counts, NaNs above the
    diagonal never read, against the fp32 product; and the 8-slice rule (16 heads x 16 sequences: 64 tiles, 512 workgroups) against the
    4-slice one to fp32 round-off of the partial sums"""
    from bdm_db1_amd import lib as db1lib
    rng = np.random.default_rng(29)
    L, D = 1024, 128
    i = np.arange(L)[:, None]; dd = np.arange(L)[None, :]

    def run(H, B, dTd, qvd, knob, dtype):
        db1lib.set_knob("tri_split", knob)
        ops._ws_query_cache.clear()          # (the workspace query depends on the knob)
        try:
            dR = torch.full((L, H * D), float("nan"), device=DEV, dtype=dtype)
            ops.gemm_batched(dTd.view(H, B * L, L).transpose(1, 2).unsqueeze(1), qvd.view(B * L, H, D).permute(1, 0, 2).unsqueeze(1),
                             dR.view(L, H, D).permute(1, 0, 2).unsqueeze(1), tri=(2, L))
            return dR
        finally:
            db1lib.load().db1_test_clear_knobs()
            ops._ws_query_cache.clear()

    H, B = 3, 8                              # 12 tiles x 8 slices either way (the plain rule takes 8 slices up to 36 tiles)
    dT = (bf(rng.standard_normal((H, B, L, L))) * (dd <= i)).astype(np.float32)
    qv = bf(rng.standard_normal((B, L, H, D))).astype(np.float32)
    ref = np.einsum("hbik,bihd->khd", dT, qv, optimize=True)
    dT[..., (i // 64) * 64 + 63 < (dd // 256) * 256] = np.nan
    dTd, qvd = dev16(dT), dev16(qv)
    outs = [run(H, B, dTd, qvd, k, torch.float32) for k in (0, 1, 2)]
    close(outs[0].view(L, H, D), ref, 2e-5, name="dR, plain walk")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "the heavy-first walk changed the result"

    H, B = 16, 16                            # the model's heads: 64 tiles; 8 slices of 2 sequences (walk) against 4 slices of 4
    g = torch.Generator(device=DEV); g.manual_seed(5)
    dTd = torch.randn(H, B, L, L, device=DEV, generator=g).to(torch.bfloat16)
    dTd.masked_fill_(torch.from_numpy((dd > i)).to(DEV), 0)
    dTd.masked_fill_(torch.from_numpy(((i // 64) * 64 + 63 < (dd // 256) * 256)).to(DEV), float("nan"))
    qvd = torch.randn(B, L, H, D, device=DEV, generator=g).to(torch.bfloat16)
    a, b_, c = (run(H, B, dTd, qvd, k, torch.float32) for k in (0, 1, 1))
    assert torch.isfinite(a).all() and torch.isfinite(b_).all()
    assert torch.equal(b_, c), "two runs of the 8-slice form differ"
    assert not torch.equal(a, b_), "the 8-slice rule did not apply (same bits as 4 slices)"
    err = (a - b_).abs().max().item() / a.abs().max().item()
    assert err < 5e-6, f"8 slices against 4: {err:.2e} of the largest value"


@pytest.mark.parametrize("case", ["weight_grad", "per_head_batched"])
def test_gemm_bf16_workspace_split_k(ops, case):
    """small outputs over a long contraction take the deterministic workspace split-K (partials + fixed-order reduce):
    a 2048x2048 weight gradient over K = 8192 (fp32, accumulating; 256x256 ping-pong kernel, 4 slices) and a batched 1024x128
    bf16 output (256x128 kernel, 4 slices);
    two runs must agree bit for bit"""
    rng = np.random.default_rng(11)
    if case == "weight_grad":
        M, N, K, nb = 2048, 2048, 8192, 1
    else:
        M, N, K, nb = 1024, 128, 8192, 2
    A = bf(rng.standard_normal((nb, M, K)) * 0.5)
    B = bf(rng.standard_normal((nb, K, N)) * 0.5)
    a_store = dev(np.ascontiguousarray(A.transpose(0, 2, 1)), torch.bfloat16)   # [nb, K, M]: M-major A
    b_store = dev(B, torch.bfloat16)                                             # [nb, K, N]: M-major B
    a, b = a_store.transpose(1, 2).unsqueeze(1), b_store.unsqueeze(1)
    if case == "weight_grad":
        C0 = rng.standard_normal((nb, M, N)).astype(np.float32)
        outs = []
        for _ in range(2):
            out = dev(C0, torch.float32)
            ops.gemm(a[0, 0], b[0, 0], out[0], beta=1.0)
            outs.append(out)
        close(outs[0], np.einsum("bmk,bkn->bmn", A, B) + C0, 3e-6, name="split-K dW")
    else:
        outs = []
        for _ in range(2):
            out = torch.empty(nb, 1, M, N, device=DEV, dtype=torch.bfloat16)
            ops.gemm_batched(a, b, out)
            outs.append(out)
        close(outs[0][:, 0], np.einsum("bmk,bkn->bmn", A, B), 6e-3, name="split-K batched")
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("form,M,N,K", [("nt", 200, 72, 128), ("nn", 328, 576, 64), ("tn", 72, 200, 192), ("nt", 1568, 2048, 256), ("tn", 576, 64, 512)])
def test_gemm_bf16_tile_tails(ops, form, M, N, K):
    """ragged M / N (image-patch counts, 64-channel convolutions) stay on the MFMA tile kernel: clamped loads, masked stores"""
    from bdm_db1_amd import lib
    rng = np.random.default_rng(M + N + K)
    A = bf(rng.standard_normal((M, K)))
    B = bf(rng.standard_normal((K, N)))
    a_store = dev(A if form != "tn" else A.T.copy(), torch.bfloat16)
    b_store = dev(B.T.copy() if form == "nt" else B, torch.bfloat16)
    a = a_store if form != "tn" else a_store.t()
    b = b_store.t() if form == "nt" else b_store
    canvas = torch.full((M + 3, N + 8), 5.0, device=DEV, dtype=torch.float32)  # guard band around the output
    out = canvas[:M, :N]
    assert lib.load().db1_gemm_would_use_fast(M, N, K, 1, 1, 0, a.stride(0), a.stride(1), b.stride(0), b.stride(1), out.stride(0), out.stride(1)) == 1
    ops.gemm(a, b, out)
    close(out, A @ B, 3e-6, name=f"tail gemm {form}")
    assert float(canvas[M:].min()) == 5.0 and float(canvas[:, N:].min()) == 5.0  # nothing written outside [M, N]


def test_gemm_bf16_split_k_weight_gradient(ops):
    """64 x 576 output over a 32 768-row contraction (patch-conv weight gradient): split-K with fp32 atomics, beta = 1"""
    rng = np.random.default_rng(21)
    M, N, K = 64, 576, 32768
    A = bf(rng.standard_normal((M, K)) * 0.1)
    B = bf(rng.standard_normal((K, N)) * 0.1)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    a = dev(A.T.copy(), torch.bfloat16).t()
    out = dev(C0)
    ops.gemm(a, dev(B, torch.bfloat16), out, beta=1.0)
    close(out, A @ B + C0, 2e-5, name="split-K TN")
    # padded im2col columns (27 -> 32) are zero
    x = rng.standard_normal((3, 3, 16, 16))
    cols = torch.full((3 * 256, 32), 9.0, device=DEV)
    ops.im2col3x3(dev(x), cols, 3, 3, 16)
    ref = np.zeros((3 * 256, 32))
    ref[:, :27] = O._im2col3x3(x).reshape(3 * 256, 27)
    close(cols, ref, 1e-7, name="padded im2col")


def test_gemm_bf16_short_last_wave_and_half_wave_split(ops):
    """dispatcher paths of the weight gradients at 65 536 tokens: (a) 264 tiles of 256x256 = one full wave + one tile row, the
    remainder rows go through a second call on the split-K path; (b) 128 tiles in two K slices; (c) 192 tiles in four K slices.
    Reference: fp32 matmul of the same bf16 operands on the device (sampled rows incl. the remainder rows)."""
    g = torch.Generator(device="cpu").manual_seed(5)
    for M, N, K in ((33 * 256, 2048, 16384), (2048, 4096, 32768), (6144, 2048, 32768)):
        a_t = (torch.randn(K, M, generator=g) * 0.1).to(torch.bfloat16).to(DEV)   # TN: A is stored [K, M]
        b = (torch.randn(K, N, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
        c0 = torch.randn(M, N, generator=g).to(DEV)
        out = c0.clone()
        ops.gemm(a_t.t(), b, out, beta=1.0)
        rows = torch.cat([torch.arange(0, 64), torch.arange(M // 2, M // 2 + 64), torch.arange(M - 300, M)]).to(DEV)
        ref = a_t.t()[rows].float() @ b.float() + c0[rows]
        err = float((out[rows] - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, f"M={M} N={N} K={K}: {err:.3e}"


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
def test_gemm_4wave_loops_are_bitwise_repeatable(ops, form):
    """the hand-scheduled 4-wave loops reuse two LDS stages under counted waits and barriers of their own: a stage overwritten too early
    or read too early would show as run-to-run differences.  20 launches of an 8-rounds-per-CU product must agree bit for bit
    (and with an fp32 matmul to bf16 accuracy)."""
    g = torch.Generator(device="cpu").manual_seed(11)
    M, N, K = 8192, 4096, 1280   # 512 tiles of 256x256, 20 k-tiles
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    B = (torch.randn(K, N, generator=g) * 0.1).to(torch.bfloat16)
    a_store = (A if form != "tn" else A.t().contiguous()).to(DEV)
    b_store = (B.t().contiguous() if form == "nt" else B).to(DEV)
    a = a_store if form != "tn" else a_store.t()
    b = b_store.t() if form == "nt" else b_store
    first = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(a, b, first)
    ref = A[:512].to(DEV).float() @ B.to(DEV).float()
    assert float((first[:512].float() - ref).abs().max() / ref.abs().max()) < 6e-3
    out = torch.empty_like(first)
    for _ in range(20):
        out.fill_(1.0)
        ops.gemm(a, b, out)
        assert torch.equal(out, first)


@pytest.mark.parametrize("M,d,K", [(4096 * 4, 1024, 512), (4096, 2048, 512)])
def test_gemm_nt_head_bias_epilogue(ops, M, d, K):
    """the attention input projection with q + r_w_bias / q + r_r_bias written from the accumulators (db1_gemm_nt_headbias): 64 x 12 tiles of
    256 x 256, and the reference's micro-batch (4096 rows x 6144 columns: 1.5 rounds of 256 x 256 tiles -> the 256 x 128 form, round 5)"""
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(3 * d, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    u = torch.randn(d, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(d, generator=g).to(torch.bfloat16).to(DEV)
    assert ops.gemm_nt_headbias_supported(M, 3 * d, K, d)
    out = torch.full((M, 3 * d), 5.0, device=DEV, dtype=torch.bfloat16)
    qu = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    qv = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt_headbias(x, w, out, qu, qv, u, v, d)
    ref = x.float() @ w.float().t()
    sc = float(ref.abs().max())
    assert float((out[:, d:].float() - ref[:, d:]).abs().max()) / sc < 6e-3
    assert float((qu.float() - (ref[:, :d] + u.float())).abs().max()) / sc < 6e-3
    assert float((qv.float() - (ref[:, :d] + v.float())).abs().max()) / sc < 6e-3
    assert bool((out[:, :d] == 5.0).all()), "the query columns of the packed output are not written"


@pytest.mark.parametrize("M,N,K,with_bias", [(4352, 4096, 512, True), (8192, 8448, 128, False), (2560 * 4, 2048 * 4, 256, True),
                                                (2048, 2304, 384, False), (1024, 1024, 2048, True), (768, 512, 640, False)])
def test_gemm_bf16_nt_more_tiles_than_cus(ops, M, N, K, with_bias):
    """NT, bf16 output, more than 256 tiles of 256x256 (several rounds of workgroups per CU, the LDS-staged epilogue with and without
    bias) and the k-tile counts of the hand-scheduled 4-wave loop (K a multiple of 128 from 256: 2 peeled k-tiles + pairs; the
    per-XCD k rotation wraps inside the loop); sampled rows against an fp32 matmul of the same operands"""
    g = torch.Generator(device="cpu").manual_seed(M + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(DEV) if with_bias else None
    y = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
    ops.gemm(x, w.t(), y, bias=b, alpha=0.5)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2 - 150, M // 2 + 150), torch.arange(M - 300, M)]).to(DEV) if M > 1200 else torch.arange(M).to(DEV)
    ref = 0.5 * (x[rows].float() @ w.float().t()) + (b.float() if with_bias else 0.0)
    err = float((y[rows].float() - ref).abs().max() / ref.abs().max())
    assert err < 6e-3, err
    # every tile written: no 3.0 left (a value the product does not produce on whole rows)
    assert not bool((y == 3.0).all(dim=1).any())


def test_gemm_bf16_tile_large_k_and_batch(ops):
    rng = np.random.default_rng(4)
    Z, M, N, K = 3, 128, 128, 1024
    A = bf(rng.standard_normal((Z, M, K)) * 0.1)
    B = bf(rng.standard_normal((Z, K, N)) * 0.1)
    a = dev(A, torch.bfloat16).unsqueeze(0)
    b = dev(B, torch.bfloat16).unsqueeze(0)
    out = torch.zeros(1, Z, M, N, device=DEV)
    ops.gemm_batched(a, b, out)
    close(out, np.einsum("zmk,zkn->zmn", A, B)[None], 3e-6, name="tile gemm batched")


# ------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dtype,d", [("f32", 128), ("f32", 2048), ("bf16", 64), ("bf16", 2048)])
def test_layernorm_residual(ops, dtype, d):
    rng = np.random.default_rng(5)
    rows, alpha, eps = 37, 1.3, 1e-5
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    rnd = (lambda a: a) if dtype == "f32" else bf
    x, r = rnd(rng.standard_normal((rows, d))), rnd(rng.standard_normal((rows, d)) + 0.5)
    gamma, beta = rnd(1 + 0.1 * rng.standard_normal(d)), rnd(0.1 * rng.standard_normal(d))
    dy = rnd(rng.standard_normal((rows, d)))
    s = rnd(alpha * x + r)
    y_ref, cache = O.layernorm_fwd(s, gamma, beta, eps)
    ds_ref, dg_ref, db_ref = O.layernorm_bwd(dy, gamma, cache)
    X, R, G, Bt = dev(x, td), dev(r, td), dev(gamma, td), dev(beta, td)
    y, s_out = torch.empty_like(X), torch.empty_like(X)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    ops.layernorm_residual_fwd(X, R, alpha, G, Bt, y, s_out, mean, rstd, eps)
    tol = 3e-6 if dtype == "f32" else 6e-3
    close(y, y_ref, tol, name="ln y")
    close(s_out, s, 1e-6 if dtype == "f32" else 4e-3, name="ln s")
    close(mean, s_out.to(torch.float64).cpu().numpy().mean(-1), 1e-5, name="ln mean")
    ds = torch.empty_like(X)
    dg, db = torch.ones(d, device=DEV), torch.ones(d, device=DEV)  # accumulate on top of ones
    ops.layernorm_residual_bwd(dev(dy, td), s_out, G, mean, rstd, ds, dg, db)
    close(ds, ds_ref, 1e-5 if dtype == "f32" else 8e-3, name="ln ds")
    close(dg, dg_ref + 1, 1e-5 if dtype == "f32" else 3e-3, name="ln dgamma")
    close(db, db_ref + 1, 1e-5 if dtype == "f32" else 3e-3, name="ln dbeta")
    # plain LN (r = NULL, no s_out)
    y2 = torch.empty_like(X)
    ops.layernorm_residual_fwd(X, None, 1.0, G, Bt, y2, None, mean, rstd, eps)
    close(y2, O.layernorm_fwd(x, gamma, beta, eps)[0], tol, name="ln plain")


@pytest.mark.parametrize("act", ["geglu", "gelu", "relu"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_ffn_activation(ops, act, dtype):
    rng = np.random.default_rng(6)
    rows, n = 19, 264
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    rnd = (lambda a: a) if dtype == "f32" else bf
    z = rnd(rng.standard_normal((rows, 2 * n if act == "geglu" else n)) * 1.5)
    dout = rnd(rng.standard_normal((rows, n)))
    if act == "geglu":
        a, b = z[:, :n], z[:, n:]
        ref = a * O.gelu(b)
        dref = np.concatenate([dout * O.gelu(b), dout * a * O.gelu_grad(b)], -1)
    elif act == "gelu":
        ref, dref = O.gelu(z), dout * O.gelu_grad(z)
    else:
        ref, dref = np.maximum(z, 0), dout * (z > 0)
    Z = dev(z, td)
    out = torch.empty(rows, n, device=DEV, dtype=td)
    ops.ffn_act_fwd(Z, out, act)
    dz = torch.empty_like(Z)
    ops.ffn_act_bwd(Z, dev(dout, td), dz, act)
    tol = 2e-6 if dtype == "f32" else 6e-3
    close(out, ref, tol, name="act fwd")
    close(dz, dref, tol, name="act bwd")
    # fused variant: same dz, plus the column sums of what it stored, accumulated into an existing vector
    dz2 = torch.empty_like(Z)
    acc = torch.full((dz2.shape[-1],), 0.5, device=DEV, dtype=torch.float32)
    ops.ffn_act_bwd_bias(Z, dev(dout, td), dz2, acc, act)
    close(dz2, dref, tol, name="act bwd (fused)")
    close(acc, dz2.double().sum(0).cpu().numpy() + 0.5, 1e-5, name="act bwd bias sums")


def test_colsum_add_cast(ops):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((300, 70))
    acc = torch.full((70,), 2.0, device=DEV)
    ops.colsum_acc(dev(x), acc)
    close(acc, x.sum(0) + 2, 1e-6, name="colsum")
    xb = dev(bf(x), torch.bfloat16)
    acc2 = torch.zeros(70, device=DEV)
    ops.colsum_acc(xb[:, :64], acc2[:64])  # strided rows
    close(acc2[:64], bf(x)[:, :64].sum(0), 1e-6, name="colsum bf16 strided")
    xv = rng.standard_normal((1000, 264))  # 16-byte column groups: the vectorised kernel
    for td, ref in ((torch.float32, xv), (torch.bfloat16, bf(xv))):
        accv = torch.full((264,), -1.0, device=DEV)
        ops.colsum_acc(dev(ref, td), accv)
        close(accv, ref.sum(0) - 1, 2e-6, name=f"colsum vec {td}")
    a, b = dev(x), dev(x * 2)
    y = torch.empty_like(a)
    ops.add(a, b, y)
    close(y, 3 * x, 1e-7, name="add")
    c = torch.empty(300, 70, device=DEV, dtype=torch.bfloat16)
    ops.cast(a, c)
    assert torch.equal(c, a.to(torch.bfloat16))
    # fused: y = a + b (y aliasing the strided b) with both column sums, as used for dq = dq_k + dq_r, du, dv_bias
    for td, rows, cols in ((torch.float32, 1000, 64), (torch.bfloat16, 2500, 256)):
        ra = rng.standard_normal((rows, cols))
        rb_ = rng.standard_normal((rows, 3 * cols))
        if td == torch.bfloat16:
            ra, rb_ = bf(ra), bf(rb_)
        a_t, b_t = dev(ra, td), dev(rb_, td)
        sa, sb = torch.full((cols,), 3.0, device=DEV), torch.full((cols,), -2.0, device=DEV)
        ops.add2d_colsums(a_t, b_t[:, :cols], b_t[:, :cols], sa, sb)
        close(sa, ra.sum(0) + 3, 2e-6, name=f"add2d_colsums sum a {td}")
        close(sb, rb_[:, :cols].sum(0) - 2, 2e-6, name=f"add2d_colsums sum b {td}")
        close(b_t[:, :cols], ra + rb_[:, :cols], 1e-6 if td == torch.float32 else 4e-3, name=f"add2d_colsums y {td}")
        assert torch.equal(b_t[:, cols:], dev(rb_, td)[:, cols:]), "add2d_colsums must leave the other columns alone"


# ------------------------------------------------------------------------------- embeddings
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_embedding_gather_scatter(ops, dtype):
    rng = np.random.default_rng(8)
    V, d, n = 50, 72, 41
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    table = bf(rng.standard_normal((V, d)))
    ids = rng.integers(-1, V, n)
    ids[3] = -1
    out = torch.full((n, d), 7.0, device=DEV, dtype=td)
    ops.embed_gather(dev(table, td), torch.from_numpy(ids).to(DEV), out)
    ref = np.where(ids[:, None] >= 0, table[np.maximum(ids, 0)], 0.0)
    close(out, ref, 1e-7, name="gather")
    dout = bf(rng.standard_normal((n, d)))
    acc = torch.zeros(V, d, device=DEV)
    ops.embed_scatter_add(dev(dout, td), torch.from_numpy(ids).to(DEV), acc)
    g = np.zeros((V, d))
    np.add.at(g, ids[ids >= 0], dout[ids >= 0])
    close(acc, g, 1e-6, name="scatter")
    # deterministic (sorted runs, no float atomics): many tokens per row, two runs, same bits; accumulates onto what is there
    n2 = 20000
    ids2 = torch.from_numpy(rng.integers(-1, 7, n2)).to(DEV)              # ~2 500 tokens per table row
    dout2 = dev(rng.standard_normal((n2, d)), td)
    acc_a, acc_b = torch.full((V, d), 0.5, device=DEV), torch.full((V, d), 0.5, device=DEV)
    ops.embed_scatter_add(dout2, ids2, acc_a)
    ops.embed_scatter_add(dout2, ids2, acc_b)
    assert torch.equal(acc_a, acc_b)
    ref2 = np.full((V, d), 0.5)
    idn = ids2.cpu().numpy()
    np.add.at(ref2, idn[idn >= 0], dout2.double().cpu().numpy()[idn >= 0])
    close(acc_a, ref2, 2e-5, name="scatter long runs")


def test_out_of_table_ids_and_labels_touch_nothing(ops):
    """ids beyond the table / position ids beyond 513 rows / labels outside [0, V) (torch's ignore_index -100): zeros in, nothing out,
    no out-of-bounds access (the reference raises an IndexError there; a device kernel must not read past its tables)"""
    rng = np.random.default_rng(81)
    V, d, n = 20, 32, 12
    table = rng.standard_normal((V, d))
    ids = np.array([0, 19, 20, 1 << 40, -1, -100, 5, 7, 19, 3, 25, 2], np.int64)
    out = torch.full((n, d), 7.0, device=DEV)
    ops.embed_gather(dev(table), torch.from_numpy(ids).to(DEV), out)
    ok = (ids >= 0) & (ids < V)
    close(out, np.where(ok[:, None], table[np.where(ok, ids, 0)], 0.0), 1e-7, name="gather oob")
    acc = torch.zeros(V + 4, d, device=DEV)
    ops.embed_scatter_add(dev(rng.standard_normal((n, d))), torch.from_numpy(ids).to(DEV), acc[:V])
    assert float(acc[V:].abs().max()) == 0.0
    # RL assembly with a position id of 513 (table has 513 rows) and a word id == V
    B, L = 1, 40
    rid = rng.integers(0, V, (B, L)); rid[0, 3] = V
    pos = rng.integers(0, 513, (B, L)); pos[0, 5] = 513; pos[0, 6] = -2
    word, post = rng.standard_normal((V, d)), rng.standard_normal((513, d))
    o = torch.empty(B, L, d, device=DEV)
    ops.rl_assemble_fwd(dev(word), dev(post), None, torch.from_numpy(rid).to(DEV), torch.from_numpy(pos).to(DEV), None, o)
    ref = np.where((rid < V)[..., None], word[np.minimum(rid, V - 1)], 0.0) + np.where(((pos >= 0) & (pos < 513))[..., None], post[np.clip(pos, 0, 512)], 0.0)
    close(o, ref, 1e-6, name="rl oob")
    # masked CE: labels -100 / V add no loss and get no gradient; the normaliser stays sum(mask), as in the reference, where
    # CrossEntropyLoss(reduction="none") returns 0 for an ignored label and the loss is (loss * mask).sum() / mask.sum()
    T, ld = 6, 24
    logits = rng.standard_normal((T, ld)).astype(np.float32)
    lab = np.array([1, -100, V - 1, V, 0, 3], np.int64)
    msk = np.ones(T, np.float32)
    lse, sums = torch.zeros(T, device=DEV), torch.zeros(2, device=DEV)
    ops.masked_ce_fwd(dev(logits), torch.from_numpy(lab).to(DEV), dev(msk), lse, sums, V)
    good = (lab >= 0) & (lab < V)
    lz = logits[:, :V].astype(np.float64)
    ref_lse = np.log(np.exp(lz - lz.max(1, keepdims=True)).sum(1)) + lz.max(1)
    nll = np.where(good, ref_lse - lz[np.arange(T), np.where(good, lab, 0)], 0.0)
    assert abs(float(sums[0]) - nll.sum()) < 1e-4 and float(sums[1]) == T
    dl = torch.empty(T, ld, device=DEV)
    ops.masked_ce_bwd(dev(logits), torch.from_numpy(lab).to(DEV), dev(msk), lse, sums, dl, V)
    assert float(dl[~torch.from_numpy(good).to(DEV)].abs().max()) == 0.0 and float(dl[0].abs().max()) > 0


def test_rl_assemble(ops):
    rng = np.random.default_rng(9)
    B, L, d, V, nvis = 3, 300, 40, 60, 9
    ids = rng.integers(0, V, (B, L))
    for b in range(B):
        ids[b, rng.choice(L, nvis - b, replace=False)] = -1  # ragged placeholder counts
    pos = rng.integers(0, 513, (B, L))
    word, post = rng.standard_normal((V, d)), rng.standard_normal((513, d))
    vis = rng.standard_normal((B, nvis, d))
    labels = rng.integers(-1, V, (B, L))
    out = torch.empty(B, L, d, device=DEV)
    lab_d = torch.from_numpy(labels).to(DEV)
    ops.rl_assemble_fwd(dev(word), dev(post), dev(vis), torch.from_numpy(ids).to(DEV), torch.from_numpy(pos).to(DEV), lab_d, out)
    ref = np.zeros((B, L, d))
    for b in range(B):
        k = 0
        for t in range(L):
            if ids[b, t] >= 0:
                ref[b, t] = word[ids[b, t]]
            else:
                ref[b, t] = vis[b, k]
                k += 1
    ref += post[pos]
    close(out, ref, 1e-6, name="rl fwd")
    assert np.array_equal(lab_d.cpu().numpy(), np.where(labels == -1, 0, labels))
    dout = rng.standard_normal((B, L, d))
    dw, dp = torch.zeros(V, d, device=DEV), torch.zeros(513, d, device=DEV)
    dvis = torch.full((B, nvis, d), 5.0, device=DEV)
    ops.rl_assemble_bwd(dev(dout), torch.from_numpy(ids).to(DEV), torch.from_numpy(pos).to(DEV), dw, dp, dvis)
    gw, gp, gv = np.zeros((V, d)), np.zeros((513, d)), np.zeros((B, nvis, d))
    np.add.at(gp, pos.reshape(-1), dout.reshape(-1, d))
    for b in range(B):
        k = 0
        for t in range(L):
            if ids[b, t] >= 0:
                gw[ids[b, t]] += dout[b, t]
            else:
                gv[b, k] = dout[b, t]
                k += 1
    close(dw, gw, 1e-5, name="rl dword")
    close(dp, gp, 1e-5, name="rl dpos")
    close(dvis, gv, 1e-6, name="rl dvis")


# ------------------------------------------------------------------------------- cross-entropy
@pytest.mark.parametrize("dtype,V,ld", [("f32", 1325, 1325), ("f32", 1000, 1024), ("bf16", 33025, 33280)])
def test_masked_ce(ops, dtype, V, ld):
    rng = np.random.default_rng(10)
    T = 24
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    rnd = (lambda a: a) if dtype == "f32" else bf
    lg = rnd(rng.standard_normal((T, V)) * 3)
    labels = rng.integers(0, V, T)
    mask = (rng.random(T) > 0.3).astype(np.float32)
    mask[0] = 1
    buf = torch.full((T, ld), 9.0, device=DEV, dtype=td)
    buf[:, :V] = dev(lg, td)
    lse, sums = torch.empty(T, device=DEV), torch.zeros(2, device=DEV)
    ops.masked_ce_fwd(buf, torch.from_numpy(labels).to(DEV), dev(mask), lse, sums, V)
    mx = lg.max(-1)
    lse_ref = mx + np.log(np.exp(lg - mx[:, None]).sum(-1))
    nll = lse_ref - lg[np.arange(T), labels]
    close(lse, lse_ref, 2e-6, name="lse")
    close(sums, [(nll * mask).sum(), mask.sum()], 2e-6, name="ce sums")
    dl = torch.empty_like(buf)
    ops.masked_ce_bwd(buf, torch.from_numpy(labels).to(DEV), dev(mask), lse, sums, dl, V, gscale=1.0)
    p = np.exp(lg - lse_ref[:, None])
    p[np.arange(T), labels] -= 1
    ref = np.zeros((T, ld))
    ref[:, :V] = p * (mask / mask.sum())[:, None]
    close(dl, ref, 3e-6 if dtype == "f32" else 6e-3, name="dlogits")
    assert float(dl[:, V:].abs().max()) == 0.0 if ld > V else True
    # both in one pass over the rows (the row stays in registers; what the chunked head + loss sweep runs): bit-equal to the pair above
    if ops.lib.load().db1_masked_ce_fwd_bwd_supported(V, ld, ops.dt_code(buf)):
        buf2, lse2, sums2 = buf.clone(), torch.empty(T, device=DEV), torch.zeros(2, device=DEV)
        ops.masked_ce_fwd_bwd(buf2, torch.from_numpy(labels).to(DEV), dev(mask), lse2, sums2, sums, V, gscale=1.0)
        assert torch.equal(lse2, lse) and torch.equal(sums2, sums) and torch.equal(buf2, dl)
    else:
        assert dtype == "f32" and ld > 17408 or dtype == "f32"


# ------------------------------------------------------------------------------- optimizer
@pytest.mark.parametrize("adamw", [False, True])
def test_adam_and_global_norm(ops, adamw):
    rng = np.random.default_rng(11)
    n = 4096 + 8
    p = rng.standard_normal(n).astype(np.float32)
    P32, M_, V_ = dev(p), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    PW = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    pr, mr, vr = p.astype(np.float64), np.zeros(n), np.zeros(n)
    for step in range(1, 4):
        g = (rng.standard_normal(n) * (0.1 if step == 2 else 2.0)).astype(np.float32)
        G = dev(g)
        nsq = torch.zeros(1, device=DEV)
        ops.sumsq_acc(G, nsq)
        close(nsq, [(g.astype(np.float64) ** 2).sum()], 1e-5, name="sumsq")
        if step == 1:   # the one-workgroup form refuses vectors it would take milliseconds for (callers are pointed at db1_grad_norm_sq)
            from bdm_db1_amd import lib as db1lib
            big = torch.empty((1 << 24) + 8, device=DEV)
            with pytest.raises(db1lib.Db1Error, match="db1_sumsq_det"):
                ops.sumsq_acc(big, nsq)
            del big
        ops.adam_step(P32, G, M_, V_, PW, 3e-3, 0.9, 0.999, 1e-8, 0.01, adamw, step, gscale=1.0, clip=1.0, norm_sq=nsq)
        coef = O.clip_coef(np.sqrt((g.astype(np.float64) ** 2).sum()), 1.0)
        pr, mr, vr = O.adam_step(pr, g.astype(np.float64), mr, vr, step, 3e-3, wd=0.01, adamw=adamw, grad_scale=coef)
        close(P32, pr, 2e-6, name=f"adam p step {step}")
        close(M_, mr, 2e-6, name="adam m")
        close(V_, vr, 2e-6, name="adam v")
        assert torch.equal(PW, P32.to(torch.bfloat16))


def test_adam_matches_torch_golden(ops):
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "adam.npz")))
    for mode in ("adam", "adamw"):
        n = 260  # fixture has 257 elements; pad to a multiple of 4
        pad = lambda a: np.concatenate([a, np.zeros(n - a.size, np.float32)])
        P32, M_, V_ = dev(pad(gold[f"{mode}/p0"])), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for i in range(3):
            G = dev(pad(gold[f"{mode}/g{i}"]))
            nsq = torch.zeros(1, device=DEV)
            ops.sumsq_acc(G, nsq)
            ops.adam_step(P32, G, M_, V_, None, 3e-3, 0.9, 0.999, 1e-8, 0.01, mode == "adamw", i + 1, clip=1.0, norm_sq=nsq)
            close(P32[:257], gold[f"{mode}/p{i + 1}"], 3e-6, name=f"{mode} p{i + 1}")
            close(V_[:257], gold[f"{mode}/v{i + 1}"], 3e-6, name=f"{mode} v{i + 1}")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_lmhead_ce_chunked_sweep(ops, dtype):
    """db1_lmhead_ce_fwd / _fwd_bwd (tied head + masked CE without the logits tensor; transformer_xl.py:593-613) against the closed form
    in float64: ragged last chunk (300 rows in chunks of 128), padded vocabulary rows, masked rows, beta on the weight gradient"""
    rng = np.random.default_rng(31)
    T, V, rows, d, chunk = 300, 1000, 1024, 64, 128
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    rt = lambda a: dev(a, td).double().cpu().numpy()
    h, W = rt(rng.standard_normal((T, d))), np.zeros((rows, d))
    W[:V] = rt(0.2 * rng.standard_normal((V, d)))
    lab = rng.integers(0, V, T)
    msk = (rng.random(T) > 0.3).astype(np.float32)
    logits = h @ W[:V].T
    mx = logits.max(1, keepdims=True)
    lse_ref = mx[:, 0] + np.log(np.exp(logits - mx).sum(1))
    nll = lse_ref - logits[np.arange(T), lab]
    gscale = 0.5
    dlog = (np.exp(logits - lse_ref[:, None]) - np.eye(V)[lab]) * (msk / msk.sum() * gscale)[:, None]
    dh_ref, dW_ref = dlog @ W[:V], dlog.T @ h
    tol = 2e-5 if dtype == "f32" else 2e-2
    hd, Wd, labd, mskd = dev(h, td), dev(W, td), torch.from_numpy(lab).to(DEV), dev(msk)
    # loss only
    lse, sums = torch.zeros(T, device=DEV), torch.tensor([1.0, 2.0], device=DEV)
    ops.lmhead_ce(hd, Wd, labd, mskd, lse, sums, V, chunk_rows=chunk)
    assert abs(float(sums[0]) - 1.0 - (nll * msk).sum()) < tol * (nll * msk).sum() and abs(float(sums[1]) - 2.0 - msk.sum()) < 1e-4
    close(lse, lse_ref, 1e-5 if dtype == "f32" else 5e-3, name="lse")
    # training sweep, accumulating onto an existing weight gradient
    lse2, sums2 = torch.zeros(T, device=DEV), torch.zeros(2, device=DEV)
    dh = torch.empty(T, d, device=DEV, dtype=td)
    prev = rng.standard_normal((rows, d))
    dW = dev(prev)
    ops.lmhead_ce(hd, Wd, labd, mskd, lse2, sums2, V, dh=dh, dW_acc=dW, beta_dw=1.0, gscale=gscale, chunk_rows=chunk)
    assert abs(float(sums2[0] / sums2[1]) - (nll * msk).sum() / msk.sum()) < tol * 10
    close(dh, dh_ref, tol, name="dh")
    got = dW.double().cpu().numpy() - prev
    assert np.abs(got[:V] - dW_ref).max() <= tol * np.abs(dW_ref).max() + 1e-6 and np.abs(got[V:]).max() <= 1e-6
    dW0 = dev(prev)
    ops.lmhead_ce(hd, Wd, labd, mskd, lse2, torch.zeros(2, device=DEV), V, dh=dh, dW_acc=dW0, beta_dw=0.0, gscale=gscale, chunk_rows=chunk)
    assert np.abs(dW0.double().cpu().numpy()[:V] - dW_ref).max() <= tol * np.abs(dW_ref).max() + 1e-6   # beta = 0: the old content is gone
    # one chunk (default size) gives the same numbers as three
    dW1, dh1 = torch.zeros(rows, d, device=DEV), torch.empty(T, d, device=DEV, dtype=td)
    ops.lmhead_ce(hd, Wd, labd, mskd, lse2, torch.zeros(2, device=DEV), V, dh=dh1, dW_acc=dW1, beta_dw=0.0, gscale=gscale)
    close(dh1, dh.double().cpu().numpy(), 1e-6 if dtype == "f32" else 1e-2, name="dh chunking")


# ------------------------------------------------------------------------------- tokenizer (bit-exact)
def test_mulaw_discretize_bit_exact(ops):
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "scalar_tokenizer.npz")))
    for key, is_action in (("known_obs", False), ("known_act", True), ("obs", False), ("act", True)):
        x = torch.from_numpy(gold[key]).to(DEV)
        ids = torch.empty(x.numel(), device=DEV, dtype=torch.int32)
        ops.mulaw_discretize(x, ids, is_action)
        assert np.array_equal(ids.cpu().numpy(), gold[key + "_ids"]), key
    # large seeded sweep against the oracle (itself pinned to the reference on 24M values)
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.standard_normal(500000) * s for s in (1e-3, 0.05, 1.0, 30.0, 500.0)]).astype(np.float32)
    ids = torch.empty(x.size, device=DEV, dtype=torch.int32)
    ops.mulaw_discretize(torch.from_numpy(x).to(DEV), ids, False)
    assert np.array_equal(ids.cpu().numpy(), O.mulaw_discretize(x, False))


def test_mulaw_decode_all_ids_and_tokenizer_class(ops, capsys):
    """db1_mulaw_decode over EVERY bin id against the reference's own decode output (scalar_tokenizer.npz: dec_obs / dec_act), and the
    product ContinuousScalarTokenizer (reference surface, scalar_tokenizer.py:20-63) on NumPy / CPU / device inputs"""
    from bdm_db1_amd.tokenizer import ContinuousScalarTokenizer
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "scalar_tokenizer.npz")))
    ids = torch.from_numpy(gold["dec_ids"]).to(DEV)
    assert ids.numel() == 1024
    for idt in (torch.int64, torch.int32):
        for key, is_action in (("dec_obs", False), ("dec_act", True)):
            out = torch.empty(ids.numel(), device=DEV, dtype=torch.float32)
            ops.mulaw_decode(ids.to(idt), out, is_action)
            # actions: pure float32 arithmetic -> bit-exact; observations: (base^|x| - 1) / mu with the float32 power within 2 ulp of
            # torch's (the kernel rounds a float64 pow once; torch-CPU's vectorised powf is itself ~1 ulp)
            if is_action:
                assert np.array_equal(out.cpu().numpy(), gold[key]), key
            else:
                tol = 2 * 2.0 ** -23 * (np.abs(gold[key].astype(np.float64)) * 100 + 1) / 100
                assert (np.abs(out.cpu().numpy().astype(np.float64) - gold[key]) <= tol).all(), key
    tok = ContinuousScalarTokenizer()
    assert (tok.num_continuous_bin, tok.mu, tok.M) == (1024, 100.0, 256.0)
    for src in (gold["known_obs"], torch.from_numpy(gold["known_obs"]), torch.from_numpy(gold["known_obs"]).to(DEV)):
        got = tok.discretize(src, is_action=False)
        assert got.dtype == torch.int32 and np.array_equal(got.cpu().numpy(), gold["known_obs_ids"])
        assert got.device.type == ("cuda" if torch.is_tensor(src) and src.is_cuda else "cpu")
    assert np.array_equal(tok.discretize(gold["act"], is_action=True).numpy(), gold["act_ids"])
    dec = tok.decode(gold["dec_ids"], is_action=True)
    assert dec.dtype == torch.float32 and dec.device.type == "cpu" and np.array_equal(dec.numpy(), gold["dec_act"])
    # discretize(decode(id)) is the identity on the action path (bin lower edges, exact float32 arithmetic both ways).  (Observations
    # decode to the mu-law image of a bin EDGE, where a 1-ulp difference of the power decides between bin id and id - 1: no identity
    # to assert there; the decoded values themselves are checked against the reference above.)
    rt = tok.discretize(tok.decode(ids, is_action=True), is_action=True)
    assert torch.equal(rt.cpu(), ids.cpu().to(torch.int32))
    rt_obs = tok.discretize(tok.decode(ids, is_action=False), is_action=False).cpu().numpy().astype(np.int64)
    assert np.abs(rt_obs - gold["dec_ids"]).max() <= 1
    # out-of-range ids: clipped, with the reference's warning (scalar_tokenizer.py:50-57)
    capsys.readouterr()
    clipped = tok.decode(np.array([-3, 0, 1023, 5000], np.int64), is_action=True).numpy()
    assert np.array_equal(clipped, gold["dec_act"][[0, 0, 1023, 1023]])
    assert "exceeded range" in capsys.readouterr().out


# ------------------------------------------------------------------------------- patch embedder pieces
def test_patch_normalize_im2col_groupnorm(ops):
    rng = np.random.default_rng(13)
    img = (rng.random((2, 3, 32, 48)) * 255).astype(np.float32)
    ref = O.patch_normalize(O.patchify(img.astype(np.float64), 16), 16)
    N = ref.shape[0]
    out = torch.empty(N, 3, 16, 16, device=DEV)
    ops.patch_normalize(dev(img), out, 16)
    close(out, ref, 3e-6, name="patch normalize")
    x = rng.standard_normal((N, 5, 16, 16))
    cols = torch.empty(N * 256, 45, device=DEV)
    ops.im2col3x3(dev(x), cols, N, 5, 16)
    close(cols, O._im2col3x3(x).reshape(N * 256, 45), 1e-7, name="im2col")
    dcols = rng.standard_normal((N, 16, 16, 45))
    dx = torch.empty(N, 5, 16, 16, device=DEV)
    ops.col2im3x3(dev(dcols.reshape(N * 256, 45)), dx, N, 5, 16)
    close(dx, O._col2im3x3(dcols, 5), 2e-6, name="col2im")
    # layout shuffles
    a = rng.standard_normal((N, 256, 64))
    y = torch.empty(N, 64, 256, device=DEV)
    ops.nhwc_to_nchw(dev(a), y, N, 64, 256)
    close(y, a.transpose(0, 2, 1), 1e-7, name="nhwc->nchw")
    z = torch.empty(N, 256, 64, device=DEV)
    ops.nchw_to_nhwc(y, z, N, 64, 256)
    close(z, a, 1e-7, name="nchw->nhwc")
    # GroupNorm + GELU
    xg = rng.standard_normal((N, 64, 16, 16)) * 2 + 0.3
    gam, bet = 1 + 0.2 * rng.standard_normal(64), 0.2 * rng.standard_normal(64)
    g_ref, cache = O.groupnorm_fwd(xg, gam, bet)
    yy = torch.empty(N, 64, 256, device=DEV)
    mean, rstd = torch.empty(N * 32, device=DEV), torch.empty(N * 32, device=DEV)
    ops.groupnorm_gelu_fwd(dev(xg), dev(gam), dev(bet), yy, mean, rstd, N, 64, 256)
    close(yy, O.gelu(g_ref).reshape(N, 64, 256), 3e-6, name="gn+gelu fwd")
    dy = rng.standard_normal((N, 64, 16, 16))
    dx_ref, dgam, dbet = O.groupnorm_bwd(dy * O.gelu_grad(g_ref), gam, cache)
    dxx = torch.empty(N, 64, 256, device=DEV)
    dga, dbe = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    ops.groupnorm_gelu_bwd(dev(dy), dev(xg), dev(gam), dev(bet), mean, rstd, dxx, dga, dbe, N, 64, 256)
    close(dxx, dx_ref.reshape(N, 64, 256), 1e-5, name="gn+gelu dx")
    close(dga, dgam, 1e-5, name="gn dgamma")
    close(dbe, dbet, 1e-5, name="gn dbeta")


def test_channels_last_vision_kernels(ops):
    """the bf16 channels-last embedder kernels against the oracle's NCHW statements (same maths, transposed layouts)"""
    rng = np.random.default_rng(29)
    img = (rng.random((2, 3, 32, 48)) * 255).astype(np.float32)
    ref = O.patch_normalize(O.patchify(img.astype(np.float64), 16), 16)            # [N, 3, 16, 16]
    N = ref.shape[0]
    out = torch.empty(N * 256, 3, device=DEV, dtype=torch.bfloat16)
    ops.patch_normalize_nhwc(dev(img), out, 16)
    close(out.view(N, 256, 3), ref.reshape(N, 3, 256).transpose(0, 2, 1), 6e-3, name="patch normalize (channels-last)")
    for C, kpad in ((64, 576), (3, 32)):
        x = bf(rng.standard_normal((N, C, 16, 16)))
        x_cl = dev16(x.reshape(N, C, 256).transpose(0, 2, 1))                      # [N, 256, C]
        cols = torch.full((N * 256, kpad), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.im2col3x3_nhwc(x_cl, cols, N, C, 16)
        ref_cols = O._im2col3x3(x).reshape(N * 256, C, 9).transpose(0, 2, 1).reshape(N * 256, 9 * C)  # (c, tap) -> (tap, c)
        got = cols.float().cpu().numpy()
        assert np.array_equal(got[:, :9 * C], ref_cols.astype(np.float32)) and (got[:, 9 * C:] == 0).all(), f"im2col C={C}"
        dcols = bf(rng.standard_normal((N * 256, kpad)))
        dx = torch.empty(N * 256, C, device=DEV, dtype=torch.bfloat16)
        ops.col2im3x3_nhwc(dev16(dcols), dx, N, C, 16)
        dc_ref = dcols[:, :9 * C].reshape(N, 16, 16, 9, C).transpose(0, 1, 2, 4, 3).reshape(N, 16, 16, C * 9)
        close(dx.view(N, 256, C), O._col2im3x3(dc_ref, C).reshape(N, C, 256).transpose(0, 2, 1), 6e-3, name=f"col2im C={C}")
        # weight operand and gradient un-permutation
        w = bf(rng.standard_normal((64, C, 3, 3)))
        wp = torch.empty(64, kpad, device=DEV, dtype=torch.bfloat16)
        ops.conv_weight_permute(dev16(w), wp, 64, C)
        wref = np.zeros((64, kpad)); wref[:, :9 * C] = w.reshape(64, C, 9).transpose(0, 2, 1).reshape(64, 9 * C)
        assert np.array_equal(wp.float().cpu().numpy(), wref.astype(np.float32))
        gp = rng.standard_normal((64, kpad)).astype(np.float32)
        g0 = rng.standard_normal((64, C, 3, 3)).astype(np.float32)
        g = dev(g0)
        ops.conv_wgrad_unpermute(dev(gp), g, 64, C)
        close(g, g0 + gp[:, :9 * C].reshape(64, 9, C).transpose(0, 2, 1).reshape(64, C, 3, 3), 1e-6, name="wgrad unpermute")
    # GroupNorm(32, 64) + GELU, channels-last
    xg = bf(rng.standard_normal((N, 64, 16, 16)) * 2 + 0.3)
    gam, bet = bf(1 + 0.2 * rng.standard_normal(64)), bf(0.2 * rng.standard_normal(64))
    g_ref, cache = O.groupnorm_fwd(xg, gam, bet)
    cl = lambda a: a.reshape(N, 64, 256).transpose(0, 2, 1)
    yy = torch.empty(N * 256, 64, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(N * 32, device=DEV), torch.empty(N * 32, device=DEV)
    ops.groupnorm_gelu_nhwc_fwd(dev16(cl(xg)), dev16(gam), dev16(bet), yy, mean, rstd, N, 64, 256)
    close(yy.view(N, 256, 64), cl(O.gelu(g_ref)), 6e-3, name="gn+gelu fwd (channels-last)")
    close(mean, xg.reshape(N * 32, -1).mean(1), 1e-5, name="gn mean")
    dy = bf(rng.standard_normal((N, 64, 16, 16)))
    dx_ref, dgam, dbet = O.groupnorm_bwd(dy * O.gelu_grad(g_ref), gam, cache)
    dxx = torch.empty(N * 256, 64, device=DEV, dtype=torch.bfloat16)
    dga, dbe = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    ops.groupnorm_gelu_nhwc_bwd(dev16(cl(dy)), dev16(cl(xg)), dev16(gam), dev16(bet), mean, rstd, dxx, dga, dbe, N, 64, 256)
    close(dxx.view(N, 256, 64), cl(dx_ref), 8e-3, name="gn+gelu dx (channels-last)")
    close(dga, dgam, 2e-4, name="gn dgamma (channels-last)")
    close(dbe, dbet, 2e-4, name="gn dbeta (channels-last)")
    # ... with the residual branch's gradient added in the same pass (fp32, one rounding): parameter gradients unchanged bit for bit
    resid = bf(rng.standard_normal((N, 64, 16, 16)))
    dxr = torch.full((N * 256, 64), float("nan"), device=DEV, dtype=torch.bfloat16)
    dga2, dbe2 = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    ops.groupnorm_gelu_nhwc_bwd(dev16(cl(dy)), dev16(cl(xg)), dev16(gam), dev16(bet), mean, rstd, dxr, dga2, dbe2, N, 64, 256, res=dev16(cl(resid)).view(N * 256, 64))
    close(dxr.view(N, 256, 64), cl(dx_ref + resid), 6e-3, name="gn+gelu dx + residual (channels-last)")
    assert torch.equal(dga, dga2) and torch.equal(dbe, dbe2)


def test_implicit_3x3_convolutions(ops):
    """implicit-GEMM conv forward / data gradient / weight gradient (no column matrix) against the oracle's im2col statement"""
    rng = np.random.default_rng(41)
    N, C = 5, 64
    x = bf(rng.standard_normal((N, C, 16, 16)))
    w = bf(rng.standard_normal((64, C, 3, 3)) * 0.1)
    bias = bf(rng.standard_normal(64))
    cl = lambda a: np.ascontiguousarray(a.reshape(a.shape[0], a.shape[1], 256).transpose(0, 2, 1))   # NCHW -> [N, 256, C]
    cols = O._im2col3x3(x).reshape(N * 256, C * 9)                        # columns (c, tap)
    y_ref = cols @ w.reshape(64, C * 9).T + bias                           # [N*256, 64]
    w_op = torch.empty(64, 576, device=DEV, dtype=torch.bfloat16)
    ops.conv_weight_permute(dev16(w), w_op, 64, C)
    X = dev16(cl(x)).view(N * 256, C)
    y = torch.empty(N * 256, 64, device=DEV, dtype=torch.bfloat16)
    ops.conv3x3_implicit_fwd(X, w_op, dev16(bias), y, N, sign=1)
    close(y, y_ref, 6e-3, name="implicit conv fwd")
    # data gradient: dx = col2im(dy . W)
    dy = bf(rng.standard_normal((N * 256, 64)))
    dcols = (dy @ w.reshape(64, C * 9)).reshape(N, 16, 16, C * 9)
    dx_ref = cl(O._col2im3x3(dcols, C)).reshape(N * 256, C)
    w_t = torch.empty(64, 576, device=DEV, dtype=torch.bfloat16)
    ops.conv_weight_permute_t(dev16(w), w_t, 64, C)
    dx = torch.empty(N * 256, C, device=DEV, dtype=torch.bfloat16)
    ops.conv3x3_implicit_fwd(dev16(dy), w_t, None, dx, N, sign=-1)
    close(dx, dx_ref, 6e-3, name="implicit conv dgrad")
    # weight gradient, accumulated
    gw_ref = (dy.T @ cols).reshape(64, C, 9).transpose(0, 2, 1).reshape(64, 576)       # tap-major columns
    g0 = rng.standard_normal((64, 576)).astype(np.float32)
    gp = dev(g0)
    b0 = rng.standard_normal(64).astype(np.float32)
    gb = dev(b0)
    ops.conv3x3_implicit_wgrad(dev16(dy), X, gp, N, gbias_acc=gb)
    close(gp, g0 + gw_ref, 1e-5, name="implicit conv wgrad")
    close(gb, b0 + dy.sum(0), 1e-5, name="implicit conv bias gradient (column sums of dy from the same kernel)")
    gp2, gb2 = dev(g0), dev(b0)
    ops.conv3x3_implicit_wgrad(dev16(dy), X, gp2, N, gbias_acc=gb2)
    assert torch.equal(gp2, gp) and torch.equal(gb2, gb)       # fixed-order partial sums: bit-reproducible


@pytest.mark.parametrize("N", [3, 256, 700])
def test_patch_resident_convolution_equals_the_tile_form(ops, N):
    """round 6: db1_conv3x3_implicit_fwd's default kernel keeps a patch (and the whole weight operand) in LDS and reads the nine taps as nine
    row addresses into it; one persistent workgroup per CU walks the patches through two image buffers.  Same products in the same order
    as the nine-k-step tile form (knob conv_patch = 0): bit-identical outputs -- forward with bias, with bias + residual, data gradient --
    for fewer patches than CUs, exactly one each, and several per workgroup (incl. the idle re-stage behind a workgroup's last patch)."""
    from bdm_db1_amd import lib as db1lib
    rng = np.random.default_rng(77 + N)
    x = dev16(rng.standard_normal((N * 256, 64)))
    w = dev16(rng.standard_normal((64, 64, 3, 3)) * 0.1)
    bias = dev16(rng.standard_normal(64))
    res = dev16(rng.standard_normal((N * 256, 64)))
    w_op, w_t = (torch.empty(64, 576, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    ops.conv_weight_permute(w, w_op, 64, 64)
    ops.conv_weight_permute_t(w, w_t, 64, 64)
    outs = {}
    for knob in (0, 1):
        db1lib.set_knob("conv_patch", knob)
        try:
            y0, y1, y2 = (torch.full((N * 256, 64), float("nan"), device=DEV, dtype=torch.bfloat16) for _ in range(3))
            ops.conv3x3_implicit_fwd(x, w_op, bias, y0, N, sign=1)
            ops.conv3x3_implicit_fwd(x, w_op, bias, y1, N, sign=1, res=res)
            ops.conv3x3_implicit_fwd(x, w_t, None, y2, N, sign=-1)
            ops.conv3x3_implicit_fwd(x, w_op, bias.float(), y0.clone(), N, sign=1)     # (fp32 bias instantiation runs)
            gp, gb = torch.full((64, 576), 0.5, device=DEV), torch.full((64,), -0.25, device=DEV)
            ops.conv3x3_implicit_wgrad(res, x, gp, N, gbias_acc=gb)                    # (res plays dY)
            gp2, gb2 = torch.full((64, 576), 0.5, device=DEV), torch.full((64,), -0.25, device=DEV)
            ops.conv3x3_implicit_wgrad(res, x, gp2, N, gbias_acc=gb2)
            assert torch.equal(gp, gp2) and torch.equal(gb, gb2), "the weight gradient is not run-to-run identical"
            outs[knob] = (y0, y1, y2, gp, gb)
        finally:
            db1lib.load().db1_test_clear_knobs()
    for a, b_, name in zip(outs[1][:3], outs[0][:3], ("fwd + bias", "fwd + bias + residual", "data gradient")):
        assert torch.isfinite(a.float()).all(), name
        assert torch.equal(a.view(torch.int16), b_.view(torch.int16)), f"patch-resident {name} differs from the tile form (N = {N})"
    # the weight / bias gradients: the same bf16 products, summed in another order in fp32 (per-workgroup partial slabs vs per-pixel-range ones)
    for a, b_, name in zip(outs[1][3:], outs[0][3:], ("weight gradient", "bias gradient")):
        ref = b_.double()
        assert torch.isfinite(a).all() and float((a.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), f"patch-resident {name} (N = {N})"
    # ... and against fp64 arithmetic on the bf16 operands, for the smallest case
    if N == 3:
        xs = x.double().view(N, 16, 16, 64)
        xp = torch.zeros(N, 18, 18, 64, dtype=torch.float64, device=DEV)
        xp[:, 1:17, 1:17] = xs
        dyf = res.double().view(N * 256, 64)
        want = torch.stack([dyf.t() @ xp[:, t // 3:t // 3 + 16, t % 3:t % 3 + 16].reshape(N * 256, 64) for t in range(9)], 1).reshape(64, 576) + 0.5
        assert float((outs[1][3].double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_vision_position_add_and_vector_add(ops, dtype):
    """db1_vision_pos_add (round 6): out += row_table[row_ids] + col_table[col_ids] in one pass (vision_embedding.py:170-178; it was two
    gathers + two adds), fp32 sum (out + row) + col with one rounding; ids outside the table add nothing.  And db1_add's 16-byte form
    (aligned, n a multiple of the vector) against its scalar form (odd n)."""
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(3)
    N, d, V = 333, 512, 128
    emb = torch.randn(N, d, generator=g).to(td).to(DEV)
    rt, ct = torch.randn(V, d, generator=g).to(td).to(DEV), torch.randn(V, d, generator=g).to(td).to(DEV)
    rid, cid = torch.randint(0, V, (N,), generator=g).to(DEV), torch.randint(0, V, (N,), generator=g).to(DEV)
    rid[5], cid[7] = -1, V + 3
    okr, okc = ((rid >= 0) & (rid < V)).unsqueeze(1), ((cid >= 0) & (cid < V)).unsqueeze(1)
    want = ((emb.float() + torch.where(okr, rt.float()[rid.clamp(0, V - 1)], torch.zeros(1, device=DEV))) +
            torch.where(okc, ct.float()[cid.clamp(0, V - 1)], torch.zeros(1, device=DEV))).to(td)
    out = emb.clone()
    ops.vision_pos_add(out, rt, ct, rid, cid)
    assert torch.equal(out, want)
    for n in (8 * 1000, 8 * 1000 + 3):
        a, b = torch.randn(n, generator=g).to(td).to(DEV), torch.randn(n, generator=g).to(td).to(DEV)
        y = torch.empty_like(a)
        ops.add(a, b, y)
        assert torch.equal(y, (a.float() + b.float()).to(td))
        ops.add(a, b, a)      # in place
        assert torch.equal(a, y)


def test_conv1_fused_kernel_equals_im2col_plus_gemm(ops):
    """db1_conv1_fused_fwd (3 -> 64 channels on 16 x 16 patches, channels-last): its column matrix is bit-equal to db1_im2col3x3_nhwc's and its
    output equals the K = 32 GEMM over that matrix up to the bf16 rounding of the result"""
    rng = np.random.default_rng(21)
    N = 37
    x = dev16(rng.standard_normal((N * 256, 3)))
    w = dev16(rng.standard_normal((64, 3, 3, 3)) * 0.2)
    bias = dev16(rng.standard_normal(64) * 0.1)
    wp = torch.empty(64, 32, device=DEV, dtype=torch.bfloat16)
    ops.conv_weight_permute(w, wp, 64, 3)
    cols_ref = torch.empty(N * 256, 32, device=DEV, dtype=torch.bfloat16)
    ops.im2col3x3_nhwc(x, cols_ref, N, 3, 16)
    y_ref = torch.empty(N * 256, 64, device=DEV, dtype=torch.bfloat16)
    ops.gemm(cols_ref, wp.t(), y_ref, bias=bias)
    cols = torch.full((N * 256, 32), 7.0, device=DEV, dtype=torch.bfloat16)
    y = torch.empty(N * 256, 64, device=DEV, dtype=torch.bfloat16)
    ops.conv1_fused_fwd(x, wp, bias, cols, y, N)
    assert torch.equal(cols, cols_ref)
    err = (y.float() - y_ref.float()).abs().max().item() / y_ref.float().abs().max().item()
    assert err < 1e-2, err
    ref64 = cols_ref.double() @ wp.double().t() + bias.double()
    assert (y.double() - ref64).abs().max().item() <= (y_ref.double() - ref64).abs().max().item() * 1.5 + 1e-3
