"""Relative-position flash attention (bf16, d_head 128) against the CPU oracle and against the materialised HIP path,
including sliding windows (0 < mem_len < L) and lengths that wrap the 256-row R ring."""
import math
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import db1_oracle as O  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def bf(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


def dev16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(torch.bfloat16)


def rel_err(got, ref):
    got = got.detach().to(torch.float64).cpu().numpy() if hasattr(got, "detach") else np.asarray(got, np.float64)
    return np.abs(got - np.asarray(ref, np.float64)).max() / (np.abs(ref).max() + 1e-30)


def make_inputs(B, L, H, D, seed, scale_q=1.0):
    rng = np.random.default_rng(seed)
    qkv = bf(rng.standard_normal((B, L, 3, H, D)) * scale_q)
    R = bf(rng.standard_normal((L, H, D)))
    u, vb = bf(rng.standard_normal((H, D)) * 0.5), bf(rng.standard_normal((H, D)) * 0.5)
    return qkv, R, u, vb


def masked_for(L, shift):
    i = np.arange(L)[:, None]
    j = np.arange(L)[None, :]
    return (~((j <= i) & (j > i - shift))).astype(np.uint8)


@pytest.mark.parametrize("B,L,H,shift", [(2, 256, 2, 256), (1, 512, 2, 512), (1, 512, 1, 130), (1, 384, 3, 33), (1, 1024, 1, 1024),
                                         (1, 128, 1, 128), (3, 256, 3, 70), (1, 2048, 1, 2048), (1, 640, 2, 1)])
def test_flash_forward_matches_oracle(B, L, H, shift):
    from bdm_db1_amd import ops
    D = 128
    qkv, R, u, vb = make_inputs(B, L, H, D, seed=L + shift)
    scale = 1.0 / math.sqrt(D)
    out_ref, (Pm, _, _, _, _) = O.relattn_core_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], R, u, vb, masked_for(L, shift), scale)
    QKV, Rd, U, VB = dev16(qkv), dev16(R), dev16(u), dev16(vb)
    qu, qv = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16), torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    ops.relattn_add_head_bias(QKV, U, VB, qu, qv, B, L, L, H, D)
    assert ops.relattn_flash_supported(B, L, H, D, torch.bfloat16)
    out = torch.full((B, L, H, D), 7.0, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device=DEV, dtype=torch.float32)
    ops.relattn_flash_fwd(qu, qv, QKV, Rd, out, lse, B, L, H, D, shift, scale)
    torch.cuda.synchronize()
    # stated tolerance: bf16 rounding of q+u / q+v, of P and of the output
    e = rel_err(out, out_ref)
    assert e < 2e-2, f"out rel err {e:.3e}"
    # lse against the oracle's scores
    qu_r, qv_r = qkv[:, :, 0] + u, qkv[:, :, 0] + vb
    AC = np.einsum("bind,bjnd->bnij", qu_r, qkv[:, :, 1])
    T = np.einsum("bind,rnd->bnir", qv_r, R)
    i = np.arange(L)[:, None]; j = np.arange(L)[None, :]
    BD = np.take_along_axis(T, np.broadcast_to(np.clip(i - j, 0, L - 1)[None, None], AC.shape), axis=3)
    S = np.where(masked_for(L, shift)[None, None].astype(bool), -np.inf, (AC + BD) * scale)
    mx = S.max(-1)
    lse_ref = mx + np.log(np.exp(S - mx[..., None]).sum(-1))
    assert np.abs(lse.cpu().numpy() - lse_ref).max() < 0.15, np.abs(lse.cpu().numpy() - lse_ref).max()


def test_flash_forward_matches_materialised_path_full_size():
    """DB1-1.3B attention geometry (H=16, D=128, L=1024): fused kernel vs the strided-GEMM + softmax path on the same inputs."""
    from bdm_db1_amd import ops
    B, L, H, D = 2, 1024, 16, 128
    qkv, R, u, vb = make_inputs(B, L, H, D, seed=7, scale_q=0.7)
    scale = 1.0 / math.sqrt(D)
    QKV, Rd, U, VB = dev16(qkv), dev16(R), dev16(u), dev16(vb)
    qu, qv = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16), torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    ops.relattn_add_head_bias(QKV, U, VB, qu, qv, B, L, L, H, D)
    out = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device=DEV, dtype=torch.float32)
    ops.relattn_flash_fwd(qu, qv, QKV, Rd, out, lse, B, L, H, D, L, scale)
    # materialised path
    qkv5 = QKV.view(B, L, 3, H, D)
    AC = torch.empty(H, B, L, L, device=DEV, dtype=torch.float32)
    ops.gemm_batched(qu.permute(2, 0, 1, 3), qkv5[:, :, 1].permute(2, 0, 3, 1), AC)
    T = torch.empty(H, B, L, L, device=DEV, dtype=torch.float32)
    ops.gemm_batched(qv.permute(2, 0, 1, 3), Rd.view(L, H, D).permute(1, 2, 0).unsqueeze(1).expand(H, B, D, L), T)
    lse2 = torch.empty(H, B, L, device=DEV, dtype=torch.float32)
    ops.relattn_softmax_fwd(AC, T, lse2, H, B, L, L, L, 0, L, scale)
    out2 = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    ops.gemm_batched(AC, qkv5[:, :, 2].permute(2, 0, 1, 3), out2.permute(2, 0, 1, 3))
    torch.cuda.synchronize()
    assert rel_err(out, out2.to(torch.float64).cpu().numpy()) < 2e-2
    assert float((lse - lse2.permute(1, 0, 2)).abs().max()) < 5e-2


@pytest.mark.parametrize("B,L,H,shift", [(2, 256, 2, 256), (1, 512, 2, 130), (1, 384, 1, 384), (1, 128, 1, 128), (3, 256, 3, 70),
                                         (1, 1024, 1, 1024), (1, 640, 1, 40)])
@pytest.mark.parametrize("mode", ["fwd_probs", "scratch_p_ds", "recompute"])
def test_flash_backward_matches_oracle(B, L, H, shift, mode):
    """the three backward passes: nothing recomputed (the forward kept p~ and its block maxima), the key side over the P and dS the
    recomputing query-side kernel left in the workspace, and both sides recomputing.  Saved tensors and workspace are pre-filled with
    NaN patterns: tiles that are never written (blocks outside the window) must not leak."""
    from bdm_db1_amd import lib, ops
    D = 128
    qkv, R, u, vb = make_inputs(B, L, H, D, seed=11 + L + shift, scale_q=0.8)
    rng = np.random.default_rng(99)
    dout = bf(rng.standard_normal((B, L, H, D)))
    scale = 1.0 / math.sqrt(D)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    out_ref, cache = O.relattn_core_fwd(q, k, v, R, u, vb, masked_for(L, shift), scale)
    dq_ref, dk_ref, dv_ref, dR_ref, du_ref, dvb_ref = O.relattn_core_bwd(dout, q, k, v, R, u, vb, scale, cache)
    QKV, Rd, U, VB = dev16(qkv), dev16(R), dev16(u), dev16(vb)
    qu, qv = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16), torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    ops.relattn_add_head_bias(QKV, U, VB, qu, qv, B, L, L, H, D)
    out = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device=DEV, dtype=torch.float32)
    probs = mblk = None
    if mode == "fwd_probs":
        probs = torch.full((B * H, ops.relattn_flash_probs_tiles(L), 512), float("nan"), device=DEV, dtype=torch.bfloat16)
        mblk = torch.full((B * H, L // 32, L), float("nan"), device=DEV, dtype=torch.float32)
    ops.relattn_flash_fwd(qu, qv, QKV, Rd, out, lse, B, L, H, D, shift, scale, probs=probs, mblk=mblk)
    # the key side of the stored-probabilities backward has two kernels -- 32 keys per wave (L % 256 == 0 and enough workgroups: the
    # benchmark's batches) and 16 keys per wave (small batches since round 5): both are checked here whatever the dispatcher would pick
    for kv3 in ((1, 0) if (mode == "fwd_probs" and L % 256 == 0) else (None,)):
        if kv3 is not None:
            lib.set_knob("flash_kv3", kv3)
        dqkv = torch.full((B, L, 3, H, D), 3.0, device=DEV, dtype=torch.bfloat16)
        dT = torch.zeros(H, B, L, L, device=DEV, dtype=torch.bfloat16)
        delta = torch.empty(B, H, L, device=DEV, dtype=torch.float32)
        if mode != "recompute":
            ops.reserve_workspace(int(lib.load().db1_relattn_flash_bwd_workspace_bytes(B, L, H, int(mode == "fwd_probs"))))
            for buf in ops._workspace.bufs.values():
                buf.fill_(0xFF)
        try:
            ops.relattn_flash_bwd(qu, qv, QKV, Rd, out, dev16(dout), lse, delta, dqkv, dT, B, L, H, D, shift, scale, store_probs=mode != "recompute",
                                  probs=probs, mblk=mblk)
            torch.cuda.synchronize()
        finally:
            if kv3 is not None:
                lib.set_knob("flash_kv3", -1)     # back to the dispatcher's own choice
        g = dqkv.to(torch.float64).cpu().numpy()
        dTn = dT.to(torch.float64).cpu().numpy()
        assert rel_err(g[:, :, 2], dv_ref) < 3e-2, ("dv", kv3)
        assert rel_err(g[:, :, 1], dk_ref) < 3e-2, ("dk", kv3)
        dqr = np.einsum("nbir,rnd->bind", dTn, R)
        assert rel_err(g[:, :, 0] + dqr, dq_ref) < 3e-2, "dq"
        dR = np.einsum("nbir,bind->rnd", dTn, qv.to(torch.float64).cpu().numpy())
        assert rel_err(dR, dR_ref) < 3e-2, "dR"
        assert rel_err(g[:, :, 0].sum((0, 1)), du_ref) < 3e-2, "du"
        assert rel_err(dqr.sum((0, 1)), dvb_ref) < 3e-2, "dv_bias"
        assert np.abs(delta.cpu().numpy() - np.einsum("bind,bind->bni", out.to(torch.float64).cpu().numpy(), dout)).max() < 5e-2


def _build_d128_model(compute_dtype, seed=5):
    from bdm_db1_amd import TransformerXL
    cfg = dict(n_embed=256, n_position=256, n_layer=2, n_head=2, n_inner=None, pre_lnorm=False, mem_len=256, same_length=True,
               untie_r=False, text_vocab_size=500, num_discrete_values=64, num_continuous_bin=64, overlap_with_text=True,
               embd_pdrop=0.0, drop=0.0, dropattn=0.0, activation_fn="geglu", layer_norm_epsilon=1e-5,
               share_input_output_embedding=True, use_deepnorm=False, fp16=False, vision_patch_size=16,
               vision_num_input_channels=3, vision_position_vocab_size=128, vision_hidden_dropout_prob=0.0)
    from golden_util import make_params
    params = make_params(cfg, seed)
    model = TransformerXL(SimpleNamespace(**cfg), compute_dtype=compute_dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    params["pos_emb.inv_freq"] = model.pos_emb.inv_freq.cpu().numpy()
    return cfg, params, model


def test_model_bf16_flash_vs_oracle_and_vs_materialised():
    from bdm_db1_amd.data import NLPTaskInput
    cfg, params, model = _build_d128_model(torch.bfloat16)
    rng = np.random.default_rng(3)
    B, L = 3, 256
    ids = rng.integers(0, 500, (B, L + 1))
    mask = (rng.random((B, L)) > 0.2).astype(np.float32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    mk = lambda: NLPTaskInput(position_id=None, attention_mask=None, loss_mask=T(mask), label=T(ids[:, 1:]), text_seq=T(ids[:, :-1]), text_len=None)
    oracle = O.OracleModel(O.OracleConfig(**cfg), params)
    ref_logits, ref_loss, _ = oracle.forward([O.TaskBatch(kind="nlp", text_seq=ids[:, :-1], label=ids[:, 1:], loss_mask=mask)])
    ref_grads = oracle.backward()
    results = {}
    for flash in ("forward", "scratch", "recompute", False):   # the three flash backward passes, then the materialised path
        model.use_flash = bool(flash)
        model.flash_probs_mode = flash or "forward"
        model.zero_grad()
        logits, loss = model([mk()])
        model.backward()
        results[flash] = (logits.float().cpu().numpy().copy(), float(loss), {n: model.G(n).cpu().numpy().copy() for n in ref_grads})
        assert rel_err(results[flash][0], ref_logits) < 3e-2, flash
        assert abs(results[flash][1] - ref_loss) < 2e-2, flash
        for n in ("h.0.dec_attn.qkv_net.weight", "h.1.dec_attn.o_net.weight", "r_w_bias", "r_r_bias", "h.0.dec_attn.r_net.weight", "word_embedding.weight"):
            assert rel_err(results[flash][2][n], ref_grads[n]) < 6e-2, (flash, n)
    assert np.abs(results["forward"][0] - results[False][0]).max() / np.abs(ref_logits).max() < 2e-2
    # the memory guard: "forward" is demoted to "scratch" when the kept probabilities would exceed the budget fraction of the device memory
    model.flash_probs_mode = "forward"
    assert model._probs_mode(B, L) == "forward"
    model.flash_probs_budget = 0.0
    assert model._probs_mode(B, L) == "scratch"
    model.flash_probs_budget = 0.25


@pytest.mark.parametrize("B,L,H", [(2, 256, 2), (3, 1024, 3), (1, 640, 1), (20, 128, 16)])
def test_dq_r_stream_kernel_matches_the_batched_gemm(B, L, H):
    """dq_r = dT.R with dT streamed once and R stationary in registers vs the same contraction on the tile GEMM"""
    from bdm_db1_amd import ops
    D = 128
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + L)
    dT = (torch.randn(H, B, L, L, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    ii = torch.arange(L, device=DEV)
    dT = dT * (ii[None, :] <= ii[:, None]).to(torch.bfloat16)          # zero above the causal diagonal (dist > i)
    R = torch.randn(L, H * D, generator=g).to(torch.bfloat16).to(DEV)
    assert ops.relattn_dqr_supported(B, L, H, D, torch.bfloat16)
    out = torch.full((B, L, H, D), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.relattn_dqr(dT, R, out)
    ref = torch.einsum("hbik,khd->bihd", dT.float(), R.view(L, H, D).float())
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    assert err < 6e-3, err
    # fused epilogue: dq = dq_k + dq_r in place over the q slot of a packed [B, L, 3, H, D] gradient (row stride 3 H D, like the model's
    # dqkv), with the column sums of both terms accumulated onto existing float32 accumulators (the u / v bias gradients)
    dqkv = (torch.randn(B, L, 3, H, D, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    dq_k = dqkv[:, :, 0].float().clone()
    others = dqkv[:, :, 1:].clone()
    du, dv = torch.full((H * D,), 0.25, device=DEV), torch.full((H * D,), -0.5, device=DEV)
    ops.relattn_dqr_fused(dT, R, dqkv[:, :, 0], du, dv)
    want = dq_k + ref
    assert float((dqkv[:, :, 0].float() - want).abs().max() / want.abs().max()) < 6e-3
    assert torch.equal(dqkv[:, :, 1:], others)                       # the k / v slots are untouched
    su, sv = dq_k.sum((0, 1)).reshape(-1) + 0.25, ref.sum((0, 1)).reshape(-1) - 0.5
    assert float((du - su).abs().max() / su.abs().max()) < 1e-4 and float((dv - sv).abs().max() / sv.abs().max()) < 2e-3
    # deterministic: a second run on the same inputs gives the same bits
    dqkv2 = torch.cat([dq_k.to(torch.bfloat16).unsqueeze(2), others], dim=2).contiguous()
    du2, dv2 = torch.full((H * D,), 0.25, device=DEV), torch.full((H * D,), -0.5, device=DEV)
    ops.relattn_dqr_fused(dT, R, dqkv2[:, :, 0], du2, dv2)
    assert torch.equal(dqkv2[:, :, 0], dqkv[:, :, 0]) and torch.equal(du2, du) and torch.equal(dv2, dv)


@pytest.mark.parametrize("B,L,H,ng", [(8, 256, 4, 4), (4, 1024, 16, 2), (16, 128, 16, 16)])
def test_dq_r_stream_with_one_R_per_group_of_sequences(B, L, H, ng):
    """db1_relattn_dqr_fused_groups (round 6): the batch is `ng` blocks of sequences and block g has its OWN R (every micro-step of an
    accumulation window draws its own position-table dropout) -- one launch for all of them, bit-identical in dq to `ng` launches on the blocks
    (every output row is the same MFMA sequence either way), the u / v column sums equal to summation order, and dq against fp32 arithmetic"""
    from bdm_db1_amd import ops
    D = 128
    g = torch.Generator(device="cpu").manual_seed(B * 100 + L + ng)
    dT = (torch.randn(H, B, L, L, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    ii = torch.arange(L, device=DEV)
    dT = (dT * (ii[None, :] <= ii[:, None]).to(torch.bfloat16)).contiguous()
    # R of group g sits inside a larger [ng, layers = 3, L, H D] tensor (the model's batched r_net output): a strided [ng, L, H D] view
    Rall = torch.randn(ng, 3, L, H * D, generator=g).to(torch.bfloat16).to(DEV)
    Rg = Rall[:, 1]
    assert ops.relattn_dqr_groups_supported(B, L, H, D, torch.bfloat16, ng)
    dqkv = (torch.randn(B, L, 3, H, D, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    dqkv2 = dqkv.clone()
    dq_k = dqkv[:, :, 0].float().clone()
    du, dv = torch.full((H * D,), 0.25, device=DEV), torch.full((H * D,), -0.5, device=DEV)
    ops.relattn_dqr_fused_groups(dT, Rg, dqkv[:, :, 0], du, dv)
    du2, dv2 = torch.full((H * D,), 0.25, device=DEV), torch.full((H * D,), -0.5, device=DEV)
    Bm = B // ng
    for k in range(ng):
        ops.relattn_dqr_fused(dT[:, k * Bm:(k + 1) * Bm].contiguous(), Rg[k], dqkv2[k * Bm:(k + 1) * Bm, :, 0], du2, dv2)
    assert torch.equal(dqkv, dqkv2), "one launch over the groups differs from a launch per group"
    assert float((du - du2).abs().max()) <= 1e-4 * float(du2.abs().max()) and float((dv - dv2).abs().max()) <= 1e-3 * float(dv2.abs().max())
    ref = torch.cat([torch.einsum("hbik,khd->bihd", dT[:, k * Bm:(k + 1) * Bm].float(), Rg[k].reshape(L, H, D).float()) for k in range(ng)], 0)
    want = dq_k + ref
    assert float((dqkv[:, :, 0].float() - want).abs().max() / want.abs().max()) < 6e-3
    assert not ops.relattn_dqr_groups_supported(B, L, H, D, torch.bfloat16, 3)       # (3 divides neither B nor the head's workgroup count)


@pytest.mark.parametrize("B,L,H", [(1, 128, 1), (2, 256, 2), (1, 384, 1), (1, 1024, 2)])
@pytest.mark.parametrize("mode", [1, 2])
def test_hand_scheduled_forward_matches_compiled_loop(B, L, H, mode):
    """the forward that keeps its probabilities, plain causal window: the hand-scheduled key-block loops (1: four waves x 32 rows,
    relattn_flash_fwd3.hip, the default; 2: eight waves x 16 rows, relattn_flash_fwd2.hip) against the compiled loop on the same inputs.
    Their block maxima are DEFERRED (the maximum a block was exponentiated against), so what is compared is what the backward rebuilds
    from the images: P = p~ exp2(m_blk c2 - lse log2 e); plus out, lse and the pattern of never-written (NaN pre-filled) tiles."""
    from bdm_db1_amd import ops
    D = 128
    qkv, R, u, vb = make_inputs(B, L, H, D, seed=3 + L, scale_q=0.8)
    scale = 1.0 / math.sqrt(D)
    QKV, Rd, U, VB = dev16(qkv), dev16(R), dev16(u), dev16(vb)
    qu, qv = torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16), torch.empty(B, L, H, D, device=DEV, dtype=torch.bfloat16)
    ops.relattn_add_head_bias(QKV, U, VB, qu, qv, B, L, L, H, D)
    res = []
    try:
        for m in (0, mode):
            ops.flash_fwd2(m)
            out = torch.full((B, L, H, D), 7.0, device=DEV, dtype=torch.bfloat16)
            lse = torch.full((B, H, L), 3.0, device=DEV, dtype=torch.float32)
            probs = torch.full((B * H, ops.relattn_flash_probs_tiles(L), 512), float("nan"), device=DEV, dtype=torch.bfloat16)
            mblk = torch.full((B * H, L // 32, L), float("nan"), device=DEV, dtype=torch.float32)
            ops.relattn_flash_fwd(qu, qv, QKV, Rd, out, lse, B, L, H, D, L, scale, probs=probs, mblk=mblk)
            torch.cuda.synchronize()
            res.append((out.float().cpu(), lse.cpu(), ops.relattn_flash_probs_full(probs, L).float().cpu(), mblk.cpu()))    # (the triangle of images, expanded)
    finally:
        ops.flash_fwd2(1)

    def rebuilt(p, m, l):   # image [bh][jb][qt][lane][8], lane & 15 = query inside its 16-row tile
        BH, NJ, NQ = p.shape[0], p.shape[1], p.shape[2]
        f = torch.exp2(m.view(BH, NJ, NQ, 1, 16) - l.reshape(BH, 1, NQ, 1, 16) * 1.4426950408889634).expand(BH, NJ, NQ, 4, 16).reshape(BH, NJ, NQ, 64, 1)
        return p.view(BH, NJ, NQ, 64, 8) * f
    (o0, l0, p0, m0), (o1, l1, p1, m1) = res
    assert bool((torch.isnan(p0) == torch.isnan(p1)).all()) and bool((torch.isnan(m0) == torch.isnan(m1)).all())
    assert bool(torch.isfinite(o1).all()) and bool(torch.isfinite(l1).all())
    P0, P1 = torch.nan_to_num(rebuilt(p0, m0, l0)), torch.nan_to_num(rebuilt(p1, m1, l1))
    assert float((P0 - P1).abs().max()) < 6e-3                    # probabilities (<= 1): bf16 rounding of p~ at two different scales
    assert float((l0 - l1).abs().max()) < 8e-3                    # row sums over the bf16 p~ (matrix pipe) vs over the fp32 values
    assert float((o0 - o1).abs().max()) < 1.5e-2 * float(o0.abs().max())
    assert float(torch.nan_to_num(p1).abs().max()) <= 256.0       # deferred maximum: p~ is bounded by 2^8
