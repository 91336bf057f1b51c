"""The "bias + GEGLU" epilogue of the feed-forward GEMMs (PositionwiseFF transformer_xl.py:246-292, GEGLU activations.py:19-32):
db1_gemm_nt_geglu / db1_gemm_nn_geglu_bwd against (a) the separate launches they replace -- z, act and dz bit for bit, the bias gradient to
fp32 summation order -- and (b) the CPU oracle's GEGLU on the bf16-rounded operands; at shapes where the activation runs inside the 4-wave GEMM
(asserted) and at shapes / dtypes where the entry points fall back to separate launches."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import db1_oracle as O  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bdm_db1_amd import ops as _ops
    return _ops


def _mk(M, K, dff, dtype, seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=g) * sc).to(dtype)
    return r(M, K), r(2 * dff, K, sc=0.03), r(2 * dff, sc=0.5), r(M, K), r(K, dff, sc=0.03)   # x, W1, b1, dy, W2 (= CoreNet.2.weight [d, dff])


@pytest.mark.parametrize("M,K,dff", [(2560, 2048, 4096), (10240, 512, 1024), (5120, 256, 1280)])
def test_geglu_epilogues_equal_the_separate_launches_bf16(ops, M, K, dff):
    dt = torch.bfloat16
    x, W1, b1, dy, W2 = _mk(M, K, dff, dt, 11)
    assert ops.gemm_nt_geglu_fused(M, dff, K, dt), "this shape is meant to take the fused forward epilogue"
    z0, a0 = torch.empty(M, 2 * dff, device=DEV, dtype=dt), torch.empty(M, dff, device=DEV, dtype=dt)
    ops.gemm(x, W1.t(), z0, bias=b1)
    ops.ffn_act_fwd(z0, a0, "geglu")
    z1, a1 = torch.full_like(z0, float("nan")), torch.full_like(a0, float("nan"))
    ops.gemm_nt_geglu(x, W1, b1, z1, a1)
    # z: the same products summed over k in a different ORDER (the NT kernel rotates its k-tiles per XCD, and a column sits in another tile
    # here), so fp32 round-off may flip the last bf16 bit of a few values -- nothing more
    dzz = (z1.float() - z0.float()).abs()
    assert float(dzz.max()) <= 2.0 ** -7 * float(z0.float().abs().max()) and float((dzz > 0).float().mean()) < 2e-2, (float(dzz.max()), float((dzz > 0).float().mean()))
    # act: computed from the ROUNDED z the kernel stored -> bit-identical to the separate activation pass over that z
    ops.ffn_act_fwd(z1, a0, "geglu")
    assert torch.equal(a1.view(torch.int16), a0.view(torch.int16)), "act of the fused epilogue differs from db1_ffn_act_fwd on the stored z"
    # against the oracle's GEGLU on the stored z (bf16 output rounding only)
    zf = z1.double().cpu().numpy()
    ref = zf[:, :dff] * O.gelu(zf[:, dff:])
    assert np.abs(a1.double().cpu().numpy() - ref).max() <= 6e-3 * np.abs(ref).max()
    # ---- backward
    if (M // 256) * (dff // 256) >= 160:
        assert ops.gemm_nn_geglu_bwd_fused(M, dff, K, dt), "this shape is meant to take the fused backward epilogue"
    da0 = torch.empty(M, dff, device=DEV, dtype=dt)
    ops.gemm(dy, W2, da0)
    dz0 = torch.empty_like(z0)
    gb0 = torch.full((2 * dff,), 0.25, device=DEV)
    ops.ffn_act_bwd_bias(z0, da0, dz0, gb0, "geglu")
    dz1 = torch.full_like(z0, float("nan"))
    gb1 = torch.full((2 * dff,), 0.25, device=DEV)
    ops.gemm_nn_geglu_bwd(dy, W2, z0, dz1, gb1)
    assert torch.equal(dz1.view(torch.int16), dz0.view(torch.int16)), "dz of the fused epilogue differs from GEMM + db1_ffn_act_bwd_bias"
    want = dz1.double().sum(0).cpu().numpy() + 0.25
    assert np.abs(gb1.double().cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max(), "bias gradient != column sums of the stored dz"
    assert np.abs(gb1.double().cpu().numpy() - gb0.double().cpu().numpy()).max() <= 2e-5 * np.abs(want).max()
    daf, g = da0.double().cpu().numpy(), zf[:, dff:]
    dref = np.concatenate([daf * O.gelu(g), daf * zf[:, :dff] * O.gelu_grad(g)], -1)
    assert np.abs(dz1.double().cpu().numpy() - dref).max() <= 6e-3 * np.abs(dref).max()
    # run-to-run determinism of the bias gradient (fixed-order partial sums)
    gb2 = torch.full((2 * dff,), 0.25, device=DEV)
    ops.gemm_nn_geglu_bwd(dy, W2, z0, torch.empty_like(z0), gb2)
    assert torch.equal(gb1, gb2)


@pytest.mark.parametrize("dtype,M,K,dff", [("f32", 48, 40, 24), ("bf16", 256, 128, 128), ("bf16", 768, 256, 384)])
def test_geglu_entry_points_fall_back_to_separate_launches(ops, dtype, M, K, dff):
    """small models and the fp32 parity gate: the same entry points run GEMM + activation as separate launches (same results)"""
    dt = torch.float32 if dtype == "f32" else torch.bfloat16
    x, W1, b1, dy, W2 = _mk(M, K, dff, dt, 12)
    assert not ops.gemm_nt_geglu_fused(M, dff, K, dt) and not ops.gemm_nn_geglu_bwd_fused(M, dff, K, dt)
    z0, a0 = torch.empty(M, 2 * dff, device=DEV, dtype=dt), torch.empty(M, dff, device=DEV, dtype=dt)
    ops.gemm(x, W1.t(), z0, bias=b1)
    ops.ffn_act_fwd(z0, a0, "geglu")
    z1, a1 = torch.empty_like(z0), torch.empty_like(a0)
    ops.gemm_nt_geglu(x, W1, b1, z1, a1)
    assert torch.equal(z1, z0) and torch.equal(a1, a0)
    da0 = torch.empty(M, dff, device=DEV, dtype=dt)
    ops.gemm(dy, W2, da0)
    dz0, dz1 = torch.empty_like(z0), torch.empty_like(z0)
    gb0, gb1 = torch.zeros(2 * dff, device=DEV), torch.zeros(2 * dff, device=DEV)
    ops.ffn_act_bwd_bias(z0, da0, dz0, gb0, "geglu")
    ops.gemm_nn_geglu_bwd(dy, W2, z0, dz1, gb1)
    assert torch.equal(dz1, dz0) and torch.equal(gb1, gb0)
    # fp32: against the oracle to fp32 round-off
    if dtype == "f32":
        xf, wf, bfz = x.double().cpu().numpy(), W1.double().cpu().numpy(), b1.double().cpu().numpy()
        zr = xf @ wf.T + bfz
        assert np.abs(z1.double().cpu().numpy() - zr).max() <= 1e-5 * np.abs(zr).max()
        ar = zr[:, :dff] * O.gelu(zr[:, dff:])
        assert np.abs(a1.double().cpu().numpy() - ar).max() <= 1e-5 * np.abs(ar).max()


def test_model_step_with_and_without_the_geglu_epilogue_agree():
    """two DB1-1.3B-geometry layers, bf16, one training step each way: the fused epilogues change which launches run and the k order of the
    first feed-forward product (last-bit flips of z), nothing else: loss and every gradient agree far inside the bf16 tolerance"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bdm_db1_amd import TransformerXL, synth
    cfg = synth.db1_config("1.3B", n_layer=2, drop=0.1, embd_pdrop=0.1)
    B, L = 12, 1024
    outs = []
    for fused in (True, False):
        torch.manual_seed(5)
        model = TransformerXL(cfg)
        model.use_geglu_epilogue = fused
        model.train()
        batch = [synth.text_batch(B, L, 77, model.dev)]
        _, loss = model(batch)
        model.backward()
        outs.append((float(loss), model.arena.grad.clone(), dict(model.arena.offsets)))
        del model
        torch.cuda.empty_cache()
    (l1, g1, offs), (l0, g0, _) = outs
    assert abs(l1 - l0) <= 1e-4 * abs(l0)
    worst = 0.0
    for name, (off, shape, alloc) in offs.items():
        a, b = g1[off:off + alloc].double(), g0[off:off + alloc].double()
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, name
            continue
        l2 = float((a - b).norm() / b.norm())
        worst = max(worst, l2)
        assert l2 <= 5e-3, (name, l2)
    print(f"worst relative L2 of a gradient against the unfused step: {worst:.2e} (limit 5e-03)")
