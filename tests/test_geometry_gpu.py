"""Parity at the geometry that is benchmarked: two decoder layers of DB1-1.3B -- d = 2048, 16 heads
of 128, GEGLU 8192 -> 4096, L = 1024, the tied 33 025-row head -- on 8 sequences, against the CPU oracle (fp32 NumPy, itself pinned
to the reference by tests/test_oracle_golden.py):

  * the fp32 instantiation: logits within 1e-3 relative (north_star's gate), loss, a sample of every kind of gradient;
  * the bf16 path with the DEFAULT dispatch -- the hand-scheduled 4-wave GEMMs in all three operand layouts, the head-bias epilogue of
    the qkv projection, flash attention forward / backward, the dq_r stream kernel, the register-resident LayerNorm -- with the stated
    bf16 tolerance, plus assertions that those kernels are what the dispatchers pick at these shapes (a silent fall-back to the
    small-tile kernels would make this test a repeat of the tiny-model ones).

8 x 1024 tokens is the smallest batch at which every in-step GEMM of a layer has the >= 160 output tiles the 256 x 256 kernels are
used for.  The oracle needs ~1 minute of host time for it (one forward + backward, shared by both tests)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import db1_oracle as O  # noqa: E402

DEV = "cuda"
B, L, D_MODEL, N_HEAD, N_LAYER = 8, 1024, 2048, 16, 2
GRAD_NAMES = ["h.0.dec_attn.qkv_net.weight", "h.1.dec_attn.o_net.weight", "h.0.dec_attn.r_net.weight", "h.1.pos_ff.CoreNet.0.weight",
              "h.0.pos_ff.CoreNet.0.bias", "h.1.pos_ff.CoreNet.2.weight", "h.0.pos_ff.CoreNet.2.bias", "h.0.dec_attn.layer_norm.weight",
              "h.1.pos_ff.layer_norm.bias", "r_w_bias", "r_r_bias", "word_embedding.weight"]


def _cfg():
    from bdm_db1_amd import synth
    return synth.db1_config("1.3B", n_layer=N_LAYER)


def _params(cfg):
    rng = np.random.default_rng(2024)
    f, d, H = np.float32, cfg.n_embed, cfg.n_head
    V = cfg.text_vocab_size + cfg.num_continuous_bin + 1
    p = {"r_w_bias": (rng.standard_normal((H, d // H)) * 0.02).astype(f), "r_r_bias": (rng.standard_normal((H, d // H)) * 0.02).astype(f),
         "word_embedding.weight": (rng.standard_normal((V, d)) * 0.02).astype(f)}
    for i in range(cfg.n_layer):
        q = f"h.{i}."
        p[q + "dec_attn.qkv_net.weight"] = (rng.standard_normal((3 * d, d)) * 0.02).astype(f)
        p[q + "dec_attn.o_net.weight"] = (rng.standard_normal((d, d)) * 0.02).astype(f)
        p[q + "dec_attn.r_net.weight"] = (rng.standard_normal((d, d)) * 0.02).astype(f)
        p[q + "pos_ff.CoreNet.0.weight"] = (rng.standard_normal((4 * d, d)) * 0.02).astype(f)
        p[q + "pos_ff.CoreNet.0.bias"] = (rng.standard_normal(4 * d) * 0.02).astype(f)
        p[q + "pos_ff.CoreNet.2.weight"] = (rng.standard_normal((d, 2 * d)) * 0.02).astype(f)
        p[q + "pos_ff.CoreNet.2.bias"] = (rng.standard_normal(d) * 0.02).astype(f)
        for ln in ("dec_attn.layer_norm", "pos_ff.layer_norm"):
            p[q + ln + ".weight"] = (1 + 0.1 * rng.standard_normal(d)).astype(f)
            p[q + ln + ".bias"] = (0.05 * rng.standard_normal(d)).astype(f)
    return p


def _batch(cfg):
    rng = np.random.default_rng(77)
    ids = rng.integers(0, cfg.text_vocab_size, (B, L + 1))
    mask = (rng.random((B, L)) > 0.1).astype(np.float32)
    return ids[:, :-1].copy(), ids[:, 1:].copy(), mask


@pytest.fixture(scope="module")
def reference():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg = _cfg()
    params = _params(cfg)
    text, label, mask = _batch(cfg)
    ocfg = {k: getattr(cfg, k) for k in O.OracleConfig.__dataclass_fields__ if hasattr(cfg, k)}
    oracle = O.OracleModel(O.OracleConfig(**ocfg), params, dtype=np.float32)
    logits, loss, _ = oracle.forward([O.TaskBatch(kind="nlp", text_seq=text, label=label, loss_mask=mask)])
    grads = oracle.backward()
    keep = {n: grads[n].astype(np.float32) for n in GRAD_NAMES}
    return SimpleNamespace(cfg=cfg, params=params, text=text, label=label, mask=mask, logits=logits.astype(np.float32), loss=float(loss), grads=keep)


def _run(ref, dtype):
    from bdm_db1_amd import TransformerXL
    from bdm_db1_amd.data import NLPTaskInput
    model = TransformerXL(ref.cfg, compute_dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in ref.params.items()}, strict=False)
    model.eval()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=T(ref.mask), label=T(ref.label), text_seq=T(ref.text), text_len=None)
    with torch.enable_grad():
        logits, loss = model([x])
    lg = logits.float().cpu().numpy()
    model.backward()
    return model, lg, float(loss)


def _rel(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max() / (np.abs(want).max() + 1e-30))


def _l2(got, want):
    g, w = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.sqrt(((g - w) ** 2).sum()) / (np.sqrt((w ** 2).sum()) + 1e-30))


def test_fp32_two_layers_of_db1_1p3b_match_the_oracle(reference):
    model, lg, loss = _run(reference, torch.float32)
    e = _rel(lg, reference.logits)
    assert e < 1e-3, f"logits rel err {e:.2e} (north_star gate: 1e-3)"
    assert abs(loss - reference.loss) < 1e-4 * abs(reference.loss)
    assert _l2(lg, reference.logits) < 1e-3
    for n in GRAD_NAMES:
        g = model.G(n).cpu().numpy()
        ge, ge2 = _rel(g, reference.grads[n]), _l2(g, reference.grads[n])
        assert ge < 2e-3 and ge2 < 2e-3, f"{n}: gradient rel err: max-norm {ge:.2e}, L2 {ge2:.2e}"


def test_bf16_default_dispatch_at_db1_1p3b_geometry(reference):
    from bdm_db1_amd import ops
    T, d = B * L, D_MODEL
    bf = lambda *s: torch.empty(*s, device=DEV, dtype=torch.bfloat16)
    # ---- what the dispatchers pick at these shapes: the 4-wave kernels in all three operand layouts, not a small-tile fall-back
    x, dy = bf(T, d), bf(T, 4 * d)
    w_ff1, w_ff2, w_o = bf(4 * d, d), bf(d, 2 * d), bf(d, d)
    assert ops.gemm_kernel_choice(x, w_ff1.t(), bf(T, 4 * d))[0] == "w4"                                 # NT: y = x W^T        (ff1)
    assert ops.gemm_kernel_choice(bf(T, 2 * d), w_ff2.t(), bf(T, d))[0] == "w4"                          # NT                   (ff2)
    assert ops.gemm_kernel_choice(x, w_o.t(), bf(T, d))[0] == "w4"                                       # NT                   (o_net)
    assert ops.gemm_kernel_choice(dy, w_ff1, bf(T, d), beta=1.0)[0] == "w4"                              # NN: dx = dy W, beta  (dff1)
    assert ops.gemm_kernel_choice(dy.t(), x, torch.empty(4 * d, d, device=DEV), beta=1.0)[0] == "w4"     # TN: dW = dy^T x      (wff1)
    k, sk, _ = ops.gemm_kernel_choice(bf(T, d).t(), x, torch.empty(d, d, device=DEV), beta=1.0)          # o_net dW: 64 tiles -> split-K
    assert sk and k in ("w4", "pp-k32")
    assert ops.gemm_nt_headbias_supported(T, 3 * d, d, d) and ops.relattn_flash_supported(B, L, N_HEAD, d // N_HEAD, torch.bfloat16)
    assert ops.relattn_dqr_supported(B, L, N_HEAD, d // N_HEAD, torch.bfloat16)
    # ---- the model through them.  Stated bf16 tolerance: logits 3e-2 of max |logit| and 2e-2 relative L2, loss 2e-2 abs, gradients 6e-2 of
    # each tensor's max and 5e-2 relative L2 (the limits of the vision-geometry tests)
    model, lg, loss = _run(reference, torch.bfloat16)
    assert model.use_flash and model.use_flash_bwd and model.use_headbias_epilogue
    e, e2 = _rel(lg, reference.logits), _l2(lg, reference.logits)
    assert e < 3e-2 and e2 < 2e-2, f"bf16 logits rel err: max-norm {e:.2e}, L2 {e2:.2e}"
    assert abs(loss - reference.loss) < 2e-2
    for n in GRAD_NAMES:
        g = model.G(n).cpu().numpy()
        ge, ge2 = _rel(g, reference.grads[n]), _l2(g, reference.grads[n])
        assert ge < 6e-2 and ge2 < 5e-2, f"{n}: bf16 gradient rel err: max-norm {ge:.2e}, L2 {ge2:.2e}"
