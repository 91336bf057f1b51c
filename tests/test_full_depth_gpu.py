"""Parity at FULL DEPTH: all 24 decoder layers of DB1-1.3B -- d = 2048, 16 heads of 128, GEGLU 8192 -> 4096, post-LN,
tied 33 025-row head, L = 1024 -- so that error growth over 24 post-LN layers is measured, not assumed from the 2-layer geometry test.

  * fp32 instantiation of the HIP path against the fp32 CPU oracle on ONE 1024-token sequence (the oracle needs ~35 s of host time for
    forward + backward at this depth): logits within 1e-3 relative -- north_star's gate --, the loss, and gradients from the first, a
    middle and the last layer plus the shared u / v and the tied embedding (every one of them has the whole depth behind or in front of it);
  * the bf16 training path with the default dispatch against the fp32 HIP run: end to end on 8 sequences (loose limits: the randomly
    initialised 24-layer network amplifies perturbations, measured in tools/exp/depth_error.py) and, sharply, EVERY layer teacher-forced
    with the fp32 run's input and output gradient (relative L2: activations 1.5e-2, gradients 3e-2, no trend with depth).
"""
import json
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
L, N_LAYER = 1024, 24
GRAD_NAMES = ["h.0.dec_attn.qkv_net.weight", "h.0.pos_ff.CoreNet.0.weight", "h.11.dec_attn.o_net.weight", "h.12.pos_ff.CoreNet.2.weight",
              "h.23.dec_attn.r_net.weight", "h.23.pos_ff.CoreNet.0.bias", "h.5.dec_attn.layer_norm.weight", "h.17.pos_ff.layer_norm.bias",
              "r_w_bias", "r_r_bias", "word_embedding.weight"]


def _cfg():
    from bdm_db1_amd import synth
    return synth.db1_config("1.3B", n_layer=N_LAYER)


def _params(cfg):
    """N(0, 0.02) weights as the reference's init (transformer_xl.py:456-468), non-trivial LayerNorm parameters and biases"""
    rng = np.random.default_rng(424242)
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


def _batch(cfg, B):
    rng = np.random.default_rng(99)
    ids = rng.integers(0, cfg.text_vocab_size, (B, L + 1))
    mask = (rng.random((B, L)) > 0.1).astype(np.float32)
    return ids[:, :-1].copy(), ids[:, 1:].copy(), mask


def _run(cfg, params, batch, dtype):
    from bdm_db1_amd import TransformerXL
    from bdm_db1_amd.data import NLPTaskInput
    text, label, mask = batch
    model = TransformerXL(cfg, compute_dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    model.eval()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=T(mask), label=T(label), text_seq=T(text), text_len=None)
    with torch.enable_grad():
        logits, loss = model([x])
    lg = logits.float().cpu().numpy()
    model.backward()
    grads = {n: model.G(n).float().cpu().numpy().copy() for n in GRAD_NAMES}
    loss = float(loss)
    del model, logits
    torch.cuda.empty_cache()
    return lg, loss, grads


def _maxrel(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max() / (np.abs(want).max() + 1e-30))


def _l2rel(got, want):
    g, w = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.sqrt(((g - w) ** 2).sum()) / (np.sqrt((w ** 2).sum()) + 1e-30))


def _record(name, rec):
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", name), "w") as f:
            json.dump(rec, f, indent=1)
    except OSError:
        pass


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg = _cfg()
    return SimpleNamespace(cfg=cfg, params=_params(cfg))


def test_fp32_all_24_layers_match_the_oracle_on_one_sequence(setup):
    cfg, params = setup.cfg, setup.params
    batch = _batch(cfg, 1)
    ocfg = {k: getattr(cfg, k) for k in O.OracleConfig.__dataclass_fields__ if hasattr(cfg, k)}
    oracle = O.OracleModel(O.OracleConfig(**ocfg), params, dtype=np.float32)
    ref_logits, ref_loss, _ = oracle.forward([O.TaskBatch(kind="nlp", text_seq=batch[0], label=batch[1], loss_mask=batch[2])])
    ref_grads = oracle.backward()
    lg, loss, grads = _run(cfg, params, batch, torch.float32)
    rec = {"logits_maxrel": _maxrel(lg, ref_logits), "logits_l2rel": _l2rel(lg, ref_logits), "loss": loss, "oracle_loss": float(ref_loss),
           "grads": {n: [_maxrel(grads[n], ref_grads[n]), _l2rel(grads[n], ref_grads[n])] for n in GRAD_NAMES}}
    _record("full_depth_fp32.json", rec)
    assert rec["logits_maxrel"] < 1e-3, f"logits rel err {rec['logits_maxrel']:.2e} at 24 layers (north_star gate: 1e-3)"
    assert rec["logits_l2rel"] < 1e-3, rec
    assert abs(loss - ref_loss) < 1e-4 * abs(ref_loss)
    for n in GRAD_NAMES:
        assert rec["grads"][n][0] < 3e-3 and rec["grads"][n][1] < 3e-3, f"{n}: gradient rel err (max-norm, L2) {rec['grads'][n]}"


def test_bf16_all_24_layers_against_the_fp32_hip_run(setup):
    """End to end, bf16 against the fp32 HIP run on 8 sequences.  What this measures at the reference's N(0, 0.02) initialisation is mostly
    the NETWORK: 24 post-LN layers whose sub-layer outputs are half the size of the residual stream amplify any perturbation (the
    feed-forward Jacobian gain exceeds its signal gain), so the 0.6 % that one bf16 layer contributes (tools/exp/depth_error.py: 0.6 / 0.9 /
    1.4 / 2.6 / 5.3 / 16 % relative L2 on the logits at 1 / 2 / 4 / 8 / 16 / 24 layers) grows faster than sqrt(depth).  The limits are
    therefore loose here (logits and gradients 0.35 relative L2, loss 2e-2 abs) and the sharp statement about the KERNELS at every depth is
    the teacher-forced test below."""
    cfg, params = setup.cfg, setup.params
    batch = _batch(cfg, 8)
    lg32, loss32, g32 = _run(cfg, params, batch, torch.float32)
    lg16, loss16, g16 = _run(cfg, params, batch, torch.bfloat16)
    rec = {"logits_maxrel": _maxrel(lg16, lg32), "logits_l2rel": _l2rel(lg16, lg32), "loss_bf16": loss16, "loss_fp32": loss32,
           "grads": {n: [_maxrel(g16[n], g32[n]), _l2rel(g16[n], g32[n])] for n in GRAD_NAMES}}
    _record("full_depth_bf16.json", rec)
    assert rec["logits_l2rel"] < 0.35, rec
    assert abs(loss16 - loss32) < 2e-2, rec
    for n in GRAD_NAMES:
        assert rec["grads"][n][1] < 0.35, f"{n}: bf16 gradient rel err (max-norm, L2) {rec['grads'][n]}"


def _layer_by_layer(cfg, params, batch, dtype, forced=None):
    """the decoder stack driven layer by layer through the model's own per-layer functions.  ``forced`` = (inputs, douts) of an earlier
    fp32 run: every layer then gets THAT run's input (forward) and output gradient (backward) cast to ``dtype`` -- teacher forcing --, so a
    layer's result is compared without whatever the layers before it accumulated.  Returns per-layer inputs, outputs, douts, dxs and the
    five weight gradients of every layer (float32 CPU arrays)."""
    from bdm_db1_amd import TransformerXL, ops
    text = batch[0]
    model = TransformerXL(cfg, compute_dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    model.eval()
    B, L, d = text.shape[0], text.shape[1], cfg.n_embed
    ids = torch.from_numpy(text).to(DEV).reshape(-1)
    R_in = model._sinusoid(L)
    shift = model._window(L, 0)
    names = ["dec_attn.qkv_net.weight", "dec_attn.o_net.weight", "dec_attn.r_net.weight", "pos_ff.CoreNet.0.weight", "pos_ff.CoreNet.2.weight"]
    ins, outs, douts, dxs, wgs = [], [], [], [], []
    with torch.cuda.device(model.dev), ops.stream_scope():
        x = model._new(B * L, d)
        ops.embed_gather(model.W("word_embedding.weight"), ids, x)
        ctxs = []
        for i in range(cfg.n_layer):
            if forced is not None:
                x = torch.from_numpy(forced[0][i]).to(DEV).to(dtype)
            ins.append(x.float().cpu().numpy())
            x, c = model._layer_fwd(i, x, R_in, B, L, 0, shift, None, True)
            outs.append(x.float().cpu().numpy())
            ctxs.append(c)
        g = torch.Generator(device=DEV)
        g.manual_seed(31)
        dh = (torch.randn(B * L, d, device=DEV, generator=g) * 1e-3).to(dtype)
        model._gb = 0.0
        for i in reversed(range(cfg.n_layer)):
            if forced is not None:
                dh = torch.from_numpy(forced[1][i]).to(DEV).to(dtype)
            douts.append(dh.float().cpu().numpy())
            dh = model._layer_bwd(i, dh, ctxs[i], R_in, B, L, shift)
            ctxs[i] = None
            dxs.append(dh.float().cpu().numpy())
            wgs.append({n: model.G(f"h.{i}.{n}").float().cpu().numpy().copy() for n in names})
    douts.reverse(); dxs.reverse(); wgs.reverse()
    del model
    torch.cuda.empty_cache()
    return ins, outs, douts, dxs, wgs


def test_bf16_every_layer_teacher_forced_against_fp32(setup):
    """All 24 layers of DB1-1.3B with their own weights, bf16 default dispatch against the fp32 HIP path, each layer fed the fp32 run's
    input and output gradient: the layer output, the input gradient and the five weight gradients of EVERY layer within the bf16 tolerance
    of ONE layer (relative L2: activations 1.5e-2, gradients 3e-2), and no trend with depth -- the kernels behave the same at layer 23 as
    at layer 0; what grows in the end-to-end comparison above is the network's own sensitivity."""
    cfg, params = setup.cfg, setup.params
    batch = _batch(cfg, 4)
    i32, o32, d32, x32, w32 = _layer_by_layer(cfg, params, batch, torch.float32)
    _, o16, _, x16, w16 = _layer_by_layer(cfg, params, batch, torch.bfloat16, forced=(i32, d32))
    rec = {"out": [], "dx": [], "wgrad": []}
    for i in range(cfg.n_layer):
        rec["out"].append(_l2rel(o16[i], o32[i]))
        rec["dx"].append(_l2rel(x16[i], x32[i]))
        rec["wgrad"].append(max(_l2rel(w16[i][n], w32[i][n]) for n in w32[i]))
    _record("full_depth_teacher_forced.json", rec)
    assert max(rec["out"]) < 1.5e-2, rec["out"]
    assert max(rec["dx"]) < 3e-2, rec["dx"]
    assert max(rec["wgrad"]) < 3e-2, rec["wgrad"]
    first, last = np.mean(rec["out"][:6]), np.mean(rec["out"][-6:])
    assert last < 2.0 * first + 1e-3, (first, last)
