"""Pins the CPU oracle (oracle/db1_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only, no GPU needed."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import db1_oracle as O  # noqa: E402
from golden_util import CASES, MEM_CASES, case_cfg, make_params, make_batch, sample_idx  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def build(name, seed):
    cfg = case_cfg(name)
    params = make_params(cfg, seed)
    gold = load("model_" + name)
    params["pos_emb.inv_freq"] = gold["inv_freq"]
    model = O.OracleModel(O.OracleConfig(**cfg), params)
    return cfg, params, gold, model


def to_tasks(tasks):
    return [O.TaskBatch(**t) for t in tasks]


TRAIN_CASES = [n for n in CASES if n not in MEM_CASES]


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_model_forward_backward_matches_reference(name):
    seed = 100 + list(CASES).index(name)
    cfg, params, gold, model = build(name, seed)
    logits, loss, _ = model.forward(to_tasks(make_batch(name, cfg, seed)))
    assert tuple(gold["logits_shape"]) == logits.shape
    # reference is fp32; oracle fp64: tolerance = fp32 round-off of the reference
    assert abs(loss - gold["loss"]) < 2e-6 * max(1.0, abs(gold["loss"]))
    ls = logits.reshape(-1)[sample_idx(logits.size, 4096)]
    np.testing.assert_allclose(ls, gold["logits_sample"], rtol=2e-4, atol=2e-5)
    assert abs(np.sqrt((logits ** 2).sum()) - gold["logits_norm"]) < 1e-5 * gold["logits_norm"]
    grads = model.backward()
    checked = 0
    for k in gold:
        if not k.startswith("gnorm/"):
            continue
        n = k[len("gnorm/"):]
        g = grads.get(n)
        if g is None:
            assert gold[k] == 0.0, n
            continue
        gn = np.sqrt((g ** 2).sum())
        assert abs(gn - gold[k]) <= 3e-5 * max(gold[k], 1e-6) + 1e-9, (n, gn, gold[k])
        gs = g.reshape(-1)[sample_idx(g.size)]
        scale = max(np.abs(gold["gsample/" + n]).max(), 1e-8)
        assert np.abs(gs - gold["gsample/" + n]).max() <= 1e-4 * scale + 1e-9, n
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize("name", list(MEM_CASES))
def test_model_with_memory_matches_reference(name):
    cfg, params, gold, model = build(name, 100 + list(CASES).index(name))
    B, ml, d = 2, cfg["mem_len"], cfg["n_embed"]
    mems = [np.zeros((B, ml, d)) for _ in range(cfg["n_layer"])]
    for step in range(len(MEM_CASES[name])):
        ids = gold[f"ids{step}"]
        logits, loss, mems = model.forward([O.TaskBatch(kind="nlp", text_seq=ids)], compute_loss=False, mems=mems)
        assert loss is None
        np.testing.assert_allclose(logits, gold[f"logits{step}"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(mems[-1], gold[f"mem_last{step}"], rtol=2e-4, atol=2e-5)


def test_patch_embedder_matches_reference():
    gold = load("patch_embed")
    cfg = case_cfg("small_mixed")
    params = {k: v.astype(np.float64) for k, v in make_params(cfg, 7).items()}
    y, cache = O.patch_embed_fwd(params, gold["img"].astype(np.float64), 16)
    np.testing.assert_allclose(y, gold["y"], rtol=1e-4, atol=2e-5)
    grads = O.patch_embed_bwd(params, gold["G"].astype(np.float64), cache)
    for k in gold:
        if k.startswith("grad/"):
            g = grads["vision_encoder.patch_embeddings." + k[5:]]
            scale = np.abs(gold[k]).max()
            assert np.abs(g - gold[k]).max() <= 2e-4 * scale, k


def test_vision_position_ids_eval():
    # vision_embedding.py:134-148 with n=14 -> known midpoint ids
    r, c = O.vision_position_ids_eval(2, 3, 128)
    assert r.tolist() == [32, 32, 32, 96, 96, 96]
    assert c.tolist() == [21, 63, 106, 21, 63, 106]


def test_scalar_tokenizer_matches_reference():
    gold = load("scalar_tokenizer")
    assert gold["known_obs_ids"].tolist() == [0, 279, 477, 512, 546, 710, 744, 860, 1023, 1023]
    assert gold["known_act_ids"].tolist() == [0, 256, 512, 768, 1023, 1023]
    assert (O.mulaw_discretize(gold["known_obs"], False) == gold["known_obs_ids"]).all()
    assert (O.mulaw_discretize(gold["known_act"], True) == gold["known_act_ids"]).all()
    assert (O.mulaw_discretize(gold["act"], True) == gold["act_ids"]).all()
    ids = O.mulaw_discretize(gold["obs"], False)
    bad = np.nonzero(ids != gold["obs_ids"])[0]
    assert len(bad) == 0, (len(bad), gold["obs"][bad][:10], ids[bad][:10], gold["obs_ids"][bad][:10])
    np.testing.assert_allclose(O.mulaw_decode(gold["dec_ids"], False), gold["dec_obs"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(O.mulaw_decode(gold["dec_ids"], True), gold["dec_act"], rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("mode", ["adam", "adamw"])
def test_adam_matches_torch(mode):
    gold = load("adam")
    p = gold[f"{mode}/p0"].astype(np.float64)
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for i in range(3):
        g = gold[f"{mode}/g{i}"].astype(np.float64)
        norm = np.sqrt((g ** 2).sum())
        assert abs(norm - gold[f"{mode}/norm{i}"]) < 1e-5 * norm
        p, m, v = O.adam_step(p, g, m, v, i + 1, 3e-3, wd=0.01, adamw=(mode == "adamw"), grad_scale=O.clip_coef(norm, 1.0))
        np.testing.assert_allclose(p, gold[f"{mode}/p{i + 1}"], rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(m, gold[f"{mode}/m{i + 1}"], rtol=2e-6, atol=1e-8)
        np.testing.assert_allclose(v, gold[f"{mode}/v{i + 1}"], rtol=2e-6, atol=1e-10)


def test_scheduler_matches_reference():
    gold = load("scheduler")
    for style in ("constant", "linear", "cosine"):
        for wstyle in ("constant", "linear", "cosine"):
            lr = [O.lr_at(int(s), 1e-3, 1e-5, 10, 100, style) for s in gold["steps"]]
            wd = [O.wd_at(int(s), 0.01 if wstyle == "constant" else 0.0, 0.01, 80, wstyle) for s in gold["steps"]]
            np.testing.assert_allclose(lr, gold[f"lr/{style}/{wstyle}"], rtol=1e-12)
            np.testing.assert_allclose(wd, gold[f"wd/{style}/{wstyle}"], rtol=1e-12)


def test_rl_packing_matches_reference():
    gold = load("rl_packing")
    i = 0
    while f"args{i}" in gold:
        f, p = O.rl_action_flag_and_position_id(*[int(x) for x in gold[f"args{i}"]])
        assert (f == gold[f"flag{i}"]).all() and (p == gold[f"pos{i}"]).all(), i
        i += 1
    assert i >= 5
    assert (O.truncate_or_pad(gold["pad_in"], 8) == gold["pad8"]).all()
    assert (O.truncate_or_pad(gold["pad_in"], 3) == gold["pad3"]).all()


def test_param_count_1p3b():
    cfg = O.OracleConfig(n_embed=2048, n_layer=24, n_head=16, n_position=1024, mem_len=1024)
    assert O.count_params(cfg) == 1_210_585_216  # SURVEY.md section 6 (meta-device instantiation of the reference)


@pytest.mark.parametrize("name", ["tiny_nlp", "small_window"])
def test_torch_cpu_restatement_matches_reference(name):
    """oracle/db1_torch_cpu.py -- the text path in torch eager ops with autograd, the second CPU baseline bench.py times -- against the
    REFERENCE's golden vectors (fp32 like the reference) and against the NumPy oracle on the same inputs"""
    torch = pytest.importorskip("torch")
    from oracle.db1_torch_cpu import TorchCpuModel
    seed = 100 + list(CASES).index(name)
    cfg, params, gold, model = build(name, seed)
    t = make_batch(name, cfg, seed)[0]
    assert t["kind"] == "nlp"
    tm = TorchCpuModel(O.OracleConfig(**cfg), params, dtype=torch.float32)
    logits, loss = tm.forward(t["text_seq"], t["label"], t["loss_mask"])
    lg = logits.detach().numpy()
    assert tuple(gold["logits_shape"]) == lg.shape
    assert abs(float(loss) - gold["loss"]) < 5e-6 * max(1.0, abs(gold["loss"]))
    np.testing.assert_allclose(lg.reshape(-1)[sample_idx(lg.size, 4096)], gold["logits_sample"], rtol=5e-4, atol=5e-5)
    grads = tm.backward()
    ref_logits, ref_loss, _ = model.forward(to_tasks([t]))
    ref_grads = model.backward()
    assert np.abs(lg - ref_logits).max() <= 2e-5 * np.abs(ref_logits).max()
    checked = 0
    for k, g in ref_grads.items():
        if k not in grads:
            continue
        assert np.abs(grads[k] - g).max() <= 2e-4 * max(np.abs(g).max(), 1e-8) + 1e-9, k
        if "gnorm/" + k in gold:
            gn = np.sqrt((grads[k].astype(np.float64) ** 2).sum())
            assert abs(gn - gold["gnorm/" + k]) <= 2e-4 * max(gold["gnorm/" + k], 1e-6) + 1e-9, k
            checked += 1
    assert checked >= 10
