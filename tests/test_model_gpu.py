"""End-to-end parity of the HIP model (through the C ABI) against the CPU oracle and against the golden
vectors captured from the reference: logits, loss, every parameter gradient, parameters after Adam steps,
and inference with Transformer-XL memory.  fp32 gate: 1e-3 relative on logits (north_star), in practice ~1e-5;
bf16 runs are compared with a stated looser tolerance."""
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
from golden_util import CASES, MEM_CASES, case_cfg, make_params, make_batch, sample_idx  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def build(name, compute_dtype=torch.float32):
    from bdm_db1_amd import TransformerXL
    seed = 100 + list(CASES).index(name)
    cfg = case_cfg(name)
    params = make_params(cfg, seed)
    gold = dict(np.load(os.path.join(G, f"model_{name}.npz")))
    params["pos_emb.inv_freq"] = gold["inv_freq"]
    model = TransformerXL(SimpleNamespace(**cfg), compute_dtype=compute_dtype)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith("ic_encoder.") or ".dec_attn.r_" in m for m in missing), missing
    model.eval()
    oracle = O.OracleModel(O.OracleConfig(**cfg), params)
    return cfg, params, gold, model, oracle, seed


def to_inputs(tasks):
    from bdm_db1_amd.data import NLPTaskInput, RLTaskInput, ICTaskInput, VQATaskInput
    T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    out = []
    for t in tasks:
        base = dict(position_id=T(t.get("position_id")), attention_mask=None, loss_mask=T(t.get("loss_mask")), label=T(t.get("label")))
        if t["kind"] == "nlp":
            out.append(NLPTaskInput(text_seq=T(t["text_seq"]), text_len=None, **base))
        elif t["kind"] == "rl":
            out.append(RLTaskInput(text_seq=None, vision_seq=T(t["vision_seq"]), tensor_seq=T(t["tensor_seq"]), **base))
        elif t["kind"] == "vqa":   # _forward_vqa (transformer_xl.py:705-748)
            out.append(VQATaskInput(prompt_seq=T(t["prompt_seq"]), img_seq=T(t["img_seq"]), text_seq=T(t["text_seq"]), img_id_seq=None,
                                    ques_id_seq=None, ques_len=T(t["ques_len"]), **base))
        else:
            out.append(ICTaskInput(prompt_seq=T(t["prompt_seq"]), img_seq=T(t["img_seq"]), text_seq=T(t["text_seq"]), img_id_seq=None, **base))
    return out


def rel_err(got, ref):
    got = got.detach().to(torch.float64).cpu().numpy() if hasattr(got, "detach") else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)


TRAIN_CASES = [n for n in CASES if n not in MEM_CASES]


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_fp32_forward_backward_parity(name):
    cfg, params, gold, model, oracle, seed = build(name)
    tasks = make_batch(name, cfg, seed)
    with torch.enable_grad():
        logits, loss = model(to_inputs(tasks))
    ref_logits, ref_loss, _ = oracle.forward([O.TaskBatch(**t) for t in tasks])
    assert tuple(logits.shape) == ref_logits.shape == tuple(gold["logits_shape"])
    # north_star tolerance: logits within 1e-3 rel of the CPU reference (fp32 gate).  Measured: ~1e-5.
    e = rel_err(logits, ref_logits)
    assert e < 1e-4, f"logits rel err {e:.2e}"
    assert abs(float(loss) - ref_loss) < 2e-5 * max(1.0, abs(ref_loss))
    # against the reference's own numbers (golden), same tolerance
    lg = logits.detach().float().cpu().numpy().reshape(-1)[sample_idx(logits.numel(), 4096)]
    assert np.abs(lg - gold["logits_sample"]).max() / np.abs(gold["logits_sample"]).max() < 1e-4
    assert abs(float(loss) - float(gold["loss"])) < 2e-5 * max(1.0, abs(float(gold["loss"])))
    model.backward()
    ref_grads = oracle.backward()
    worst = ("", 0.0)
    for n in ref_grads:
        g = model.G(n)
        e = rel_err(g, ref_grads[n])
        if e > worst[1]:
            worst = (n, e)
        gs = g.detach().cpu().numpy().reshape(-1)[sample_idx(g.numel())]
        scale = max(np.abs(gold["gsample/" + n]).max(), 1e-8)
        assert np.abs(gs - gold["gsample/" + n]).max() <= 2e-3 * scale + 1e-9, f"{n} vs golden"
    assert worst[1] < 1e-3, f"worst gradient {worst[0]}: rel err {worst[1]:.2e}"
    # the vocabulary padding rows never receive gradient
    pad = model.arena.view(model.arena.grad, "word_embedding.weight", full=True).view(model.vocab_pad, -1)[model.total_vocab_size:]
    assert float(pad.abs().max()) == 0.0 if pad.numel() else True


def test_rl_label_placeholder_fix_and_caller_tensor_untouched():
    cfg, params, gold, model, oracle, seed = build("small_mixed")
    tasks = make_batch("small_mixed", cfg, seed)
    tasks[0]["label"] = tasks[0]["label"].copy()
    tasks[0]["label"][0, 0] = -1  # a label that points at an image placeholder (transformer_xl.py:644-645)
    tasks[0]["loss_mask"][0, 0] = 0.0
    inp = to_inputs(tasks)
    before = inp[0].label.clone()
    logits, loss = model(inp)
    _, ref_loss, _ = oracle.forward([O.TaskBatch(**t) for t in tasks])
    assert abs(float(loss) - ref_loss) < 2e-5 * max(1.0, abs(ref_loss))
    assert torch.equal(inp[0].label, before)


@pytest.mark.parametrize("adamw", [False, True])
def test_engine_three_adam_steps_match_oracle(adamw):
    from bdm_db1_amd import initialize
    name = "small_window"
    cfg, params, gold, model, oracle, seed = build(name)
    args = SimpleNamespace(lr=2e-3, weight_decay=0.01, clip_grad=1.0, optimizer="adamw" if adamw else "adam",
                           adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8, keep_logits=True)
    engine, opt, _, _ = initialize(args, model)
    tasks = make_batch(name, cfg, seed)
    m = {k: np.zeros_like(v, dtype=np.float64) for k, v in oracle.p.items()}
    v = {k: np.zeros_like(val, dtype=np.float64) for k, val in oracle.p.items()}
    engine.train()
    for step in range(1, 4):
        logits, loss = engine(to_inputs(tasks))
        engine.backward(loss)
        engine.step()
        _, ref_loss, _ = oracle.forward([O.TaskBatch(**t) for t in tasks])
        assert abs(float(loss) - ref_loss) < 5e-5 * max(1.0, abs(ref_loss)), step
        grads = oracle.backward()
        norm = np.sqrt(sum((g ** 2).sum() for g in grads.values()))
        coef = O.clip_coef(norm, 1.0)
        for k in oracle.p:
            g = grads.get(k, np.zeros_like(oracle.p[k]))
            oracle.p[k], m[k], v[k] = O.adam_step(oracle.p[k], g, m[k], v[k], step, 2e-3, wd=0.01, adamw=adamw, grad_scale=coef)
    sd = model.state_dict()
    worst = max(rel_err(sd[k], oracle.p[k]) for k in oracle.p)
    assert worst < 2e-4, worst
    # after the step: the scattered accumulators are cleared (one launch over a segment table); the weight gradients are NOT -- the next
    # backward's GEMMs write them with beta = 0 (which the three matching steps above depend on)
    assert model._grad_fresh
    big = set(model.gemm_first_grads())
    for n in model.arena.offsets:
        if n not in big:
            assert float(model.G(n).abs().max()) == 0.0, n
    assert float(model.G("h.0.dec_attn.qkv_net.weight").abs().max()) > 0.0
    # a second backward without a step in between accumulates (gradient accumulation): twice the single gradient
    model.zero_grad()
    model(to_inputs(tasks))[1]
    model.backward()
    g1 = model.G("h.1.pos_ff.CoreNet.0.weight").clone()
    model(to_inputs(tasks))
    model.backward()
    assert rel_err(model.G("h.1.pos_ff.CoreNet.0.weight"), 2 * g1.double().cpu().numpy()) < 1e-5


@pytest.mark.parametrize("name", list(MEM_CASES))
def test_inference_with_memory_matches_reference_golden(name):
    cfg, params, gold, model, oracle, seed = build(name)
    from bdm_db1_amd.data import NLPTaskInput
    mems = model.init_mem(2)
    assert len(mems) == cfg["n_layer"] and tuple(mems[0].shape) == (2, cfg["mem_len"], cfg["n_embed"])
    with torch.no_grad():
        for step in range(len(MEM_CASES[name])):
            ids = torch.from_numpy(gold[f"ids{step}"]).to(DEV)
            x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
            logits, loss, mems = model([x], compute_loss=False, mems=mems)
            assert loss is None
            assert rel_err(logits, gold[f"logits{step}"]) < 1e-4
            assert rel_err(mems[-1], gold[f"mem_last{step}"]) < 1e-4


@pytest.mark.parametrize("edit_mems_in_place", [False, True])
def test_bf16_kv_cached_decode_matches_reference_golden(edit_mems_in_place):
    """The bf16 K/V-cached decode path (skinny projections, fused decode attention, cached keys / values; d_head = 128 like DB1-1.3B)
    against the REFERENCE's own memory run (model_mems_d128.npz: calls of 6 / 1 / 1 / 9 / 1 tokens), not against another HIP path.
    Stated tolerance: logits 3e-2 of max |logit|, memory 3e-2 (bf16 storage of activations and weights).
    With ``edit_mems_in_place`` the caller touches the returned memory between calls (`mems[i] *= 1`: same values, same tensor
    identity, new version counter): the cache must notice and rebuild itself from the hidden states instead of trusting identity."""
    name = "mems_d128"
    cfg, params, gold, model, oracle, seed = build(name, compute_dtype=torch.bfloat16)
    from bdm_db1_amd.data import NLPTaskInput
    assert model.d_head == 128 and model.use_decode
    mems = model.init_mem(2)
    rebuilt = 0
    with torch.no_grad():
        for step in range(len(MEM_CASES[name])):
            ids = torch.from_numpy(gold[f"ids{step}"]).to(DEV)
            x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
            st_before = model._dec_state
            logits, loss, mems = model([x], compute_loss=False, mems=mems)
            assert model._dec_state is not None, "the fused decode path did not run"
            if step > 0 and edit_mems_in_place:
                assert st_before is not None and any(m._version != v for m, v in zip(st_before.mems, st_before.mem_versions))
                rebuilt += 1
            assert rel_err(logits, gold[f"logits{step}"]) < 3e-2, step
            assert rel_err(mems[-1], gold[f"mem_last{step}"]) < 3e-2, step
            if edit_mems_in_place:
                for m in mems:
                    m.mul_(1.0)
    assert rebuilt == (len(MEM_CASES[name]) - 1 if edit_mems_in_place else 0)


def test_mask_edge_cases_follow_the_reference():
    """transformer_xl.py:177,205-206,551-567.  (a) `mem_len = 0` under same_length hides EVERY key: the reference does not raise, it
    attends uniformly (the golden case small_memlen0 pins forward and backward; here: no exception, finite loss).  (b) a call whose
    mask hides NOTHING raises ValueError in the reference: one query token without a full memory."""
    from bdm_db1_amd.data import NLPTaskInput
    cfg, params, gold, model, oracle, seed = build("small_memlen0")
    tasks = make_batch("small_memlen0", cfg, seed)
    logits, loss = model(to_inputs(tasks))
    assert np.isfinite(float(loss)) and abs(float(loss) - float(gold["loss"])) < 2e-5 * abs(float(gold["loss"]))
    cfg, params, gold, model, oracle, seed = build("small_window")
    one = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=torch.ones(1, 1, device=DEV), label=torch.zeros(1, 1, dtype=torch.long, device=DEV),
                       text_seq=torch.zeros(1, 1, dtype=torch.long, device=DEV), text_len=None)
    with pytest.raises(ValueError):
        model([one])
    with pytest.raises(ValueError):
        oracle.forward([O.TaskBatch(kind="nlp", text_seq=np.zeros((1, 1), np.int64), label=np.zeros((1, 1), np.int64), loss_mask=np.ones((1, 1), np.float32))])
    # ... but with a full memory the same_length window hides key 0, and the call goes through (evaluate_rl's 1-token calls)
    cfg, params, gold, model, oracle, seed = build("small_mems")
    with torch.no_grad():
        x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=torch.zeros(2, 1, dtype=torch.long, device=DEV), text_len=None)
        logits, _, mems = model([x], compute_loss=False, mems=model.init_mem(2))
    assert tuple(logits.shape) == (2, 1, model.total_vocab_size)


def test_fused_head_loss_engine_matches_materialised_logits():
    """keep_logits=False: head GEMM + masked CE + the head's backward run as one chunked sweep inside the forward (db1_lmhead_ce_fwd_bwd)
    and engine(batch) returns (None, loss).  Two Adam steps, with gradient accumulation over two micro-steps, must give the losses and
    parameters of the engine that materialises the logits (same kernels underneath: fp32 agreement to round-off)"""
    from bdm_db1_amd import initialize
    name = "small_window"
    res = {}
    for fused in (False, True):
        cfg, params, gold, model, oracle, seed = build(name)
        args = SimpleNamespace(lr=2e-3, weight_decay=0.01, clip_grad=1.0, optimizer="adamw", keep_logits=not fused, gradient_accumulation_steps=2)
        engine, _, _, _ = initialize(args, model)
        assert model.fuse_head_loss == fused
        tasks = make_batch(name, cfg, seed)
        engine.train()
        losses = []
        for micro in range(4):
            logits, loss = engine(to_inputs(tasks))
            assert (logits is None) == fused
            engine.backward(loss)
            engine.step()
            losses.append(float(loss))
        assert engine.global_steps == 2
        res[fused] = (losses, {k: v.detach().double().cpu().numpy() for k, v in model.state_dict().items()})
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) < 1e-5 * max(1.0, abs(b))
    for k, v in res[False][1].items():   # (the sweep adds the head's weight gradient chunk by chunk: fp32 sums in another order, then two AdamW steps)
        assert rel_err(res[True][1][k], v) < 1e-4, k


def test_bf16_text_step_is_bit_reproducible():
    """every reduction on the bf16 text path is order-fixed (split-K partials, the register-resident LayerNorm's parameter sums, bias
    column sums, the CE loss, and -- since round 2 -- the embedding-table gradient, which used float atomics): two runs of forward +
    backward on the same inputs give the same bits.  (d = 512: the generic LayerNorm kernels of the fp32 parity path still use atomics.)"""
    from bdm_db1_amd import TransformerXL
    cfg = dict(case_cfg("small_window"), n_embed=512, n_head=4, n_position=64, mem_len=64)
    params = make_params(cfg, 5)
    tasks = make_batch("small_window", cfg, 5)
    runs = []
    for _ in range(2):
        model = TransformerXL(SimpleNamespace(**cfg), compute_dtype=torch.bfloat16)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
        logits, loss = model(to_inputs(tasks))
        model.backward()
        runs.append((float(loss), logits.clone(), model.arena.grad.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    bad = [n for n in model.arena.offsets if not torch.equal(model.arena.view(runs[0][2], n), model.arena.view(runs[1][2], n))]
    assert not bad, bad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_get_action_replays_the_references_episode_calls(dtype):
    """bdm_db1_amd.evaluation.get_action (evaluate_rl.py:157-266) on the HIP model against the REFERENCE's get_action on the reference
    model (tests/golden/get_action.npz): memory mode (observation tokens, one token per call, the memorising call; also a discrete
    action space with an environment action mask) and window mode (sliding window, fixed prompt).  fp32: actions, token windows and
    the last layer's memory must match; bf16 (d_head 32: the materialised path): the same actions and windows."""
    from golden_util import GET_ACTION_CASES, get_action_inputs
    from bdm_db1_amd.evaluation import get_action
    from bdm_db1_amd.tokenizer import ContinuousScalarTokenizer
    gold = dict(np.load(os.path.join(G, "get_action.npz")))
    cfg, params, _, model, oracle, seed = build("small_mems", compute_dtype=dtype)
    tok = ContinuousScalarTokenizer(cfg["num_continuous_bin"])
    for case, (mem, disc, ol, al, steps, strat, use_prompt, lfp) in GET_ACTION_CASES.items():
        args = SimpleNamespace(overlap_with_text=cfg["overlap_with_text"], text_vocab_size=cfg["text_vocab_size"], num_discrete_values=cfg["num_discrete_values"],
                               n_position=cfg["n_position"], use_prompt=use_prompt)
        obs, prompt, masks = get_action_inputs(case, cfg)
        space = SimpleNamespace(n=6) if disc else None
        memory = model.init_mem(1) if mem else None
        seq = torch.from_numpy(prompt) if prompt is not None else torch.zeros(0, dtype=torch.long)
        with torch.no_grad():
            for st in range(steps):
                seq = torch.from_numpy(obs[st]) if mem else torch.cat([seq, torch.from_numpy(obs[st])])
                act, (seq, vis), memory = get_action(args, model, seq, None, tok, lfp, 0, ol, al, disc, space, memory, prompt_strategy=strat, action_mask=masks[st])
                assert np.array_equal(np.asarray(act, np.float64), gold[f"{case}/{st}/act"]), (case, st, act)
                assert seq.device.type == "cpu" and np.array_equal(seq.numpy(), gold[f"{case}/{st}/seq"]), (case, st)
                if mem:
                    assert rel_err(memory[-1], gold[f"{case}/{st}/mem_last"]) < (1e-4 if dtype == torch.float32 else 3e-2), (case, st)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_get_action_batched_replays_the_references_episode_in_every_row(dtype):
    """bdm_db1_amd.evaluation.get_action_batched: M = 3 environments through ONE model call per token (the batch dimension the reference's
    evaluation leaves at 1, evaluate_rl.py:452-482).  Every row replays the REFERENCE's golden episode (tests/golden/get_action.npz, memory
    cases): the same actions in every row, and the last layer's memory of every row equals the reference's."""
    from golden_util import GET_ACTION_CASES, get_action_inputs
    from bdm_db1_amd.evaluation import get_action_batched
    from bdm_db1_amd.tokenizer import ContinuousScalarTokenizer
    gold = dict(np.load(os.path.join(G, "get_action.npz")))
    cfg, params, _, model, oracle, seed = build("small_mems", compute_dtype=dtype)
    tok = ContinuousScalarTokenizer(cfg["num_continuous_bin"])
    M = 3
    for case, (mem, disc, ol, al, steps, strat, use_prompt, lfp) in GET_ACTION_CASES.items():
        if not mem:
            continue
        args = SimpleNamespace(overlap_with_text=cfg["overlap_with_text"], text_vocab_size=cfg["text_vocab_size"], num_discrete_values=cfg["num_discrete_values"],
                               n_position=cfg["n_position"], use_prompt=use_prompt)
        obs, prompt, masks = get_action_inputs(case, cfg)
        space = SimpleNamespace(n=6) if disc else None
        memory = model.init_mem(M)
        with torch.no_grad():
            for st in range(steps):
                toks = torch.from_numpy(np.tile(obs[st][None, :], (M, 1)))
                am = None if masks[st] is None else np.tile(masks[st][None, :], (M, 1))
                acts, last, memory = get_action_batched(args, model, toks, tok, ol, al, disc, space, memory, action_masks=am)
                for row in range(M):
                    assert np.array_equal(np.asarray(acts[row], np.float64).reshape(-1), np.asarray(gold[f"{case}/{st}/act"], np.float64).reshape(-1)), (case, st, row)
                    assert np.array_equal(last[row].numpy(), gold[f"{case}/{st}/seq"]), (case, st, row)
                    assert rel_err(memory[-1][row:row + 1], gold[f"{case}/{st}/mem_last"]) < (1e-4 if dtype == torch.float32 else 3e-2), (case, st, row)


def test_training_procedure_with_the_references_surface(tmp_path):
    """bdm_db1_amd.train_utils.train (the reference's train / train_step / forward_and_backward_step, src/train_utils/train.py:32-243):
    4 optimizer steps of 2 micro-steps each from an iterator of batches, losses returned per micro-step, TensorBoard-style writer calls,
    validation loss without gradient side effects, a checkpoint at the save interval that a fresh engine resumes from"""
    from bdm_db1_amd import initialize
    from bdm_db1_amd.train_utils import train, evaluate_loss
    name = "small_window"
    cfg, params, gold, model, oracle, seed = build(name)
    args = SimpleNamespace(lr=2e-3, weight_decay=0.01, clip_grad=1.0, optimizer="adamw", keep_logits=False, gradient_accumulation_steps=2,
                           iteration=0, train_iters=4, eval_interval=2, eval_iters=1, save_dir=str(tmp_path), save_interval=2)
    engine, _, _, _ = initialize(args, model)
    tasks = make_batch(name, cfg, seed)
    calls = {"train": 0, "valid": 0}

    def batches(kind):
        while True:
            calls[kind] += 1
            yield to_inputs(tasks)

    get_batch = lambda a, it: next(it)
    log = []
    writer = SimpleNamespace(add_scalar=lambda tag, v, it: log.append((tag, float(v), it)))
    first = evaluate_loss(args, engine, batches("valid"), get_batch)
    assert abs(first - float(gold["loss"])) < 1e-4 and engine.global_steps == 0 and float(model.arena.grad.abs().max()) == 0.0
    done = train(args, engine, batches("train"), batches("valid"), get_batch, sm_writer=writer)
    assert done == 4 and args.iteration == 3 and engine.global_steps == 4 and calls["train"] == 8
    tr = [v for tag, v, it in log if tag == "Train loss"]
    assert len(tr) == 4 and tr[-1] < tr[0] - 0.05                      # the same batch four times: the loss goes down
    assert [it for tag, v, it in log if tag == "Valid loss"] == [0, 2, 3]
    assert (tmp_path / "latest").read_text().strip() == "latest_model"
    cfg2, _, _, model2, _, _ = build(name)
    engine2, _, _, _ = initialize(args, model2)
    path, client = engine2.load_checkpoint(str(tmp_path), None)
    assert client["iteration"] == 4 and engine2.global_steps == 4
    for k, v in model.state_dict().items():
        assert torch.equal(model2.state_dict()[k], v), k


def test_state_dict_names_match_reference():
    cfg, params, gold, model, oracle, seed = build("small_mixed")
    names = set(model.state_dict().keys())
    want = set(params.keys())
    want |= {k.replace("vision_encoder.", "ic_encoder.") for k in params if k.startswith("vision_encoder.")}
    for i in range(cfg["n_layer"]):
        want |= {f"h.{i}.dec_attn.r_r_bias", f"h.{i}.dec_attn.r_w_bias"}
    assert names == want, (names ^ want)
    assert model.total_vocab_size == 300 + 64 + 1 and model.rl_separator_token_id == 364


def test_bf16_forward_backward_close_to_oracle():
    """bf16 storage / MFMA path on the tiny text config (uses the tile GEMMs for the tied head).
    Stated tolerance: logits 3e-2 of max |logit|, loss 2e-2 abs, gradients 6e-2 of each tensor's max."""
    cfg, params, gold, model, oracle, seed = build("tiny_nlp", compute_dtype=torch.bfloat16)
    tasks = make_batch("tiny_nlp", cfg, seed)
    logits, loss = model(to_inputs(tasks))
    ref_logits, ref_loss, _ = oracle.forward([O.TaskBatch(**t) for t in tasks])
    assert rel_err(logits, ref_logits) < 3e-2
    assert abs(float(loss) - ref_loss) < 2e-2
    model.backward()
    ref_grads = oracle.backward()
    for n in ("h.0.dec_attn.qkv_net.weight", "h.1.pos_ff.CoreNet.0.weight", "word_embedding.weight", "r_w_bias", "r_r_bias",
              "h.0.dec_attn.r_net.weight", "h.1.pos_ff.layer_norm.weight"):
        assert rel_err(model.G(n), ref_grads[n]) < 6e-2, n


def test_checkpoint_round_trip_deepspeed_layout(tmp_path):
    """save_checkpoint / load_checkpoint (checkpointing.py:17-22, evaluate_rl.py:510): <dir>/<tag>/mp_rank_00_model_states.pt with the
    weights under "module" in the reference's parameter names, `latest` tag file, optimizer moments restored; a file written the way
    DeepSpeed writes it (only "module" + client keys) loads too."""
    from bdm_db1_amd import initialize
    name = "small_window"
    cfg, params, gold, model, oracle, seed = build(name)
    args = SimpleNamespace(lr=2e-3, weight_decay=0.0, clip_grad=1.0, optimizer="adam", keep_logits=True)
    engine, _, _, _ = initialize(args, model)
    tasks = make_batch(name, cfg, seed)
    engine.train()
    for _ in range(2):
        logits, loss = engine(to_inputs(tasks))
        engine.backward(loss)
        engine.step()
    engine.save_checkpoint(str(tmp_path), tag="latest_model", client_state={"iteration": 2, "args": {"n_layer": cfg["n_layer"]}})
    assert (tmp_path / "latest").read_text().strip() == "latest_model"
    blob = torch.load(tmp_path / "latest_model" / "mp_rank_00_model_states.pt", map_location="cpu", weights_only=False)
    assert set(params.keys()) <= set(blob["module"].keys()) and blob["iteration"] == 2
    sd_before = {k: v.clone() for k, v in model.state_dict().items()}
    m_before = model.arena.exp_avg.clone()
    # a fresh model + engine restores weights, moments and the step counter, and produces the same next step
    cfg2, _, _, model2, _, _ = build(name)
    engine2, _, _, _ = initialize(args, model2)
    path, client = engine2.load_checkpoint(str(tmp_path), None)
    assert path.endswith("mp_rank_00_model_states.pt") and client["iteration"] == 2
    for k, v in sd_before.items():
        assert torch.equal(model2.state_dict()[k].cpu(), v.cpu()), k
    assert torch.equal(model2.arena.exp_avg, m_before) and engine2.global_steps == 2
    engine2.train()
    for e in (engine, engine2):
        lg, ls = e(to_inputs(tasks))
        e.backward(ls)
        e.step()
    for k in sd_before:  # (the embedding scatter-add uses fp32 atomics: not bit-reproducible between two runs, hence a tolerance)
        assert rel_err(model2.state_dict()[k], model.state_dict()[k].double().cpu().numpy()) < 1e-6, k
    # DeepSpeed-shaped file: weights only
    ds_dir = tmp_path / "ds" / "db1_870task_checkpoint"
    ds_dir.mkdir(parents=True)
    torch.save({"module": {k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}, "iteration": 7},
               ds_dir / "mp_rank_00_model_states.pt")
    cfg3, _, _, model3, _, _ = build(name)
    engine3, _, _, _ = initialize(args, model3)
    _, client3 = engine3.load_checkpoint(str(tmp_path / "ds"), "db1_870task_checkpoint")
    assert client3["iteration"] == 7
    for k, v in params.items():
        assert rel_err(model3.state_dict()[k], v) < 1e-7, k


def test_bf16_channels_last_vision_path_matches_nchw_path_and_oracle():
    """RL + caption batch in bf16: the channels-last image-patch embedder (default) against the NCHW kernels of the fp32 path, and
    both against the CPU oracle.  Stated tolerance: loss 3e-2 abs; vision-encoder gradients 8e-2 of each tensor's max between a bf16
    pipeline and the fp64 oracle, 3e-2 between the two bf16 pipelines (same maths, different rounding points)."""
    name = "small_mixed"
    names = ["vision_encoder.patch_embeddings.conv1.weight", "vision_encoder.patch_embeddings.conv1.bias",
             "vision_encoder.patch_embeddings.residual_path.0.weight", "vision_encoder.patch_embeddings.residual_path.2.weight",
             "vision_encoder.patch_embeddings.residual_path.3.bias", "vision_encoder.patch_embeddings.residual_path.5.weight",
             "vision_encoder.patch_embeddings.residual_path.5.bias", "vision_encoder.patch_embeddings.projection.weight",
             "vision_encoder.row_position_embeddings.weight", "h.0.dec_attn.qkv_net.weight"]
    res = {}
    for cl in (True, "explicit-columns", False):
        cfg, params, gold, model, oracle, seed = build(name, compute_dtype=torch.bfloat16)
        model.use_channels_last = bool(cl)
        model.use_implicit_conv = cl is True
        tasks = make_batch(name, cfg, seed)
        logits, loss = model(to_inputs(tasks))
        model.backward()
        res[cl] = (float(loss), {n: model.G(n).detach().double().cpu().numpy().copy() for n in names})
    _, ref_loss, _ = oracle.forward([O.TaskBatch(**t) for t in tasks])
    ref_grads = oracle.backward()
    for cl in (True, "explicit-columns", False):
        assert abs(res[cl][0] - ref_loss) < 3e-2, (cl, res[cl][0], ref_loss)
        for n in names:
            assert rel_err(res[cl][1][n], ref_grads[n]) < 8e-2, (cl, n)
    for n in names:
        assert rel_err(res[True][1][n], res[False][1][n]) < 3e-2, n
        assert rel_err(res[True][1][n], res["explicit-columns"][1][n]) < 3e-2, n


@pytest.mark.parametrize("workload", ["text", "mixture"])
def test_bench_line_contract(workload):
    """bench.py prints ONE JSON line with the driver's keys, the roofline and (here skipped) cpu_baseline objects; tiny debug geometry.
    `mixture` = RL + text + caption rows through the patch embedder"""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--layers", "2", "--batch", "4",
                        "--no-cpu-baseline", "--workload", workload], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"].startswith("synthetic") and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 4 * 1024 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert workload in d["config"]["workload"] and np.isfinite(d["final_loss"])
    if workload == "text":    # the default line carries BASELINE configs 4 / 5 as legs on the same model (ga16 / decode: 24 layers only)
        for wl in ("rl", "mixture"):
            assert d[wl].get("error") is None and d[wl]["tokens_per_s"] > 0 and d[wl]["n_gpus"] == 1, d[wl]
        assert "ga16" not in d
    else:
        assert "rl" not in d and "mixture" not in d


@pytest.mark.parametrize("tag", ["plain", "deepnorm"])
def test_init_matches_reference_distribution(tag):
    """a14 (transformer_xl.py:444-468): the build's own initialisation against the moments of the REFERENCE's initialisation of the same
    configuration (tests/golden/init_stats.npz, written by make_golden.py init): constants exactly (LayerNorm / GroupNorm (1, 0), Linear
    biases 0), random tensors by mean / std / range within sampling error -- N(0, 0.02) for Linear / Embedding / u, v, the nn.Conv2d
    default U(+-1/sqrt(fan_in)) for the patch embedder's convolutions AND their biases, xavier-uniform with the DeepNorm gains
    (qkv: gain 1, its value third and o_net / FF: beta = (8 N)^-1/4)."""
    from bdm_db1_amd import TransformerXL
    from golden_util import case_cfg
    ref = np.load(os.path.join(G, "init_stats.npz"))
    cfg = case_cfg("small_mixed")
    cfg.update(dict(n_embed=int(ref["cfg_n_embed"]), n_head=int(ref["cfg_n_head"]), n_layer=int(ref["cfg_n_layer"]), text_vocab_size=int(ref["cfg_text_vocab_size"]),
                    use_deepnorm=(tag == "deepnorm")))
    torch.manual_seed(4321)
    model = TransformerXL(SimpleNamespace(**cfg), compute_dtype=torch.float32)
    sd = model.state_dict()
    if tag == "deepnorm":
        assert abs(model.deepnorm_alpha - ref["deepnorm/alpha_beta"][0]) < 1e-6 and abs(model.deepnorm_beta - ref["deepnorm/alpha_beta"][1]) < 1e-6
    names = [str(n) for n in ref[f"{tag}/names"]]
    assert len(names) > 40
    for n in names:
        rmean, rstd, rmin, rmax, numel = ref[f"{tag}/{n}"]
        v = sd[n].double().cpu()
        assert v.numel() == int(numel), n
        mean, std, mn, mx = float(v.mean()), float(v.std(unbiased=False)), float(v.min()), float(v.max())
        if rstd == 0.0:
            assert (mean, mn, mx) == (rmean, rmin, rmax), (n, mean, mn, mx)
            continue
        se = 1.0 / math.sqrt(numel)
        assert abs(mean - rmean) < 6.0 * rstd * se + 1e-9, (n, mean, rmean)
        assert abs(std / rstd - 1.0) < max(5.0 * se, 0.01) + (0.12 if numel < 200 else 0.0), (n, std, rstd)
        if numel >= 1000:   # the shape of the distribution: range / std is 2 sqrt(3) for a uniform law, ~8-10 for a normal sample of this size
            assert abs((mx - mn) / std - (rmax - rmin) / rstd) < 0.18 * (rmax - rmin) / rstd, (n, (mx - mn) / std, (rmax - rmin) / rstd)
        else:               # small uniform tensors (convolution biases): inside the reference's bound
            assert mx <= rmax * 1.15 + 1e-6 and mn >= rmin * 1.15 - 1e-6, (n, mn, mx, rmin, rmax)


def test_graphed_training_micro_steps_equal_eager_steps():
    """bdm_db1_amd.GraphedTrainStep: forward + backward of a micro-step as a hipGraph replay (two graphs: first / accumulating micro-step of
    an accumulation window, dropout step from a device counter) against the eager engine on the same batches: three optimizer steps of
    two micro-steps with dropout 0.1 -- the losses and every parameter afterwards are bit-identical (same kernels, same masks)."""
    from bdm_db1_amd import GraphedTrainStep, TransformerXL, initialize
    from bdm_db1_amd.data import NLPTaskInput
    cfg = dict(case_cfg("small_mixed"))
    # (d = 512: the register-resident LayerNorm kernels -- the generic ones of other widths add their parameter gradients with atomics)
    cfg.update(dict(n_embed=512, n_head=4, n_layer=2, n_position=256, mem_len=256, text_vocab_size=2000, drop=0.1, embd_pdrop=0.1))
    params = make_params(cfg, 17)
    rng = np.random.default_rng(5)
    batches = []
    for _ in range(6):
        ids = rng.integers(0, 2000, (4, 257))
        batches.append(NLPTaskInput(position_id=None, attention_mask=None, loss_mask=torch.ones(4, 256, device=DEV), label=torch.from_numpy(ids[:, 1:].copy()).to(DEV),
                                    text_seq=torch.from_numpy(ids[:, :-1].copy()).to(DEV), text_len=None))

    def run(graphed):
        torch.manual_seed(99)   # -> the same dropout seed on both sides
        model = TransformerXL(SimpleNamespace(**cfg), compute_dtype=torch.bfloat16)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
        eargs = SimpleNamespace(lr=2e-3, weight_decay=0.01, clip_grad=1.0, optimizer="adamw", keep_logits=False, fuse_head_loss=True, gradient_accumulation_steps=2)
        engine, _, _, _ = initialize(eargs, model)
        engine.train()
        g = GraphedTrainStep(engine, [batches[0]]) if graphed else None
        losses = []
        for b in batches:
            if g is not None:
                loss = g([b])
            else:
                _, loss = engine([b])
                engine.backward(loss)
            engine.step()
            losses.append(float(loss))
        if g is not None:
            g.close()
            assert model._drop_step == 6
        return losses, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    l0, p0 = run(False)
    l1, p1 = run(True)
    assert l0 == l1, (l0, l1)
    assert len(set(round(x, 6) for x in l0)) == len(l0)      # different batches / masks every micro-step
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_bf16_mixture_optimizer_step_is_bit_reproducible():
    """a full optimizer step -- forward + backward of an RL + text + caption batch through the patch embedder (channels-last, implicit
    convolutions), global-norm clip, AdamW, with dropout -- twice from the same state: identical loss, gradients, global norm and
    parameters, bit for bit.  Round 3 removed the last float atomics of this path: the convolution weight gradients (fixed-order
    partial sums), the GroupNorm parameter gradients, the 128-tile split-K of conv1's weight gradient, the global norm, short column sums."""
    from bdm_db1_amd import TransformerXL, initialize, synth
    cfg = synth.db1_config("1.3B", n_layer=2, n_embed=512, n_head=4, drop=0.1, embd_pdrop=0.1)
    runs = []
    for _ in range(2):
        torch.manual_seed(7)
        model = TransformerXL(cfg, compute_dtype=torch.bfloat16)
        eargs = SimpleNamespace(lr=1e-3, weight_decay=0.01, clip_grad=1.0, optimizer="adamw", keep_logits=False, fuse_head_loss=True)
        engine, _, _, _ = initialize(eargs, model)
        engine.train()
        batch = synth.mixture_batch(4, cfg.n_position, 3, DEV, cfg)
        losses = []
        for _step in range(2):
            _, loss = engine(batch)
            engine.backward(loss)
            grads = model.arena.grad.clone()
            engine.step()
            losses.append(float(loss))
        runs.append((losses, grads, float(engine._norm_sq), {k: v.detach().clone() for k, v in model.state_dict().items()}))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert runs[0][2] == runs[1][2]
    bad = [n for n in model.arena.offsets if not torch.equal(model.arena.view(runs[0][1], n), model.arena.view(runs[1][1], n))]
    assert not bad, bad
    badp = [k for k in runs[0][3] if not torch.equal(runs[0][3][k], runs[1][3][k])]
    assert not badp, badp


def test_graphed_training_with_vision_batches():
    """GraphedTrainStep on an RL + text + caption batch: the vision position ids -- drawn per step on the host by the reference's training
    mode (vision_embedding.py:150-169), which a capture cannot contain -- are static device inputs of the graph, refilled with a fresh draw
    before every replay; with ids supplied by the caller the graphed micro-step reproduces the eager one bit for bit"""
    from bdm_db1_amd import GraphedTrainStep, TransformerXL, initialize, synth
    cfg = synth.db1_config("1.3B", n_layer=2, n_embed=512, n_head=4, drop=0.1, embd_pdrop=0.1)

    def build():
        torch.manual_seed(7)
        model = TransformerXL(cfg, compute_dtype=torch.bfloat16)
        engine, _, _, _ = initialize(SimpleNamespace(lr=1e-3, weight_decay=0.01, clip_grad=1.0, optimizer="adamw", keep_logits=False, fuse_head_loss=True), model)
        engine.train()
        return model, engine
    model, engine = build()
    batch = synth.mixture_batch(4, cfg.n_position, 3, DEV, cfg)
    g = GraphedTrainStep(engine, batch)
    assert g._vis, "the mixture batch has image tasks without explicit position ids"
    st = g._vis[0][0]
    loss_a = float(g(batch)); ids_a = st.vision_row_ids.clone()
    engine.step()
    loss_b = float(g(batch)); ids_b = st.vision_row_ids.clone()
    engine.step()
    assert np.isfinite(loss_a) and np.isfinite(loss_b) and not torch.equal(ids_a, ids_b)      # a fresh draw per micro-step
    g.close()
    # caller-supplied ids: graph == eager
    torch.manual_seed(3)
    fixed = []
    for t in batch:
        geo = g._vision_geometry(t)
        if geo is not None:
            r, c = model._vision_position_ids(geo[1], geo[2], geo[0])
            t.vision_row_ids, t.vision_col_ids = r.to(DEV), c.to(DEV)
            fixed.append(t)
    assert fixed
    res = []
    for graphed in (False, True):
        model, engine = build()
        if graphed:
            gs = GraphedTrainStep(engine, batch)
            loss = gs(batch)
        else:
            _, loss = engine(batch)
            engine.backward(loss)
        res.append((float(loss), model.arena.grad.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
