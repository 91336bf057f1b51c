"""A bf16 TRAINING TRAJECTORY against the fp32 oracle (VERDICT r4 "missing" 3): the reference trains in a loop (src/train_utils/train.py:32-83,
216-232: GA micro-steps of ``engine(x) -> backward -> step``); the other parity tests pin ONE forward / backward in bf16 and three Adam steps
in fp32 on the tiny model.  Here: four decoder layers at the DB1-1.3B geometry (d 2048, 16 heads of 128, GEGLU 8192 -> 4096, L 1024, the tied
33 025-row head), 2 x 1024 tokens per micro-step, dropout 0.1 on embeddings / attention / feed-forward outputs under the shared counter-based
Philox masks, AdamW (lr 5e-4, wd 0.01) with the global-norm clip at 1.0, TWENTY optimizer steps, on learnable data (Zipf-distributed ids
over 512 tokens: the loss falls from ln 33 025 = 10.4 to about 6):

  * the fp32 HIP engine follows the NumPy oracle (oracle/db1_oracle.py: forward, hand-written backward, clip, AdamW, the same keep
    decisions) step by step -- by default for the first 6 optimizer steps (the oracle needs 12-22 s of host time per step at this size);
    ``DB1_TRAJ_ORACLE_STEPS=20`` runs all twenty (profiles/r05_trajectory.json holds that run's record);
  * the bf16 HIP engine -- eager, as hipGraph replays, and with gradient accumulation 4 + deferred weight gradients (``defer_wgrad``,
    eager and graphed) -- follows the fp32 HIP engine's loss curve and final parameters over all twenty steps within a STATED tolerance:
    |loss_bf16 - loss_fp32| <= 2e-2 at every step; per weight matrix ||p_bf16 - p_fp32|| <= 2e-2 ||p_fp32||; and per tensor (LayerNorm
    parameters and biases included, which start at exactly 1 / 0) the UPDATE agrees: ||p_bf16 - p_fp32|| <= 0.15 ||p_fp32 - p_init|| (Adam's
    steps are sign-like, so elements whose gradient is ~0 walk differently under bf16 noise: measured 0.09 at worst, 0.03 in the median);
  * the tolerance discriminates: the same bf16 run with a deliberately wrong clip threshold (0.25 instead of 1.0; the gradient norm is
    10-40 here, so the clip is active at every step and its threshold changes how the steps are weighted inside Adam's moments) breaks
    it by a wide margin (loss 0.12, matrices 3.5e-2, updates 0.41).
"""
import json
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
N_LAYER, L, NSEQ, STEPS, GA = 4, 1024, 2, 20, 4
LR, WD, CLIP = 5e-4, 0.01, 1.0
ORACLE_STEPS = int(os.environ.get("DB1_TRAJ_ORACLE_STEPS", "6"))
LOSS_TOL, PARAM_TOL, UPDATE_TOL = 2e-2, 2e-2, 0.15         # bf16 vs fp32: the statement of this test (measured: 0.8e-2 / 1.2e-2 / 0.09; GA 4: 1.7e-2 / 0.6e-2 / 0.05)


def _cfg():
    from bdm_db1_amd import synth
    return synth.db1_config("1.3B", n_layer=N_LAYER, drop=0.1, embd_pdrop=0.1)


def _params(cfg):
    from golden_util import param_shapes
    cd = {k: getattr(cfg, k) for k in ("n_embed", "n_head", "n_inner", "n_layer", "text_vocab_size", "num_continuous_bin", "num_discrete_values",
                                       "overlap_with_text", "vision_patch_size", "vision_num_input_channels", "vision_position_vocab_size", "untie_r", "activation_fn",
                                       "share_input_output_embedding")}
    rng = np.random.default_rng(77)
    out = {}
    for name, shape in param_shapes(cd):
        if name.endswith("layer_norm.weight") or (".residual_path." in name and name.endswith(".weight") and len(shape) == 1):
            a = np.ones(shape)
        elif name.endswith(".bias"):
            a = np.zeros(shape)
        else:
            a = 0.02 * rng.standard_normal(shape)          # the reference's init (transformer_xl.py:456-468)
        out[name] = a.astype(np.float32)
    return out


def _ids(n_micro):
    """Zipf-distributed ids over 512 tokens scattered through the vocabulary: learnable (the unigram entropy is ~ 5.3 nats)"""
    rng = np.random.default_rng(5)
    vocab = rng.choice(32000, 512, replace=False)
    p = 1.0 / (np.arange(512) + 1.0)
    p /= p.sum()
    return vocab[rng.choice(512, size=(n_micro, NSEQ, L + 1), p=p)]


def _batch(ids):
    from bdm_db1_amd.data import NLPTaskInput
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(device=DEV, dtype=dt)
    return NLPTaskInput(position_id=None, attention_mask=None, loss_mask=t(np.ones((NSEQ, L), np.float32)), label=t(ids[:, 1:]), text_seq=t(ids[:, :-1]),
                        text_len=None)


def _run(cfg, params, ids, dtype, ga=1, defer=False, graphed=False, clip=CLIP, oracle_steps=0, record=None, defer_backward=False):
    """STEPS optimizer steps of ``ga`` micro-steps each -> (loss per optimizer step, final parameters); optionally the oracle beside it"""
    from bdm_db1_amd import GraphedTrainStep, TransformerXL, initialize
    torch.manual_seed(4242)                     # (the dropout seed of every variant)
    model = TransformerXL(cfg, compute_dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    eargs = SimpleNamespace(lr=LR, weight_decay=WD, clip_grad=clip, optimizer="adamw", adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8, keep_logits=False,
                            fuse_head_loss=True, gradient_accumulation_steps=ga, defer_wgrad=defer, defer_backward=defer_backward)
    engine, _, _, _ = initialize(eargs, model)
    engine.train()
    batches = [_batch(ids[k]) for k in range(STEPS * ga)]
    g = GraphedTrainStep(engine, [batches[0]]) if graphed else None
    oracle = None
    if oracle_steps:
        ocfg = {k: getattr(cfg, k) for k in O.OracleConfig.__dataclass_fields__ if hasattr(cfg, k)}
        oracle = O.OracleModel(O.OracleConfig(**ocfg), {k: v.copy() for k, v in params.items()}, dtype=np.float32)
        om = {k: np.zeros_like(v, dtype=np.float64) for k, v in oracle.p.items()}
        ov = {k: np.zeros_like(v, dtype=np.float64) for k, v in oracle.p.items()}
        seed = int(model.dropout_seed)
    losses = []
    for step in range(STEPS):
        acc = 0.0
        for micro in range(ga):
            b = batches[step * ga + micro]
            if g is not None:
                loss = g([b])
            else:
                _, loss = engine([b])
                engine.backward(loss)
            engine.step()
            acc += float(loss)
        losses.append(acc / ga)
        if oracle is not None and step < oracle_steps:
            assert ga == 1
            i = ids[step]
            tb = O.TaskBatch(kind="nlp", text_seq=i[:, :-1], label=i[:, 1:], loss_mask=np.ones((NSEQ, L), np.float32))
            _, ref_loss, _ = oracle.forward([tb], dropout={"seed": seed, "step": step + 1})
            grads = oracle.backward()
            norm = np.sqrt(sum((gr.astype(np.float64) ** 2).sum() for gr in grads.values()))
            coef = O.clip_coef(norm, clip)
            for k in oracle.p:
                gr = grads.get(k, np.zeros_like(oracle.p[k]))
                oracle.p[k], om[k], ov[k] = O.adam_step(oracle.p[k], gr, om[k], ov[k], step + 1, LR, wd=WD, adamw=True, grad_scale=coef)
            sd = model.state_dict()
            worst = max(float(np.linalg.norm(sd[k].double().cpu().numpy() - oracle.p[k]) / max(np.linalg.norm(oracle.p[k]), 1e-30)) for k in oracle.p)
            if record is not None:
                record.setdefault("oracle", []).append({"step": step + 1, "loss_hip_fp32": losses[-1], "loss_oracle": float(ref_loss), "grad_norm_oracle": float(norm),
                                                        "worst_param_rel_l2": worst})
            assert abs(losses[-1] - ref_loss) < 2e-4 * max(1.0, abs(ref_loss)), (step, losses[-1], ref_loss)
            assert worst < 1e-3, (step, worst)
    if g is not None:
        g.close()
    final = {k: v.detach().float().cpu().numpy().copy() for k, v in model.state_dict().items() if k in params}
    del engine, model
    torch.cuda.empty_cache()
    return losses, final


def _compare(tag, losses, final, ref_losses, ref_final, record, init):
    """-> (max |loss difference|, worst relative L2 of a WEIGHT MATRIX against the fp32 run's, worst relative L2 of any tensor's UPDATE
    (p - p_init) against the fp32 run's update)"""
    dl = max(abs(a - b) for a, b in zip(losses, ref_losses))
    n = lambda a: float(np.linalg.norm(a.astype(np.float64)))
    rel_p = {k: n(final[k] - ref_final[k]) / max(n(ref_final[k]), 1e-30) for k in ref_final if ref_final[k].ndim >= 2}
    rel_u = {k: n(final[k] - ref_final[k]) / max(n(ref_final[k] - init[k]), 1e-30) for k in ref_final}
    wp, wu = max(rel_p, key=rel_p.get), max(rel_u, key=rel_u.get)
    record[tag] = {"max_abs_loss_diff": dl, "worst_matrix_rel_l2": rel_p[wp], "worst_matrix": wp, "worst_update_rel_l2": rel_u[wu], "worst_update": wu,
                   "median_update_rel_l2": float(np.median(list(rel_u.values()))), "losses": [round(x, 5) for x in losses]}
    return dl, rel_p[wp], rel_u[wu]


@pytest.fixture(scope="module")
def traj():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg = _cfg()
    params = _params(cfg)
    ids = _ids(STEPS * GA)
    record = {"config": {"n_layer": N_LAYER, "tokens_per_micro_step": NSEQ * L, "steps": STEPS, "lr": LR, "wd": WD, "clip": CLIP, "dropout": 0.1}}
    l32, p32 = _run(cfg, params, ids, torch.float32, oracle_steps=ORACLE_STEPS, record=record)
    record["fp32_losses"] = [round(x, 5) for x in l32]
    return SimpleNamespace(cfg=cfg, params=params, ids=ids, l32=l32, p32=p32, record=record)


def _dump(record):
    out = os.environ.get("DB1_TRAJ_RECORD")
    if out:
        with open(out, "w") as f:
            json.dump(record, f, indent=1)


def test_fp32_engine_follows_the_oracle_and_learns(traj):
    assert len(traj.record.get("oracle", [])) == min(ORACLE_STEPS, STEPS)
    assert traj.l32[0] > 10.0 and traj.l32[-1] < 7.5, traj.l32          # ln 33 025 at the start; the Zipf data is being learned
    _dump(traj.record)


@pytest.mark.parametrize("graphed", [False, True])
def test_bf16_trajectory_follows_fp32(traj, graphed):
    tag = "bf16_graphed" if graphed else "bf16_eager"
    losses, final = _run(traj.cfg, traj.params, traj.ids, torch.bfloat16, graphed=graphed)
    dl, dp, du = _compare(tag, losses, final, traj.l32, traj.p32, traj.record, traj.params)
    _dump(traj.record)
    assert dl <= LOSS_TOL, (tag, dl, losses, traj.l32)
    assert dp <= PARAM_TOL and du <= UPDATE_TOL, (tag, traj.record[tag])


@pytest.mark.parametrize("graphed", [False, True])
@pytest.mark.parametrize("mode", ["wgrad", "backward"])
def test_bf16_accumulation_with_deferred_weight_gradients_follows_fp32(traj, mode, graphed):
    """GA 4 micro-steps of 2 x 1024 tokens per optimizer step against the fp32 engine on the same 80 micro-batches (per-micro-step backward
    and products): "wgrad" = weight gradients formed once per step from the stashed operands (defer_wgrad), "backward" = ONE backward per
    optimizer step over the whole accumulation window (defer_backward, round 6)"""
    if getattr(traj, "ga32", None) is None:
        traj.ga32 = _run(traj.cfg, traj.params, traj.ids, torch.float32, ga=GA)
    l32, p32 = traj.ga32
    tag = ("bf16_ga4_defer_" if mode == "wgrad" else "bf16_ga4_window_") + ("graphed" if graphed else "eager")
    losses, final = _run(traj.cfg, traj.params, traj.ids, torch.bfloat16, ga=GA, defer=mode == "wgrad", graphed=graphed, defer_backward=mode == "backward")
    dl, dp, du = _compare(tag, losses, final, l32, p32, traj.record, traj.params)
    traj.record[tag]["fp32_ga4_losses"] = [round(x, 5) for x in l32]
    _dump(traj.record)
    assert dl <= LOSS_TOL, (tag, dl, losses, l32)
    assert dp <= PARAM_TOL and du <= UPDATE_TOL, (tag, traj.record[tag])


def test_a_wrong_clip_threshold_breaks_the_tolerance(traj):
    losses, final = _run(traj.cfg, traj.params, traj.ids, torch.bfloat16, clip=0.25)
    dl, dp, du = _compare("bf16_wrong_clip", losses, final, traj.l32, traj.p32, traj.record, traj.params)
    _dump(traj.record)
    assert dl > 2 * LOSS_TOL or dp > 2 * PARAM_TOL or du > 2 * UPDATE_TOL, traj.record["bf16_wrong_clip"]
