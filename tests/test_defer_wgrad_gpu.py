"""Gradient accumulation with the weight gradients formed ONCE per optimizer step (engine option ``defer_wgrad``, model.WgradStash): the
operands (x, dy) of the four big linear maps of every layer are kept for all micro-steps of an accumulation window and dW = dy^T x runs over
K = ga * T rows on the boundary micro-step.  Same gradients as the per-micro-step products up to fp32 summation order; the hipGraph-captured
form (one graph per micro-step of the window) is bit-identical to the eager one.  Reference cadence: src/train_utils/train.py:216-232."""
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

from golden_util import case_cfg, make_params  # noqa: E402

DEV = "cuda"
GA = 3


def _setup(drop, ragged=False):
    cfg = dict(case_cfg("small_mixed"))
    cfg.update(dict(n_embed=512, n_head=4, n_layer=3, n_position=256, mem_len=256, text_vocab_size=2000, drop=drop, embd_pdrop=drop))
    params = make_params(cfg, 23)
    rng = np.random.default_rng(8)
    from bdm_db1_amd.data import NLPTaskInput
    batches = []
    for k in range(2 * GA):
        nseq = 2 if (ragged and k % GA == 1) else 4      # ragged: the middle micro-step of every window is a short batch
        ids = rng.integers(0, 2000, (nseq, 257))
        batches.append(NLPTaskInput(position_id=None, attention_mask=None, loss_mask=torch.ones(nseq, 256, device=DEV), label=torch.from_numpy(ids[:, 1:].copy()).to(DEV),
                                    text_seq=torch.from_numpy(ids[:, :-1].copy()).to(DEV), text_len=None))
    return cfg, params, batches


def _run(cfg, params, batches, defer, graphed=False, dtype=torch.bfloat16):
    from bdm_db1_amd import GraphedTrainStep, TransformerXL, initialize
    torch.manual_seed(99)
    model = TransformerXL(SimpleNamespace(**cfg), compute_dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    eargs = SimpleNamespace(lr=2e-3, weight_decay=0.01, clip_grad=1.0, optimizer="adamw", keep_logits=False, fuse_head_loss=True,
                            gradient_accumulation_steps=GA, defer_wgrad=defer)
    engine, _, _, _ = initialize(eargs, model)
    engine.train()
    assert engine.defer_wgrad == defer
    g = GraphedTrainStep(engine, [batches[0]]) if graphed else None
    losses, grads_at_boundary = [], None
    for k, b in enumerate(batches):
        if g is not None:
            loss = g([b])
        else:
            _, loss = engine([b])
            engine.backward(loss)
        if k == GA - 1:
            grads_at_boundary = model.arena.grad.detach().clone()     # the whole window's gradients, before the optimizer consumes them
        engine.step()
        losses.append(float(loss))
    if g is not None:
        g.close()
    if defer:
        assert model.wgrad_stash is not None and model.wgrad_stash.ga == GA and model.wgrad_stash.T == 4 * 256
    return losses, grads_at_boundary, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}, dict(model.arena.offsets)


@pytest.mark.parametrize("drop", [0.1, 0.0])
def test_deferred_weight_gradients_equal_per_micro_step_products(drop):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg, params, batches = _setup(drop)
    l0, g0, p0, offs = _run(cfg, params, batches, defer=False)
    l1, g1, p1, _ = _run(cfg, params, batches, defer=True)
    assert l0[:GA] == l1[:GA]                 # the forward passes of the first window are the same launches on the same weights
    for name, (off, shape, alloc) in offs.items():
        a, b = g1[off:off + alloc].double(), g0[off:off + alloc].double()
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, name
            continue
        err = float((a - b).abs().max() / b.abs().max())
        # bf16 operands are identical on both sides; only the fp32 summation order differs (one K = ga * T product vs ga accumulated ones)
        assert err < 2e-5, (name, err)
    # (the parameters after the optimizer steps are NOT compared element-wise: Adam's first updates are lr * sign-like, so an element whose
    #  gradient is ~0 takes its step in the other direction on a last-bit difference.  The second window is covered by the losses -- and by
    #  the bit-level graphed-vs-eager test below, which runs two windows through the same stash.)
    assert all(abs(a - b) < 2e-2 for a, b in zip(l0[GA:], l1[GA:])), (l0, l1)


def test_deferred_weight_gradients_graphed_equal_eager():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg, params, batches = _setup(0.1)
    l0, g0, p0, _ = _run(cfg, params, batches, defer=True)
    l1, g1, p1, _ = _run(cfg, params, batches, defer=True, graphed=True)
    assert l0 == l1, (l0, l1)
    assert torch.equal(g0, g1)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_deferred_weight_gradients_fp32_path():
    """the fp32 parity path stashes too (generic GEMM kernels): same gradients as without"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg, params, batches = _setup(0.0)
    _, g0, _, offs = _run(cfg, params, batches, defer=False, dtype=torch.float32)
    _, g1, _, _ = _run(cfg, params, batches, defer=True, dtype=torch.float32)
    for name, (off, shape, alloc) in offs.items():
        a, b = g1[off:off + alloc].double(), g0[off:off + alloc].double()
        if float(b.abs().max()) > 0.0:
            assert float((a - b).abs().max() / b.abs().max()) < 2e-5, name


def test_token_count_changes_inside_an_accumulation_window():
    """a micro-step with another token count (a short batch) inside an accumulation window makes the model rebuild its stash;
    the operands of the micro-steps done so far must not be lost -- they are flushed into the gradient arena before the old stash goes, and
    the new stash starts at the current micro-step and accumulates on top.  Window = (4, 2, 4) sequences: two rebuilds per window."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg, params, batches = _setup(0.0, ragged=True)
    l0, g0, p0, offs = _run(cfg, params, batches, defer=False)
    l1, g1, p1, _ = _run(cfg, params, batches, defer=True)
    assert l0[:GA] == l1[:GA]
    assert torch.isfinite(g1).all()
    for name, (off, shape, alloc) in offs.items():
        a, b = g1[off:off + alloc].double(), g0[off:off + alloc].double()
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, name
            continue
        assert float((a - b).abs().max() / b.abs().max()) < 2e-5, name
    assert all(abs(a - b) < 2e-2 for a, b in zip(l0[GA:], l1[GA:])), (l0, l1)


def test_partial_sum_stash_at_the_reference_micro_batch_geometry():
    """4 sequences of 1024 tokens per micro-step at the DB1-1.3B layer geometry (2 layers), GA 2: here every producer of a small reduction has
    its partials-only form (db1_layernorm_residual_bwd_parts, db1_relattn_dqr_fused_parts, db1_gemm_nn_geglu_bwd_parts), so the LayerNorm
    parameter gradients, the u / v column sums and the first feed-forward bias's column sums of BOTH micro-steps are added up once per layer at
    the flush (WgradStash.alloc_parts), r_net's weight gradient and the second bias come out of the stash too, and r_net runs batched over
    the layers.  Every gradient equals the per-micro-step engine's up to summation order; the graphed run equals the eager one bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bdm_db1_amd import GraphedTrainStep, TransformerXL, initialize, synth
    cfg = synth.db1_config("1.3B", n_layer=2, drop=0.1, embd_pdrop=0.1)
    ga, B, L = 2, 4, 1024
    batches = [synth.text_batch(B, L, 40 + k, DEV) for k in range(ga)]

    def run(defer, graphed=False):
        torch.manual_seed(7)
        model = TransformerXL(cfg)
        eargs = SimpleNamespace(lr=1e-4, weight_decay=0.01, clip_grad=1.0, optimizer="adamw", keep_logits=False, fuse_head_loss=True,
                                gradient_accumulation_steps=ga, defer_wgrad=defer)
        engine, _, _, _ = initialize(eargs, model)
        engine.train()
        g = GraphedTrainStep(engine, [batches[0]]) if graphed else None
        losses = []
        for k, b in enumerate(batches):
            if g is not None:
                loss = g([b])
            else:
                _, loss = engine([b])
                engine.backward(loss)
            if k == ga - 1:
                grads = model.arena.grad.detach().clone()
            engine.step()
            losses.append(float(loss))
        if defer:
            st = model.wgrad_stash
            assert st is not None and st.parts and st.r_used, "this geometry is meant to take the partial-sum stash"
        offs = dict(model.arena.offsets)
        if g is not None:
            g.close()
        del engine, model
        torch.cuda.empty_cache()
        return losses, grads, offs
    l0, g0, offs = run(False)
    l1, g1, _ = run(True)
    assert l0 == l1
    for name, (off, shape, alloc) in offs.items():
        a, b = g1[off:off + alloc].double(), g0[off:off + alloc].double()
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, name
            continue
        err = float((a - b).abs().max() / b.abs().max())
        assert err < 5e-5, (name, err)
    l2, g2, _ = run(True, graphed=True)
    assert l2 == l1 and torch.equal(g2, g1)
