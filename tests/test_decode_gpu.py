"""Inference with Transformer-XL memory: the fused decode attention kernel (1..64 queries against mlen + q cached keys) against a
NumPy statement of the closed form, and the model's K/V-cached memory path against the reference's golden logits."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def bf(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


def dev16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(torch.bfloat16)


@pytest.mark.parametrize("B,q,mlen,H,shift", [(1, 1, 1024, 16, 1024), (2, 5, 100, 2, 105), (1, 22, 1024, 3, 1024), (1, 40, 300, 2, 64),
                                               (2, 64, 64, 1, 128), (1, 1, 0, 2, 1), (1, 17, 7, 2, 5)])
def test_decode_attention_matches_closed_form(B, q, mlen, H, shift):
    """score[i,j] = ((q_i+u).k_j + (q_i+v).R[mlen+i-j]) / sqrt(d), visible iff i - shift < j <= i + mlen (transformer_xl.py:160-209)"""
    from bdm_db1_amd import ops
    D, klen = 128, mlen + q
    rng = np.random.default_rng(q * 1000 + mlen)
    qu, qv = bf(rng.standard_normal((B, q, H, D))), bf(rng.standard_normal((B, q, H, D)))
    kv = bf(rng.standard_normal((B, klen, 2, H, D)))
    R = bf(rng.standard_normal((klen, H, D)))
    scale = 1.0 / math.sqrt(D)
    i = np.arange(q)[:, None]; j = np.arange(klen)[None, :]
    AC = np.einsum("bihd,bjhd->bhij", qu, kv[:, :, 0])
    T = np.einsum("bihd,rhd->bhir", qv, R)
    BD = np.take_along_axis(T, np.broadcast_to(np.clip(mlen + i - j, 0, klen - 1)[None, None], AC.shape), axis=3)
    vis = (j <= i + mlen) & (j > i - shift)
    S = np.where(vis[None, None], (AC + BD) * scale, -np.inf)
    Pm = np.exp(S - S.max(-1, keepdims=True))
    Pm /= Pm.sum(-1, keepdims=True)
    ref = np.einsum("bhij,bjhd->bihd", Pm, kv[:, :, 1])
    KV = dev16(kv)
    out = torch.full((B, q, H, D), 7.0, device=DEV, dtype=torch.bfloat16)
    assert ops.relattn_decode_supported(B, q, klen, H, D, torch.bfloat16)
    ops.relattn_decode_fwd(dev16(qu), dev16(qv), KV[:, :, 0], KV[:, :, 1], dev16(R).view(klen, H * D), out, B, q, klen, mlen, H, D, shift, scale)
    torch.cuda.synchronize()
    err = np.abs(out.double().cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 2e-2, f"decode attention rel err {err:.3e}"  # bf16 rounding of P and of the output


@pytest.mark.parametrize("pre_lnorm", [False, True])
def test_model_kv_cached_memory_inference_matches_rereojection(pre_lnorm):
    """the K/V-cached path (new tokens projected once, fused decode attention) against the reference-shaped path that
    re-projects cat([mem, w]) every call (transformer_xl.py:124-133), over a call sequence like evaluate_rl's: a multi-token
    observation, then single tokens; memory first zero (init_mem), then filling, then sliding.  Also: a caller that hands in
    copies of the memory (cache miss -> rebuilt from the hidden states) gets the same logits."""
    from bdm_db1_amd import TransformerXL, synth
    from bdm_db1_amd.data import NLPTaskInput
    cfg = synth.db1_config("tiny", n_embed=256, n_head=2, n_layer=2, n_position=64, mem_len=40, pre_lnorm=pre_lnorm, fp16=True)
    torch.manual_seed(3)
    model = TransformerXL(cfg, device=torch.device(DEV), compute_dtype=torch.bfloat16)
    model.eval()
    assert model.d_head == 128
    rng = np.random.default_rng(0)
    calls = [rng.integers(0, 32000, (2, q)) for q in (7, 1, 1, 22, 1, 30, 1, 1)]

    def run(use_decode, copy_mems):
        model.use_decode = use_decode
        model._dec_state = None
        mems = model.init_mem(2)
        outs = []
        with torch.no_grad():
            for ids in calls:
                x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None,
                                 text_seq=torch.from_numpy(ids).to(DEV), text_len=None)
                logits, _, mems = model([x], compute_loss=False, mems=mems)
                outs.append(logits.float().cpu().numpy())
                if copy_mems:
                    mems = [m.clone() for m in mems]
        return outs

    ref = run(False, False)
    for copy_mems in (False, True):
        got = run(True, copy_mems)
        for step, (a, b) in enumerate(zip(got, ref)):
            err = np.abs(a - b).max() / np.abs(b).max()
            assert err < 3e-2, f"step {step} (copy_mems={copy_mems}): rel err {err:.3e}"  # two bf16 pipelines, different rounding points
    assert model._dec_state is not None  # the fused path really ran


def test_graphed_memory_step_matches_eager_calls():
    """one hipGraph replay per call (static buffers) reproduces the eager K/V-cached calls on the same token stream"""
    from bdm_db1_amd import TransformerXL, GraphedMemoryStep, synth
    from bdm_db1_amd.data import NLPTaskInput
    cfg = synth.db1_config("tiny", n_embed=256, n_head=2, n_layer=2, n_position=64, mem_len=40, fp16=True)
    torch.manual_seed(5)
    model = TransformerXL(cfg, device=torch.device(DEV), compute_dtype=torch.bfloat16)
    model.eval()
    rng = np.random.default_rng(1)
    stream = [torch.from_numpy(rng.integers(0, 32000, (1, 3))).to(DEV) for _ in range(12)]
    mems, eager = model.init_mem(1), []
    with torch.no_grad():
        for ids in stream:
            x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
            logits, _, mems = model([x], compute_loss=False, mems=mems)
            eager.append(logits.float().cpu().numpy())
    step = GraphedMemoryStep(model, batch_size=1, n_new=3)
    for rep in range(2):  # second pass after reset_memory(): a new episode starts from the zero memory again
        for t, ids in enumerate(stream):
            logits, _ = step(ids)
            got = logits.float().cpu().numpy()
            err = np.abs(got - eager[t]).max() / np.abs(eager[t]).max()
            assert err < 1e-6, f"pass {rep} call {t}: graph vs eager rel err {err:.3e}"
        step.reset_memory()


@pytest.mark.parametrize("pre_lnorm", [False, True])
def test_ring_memory_matches_the_kv_cached_path(pre_lnorm):
    """RingMemory (keys / values of the memory in a ring, appended in place by db1_relattn_decode_ring_fwd, origin on the device) against
    the K/V-cached list-memory path over an evaluate_rl-like call sequence with 2 sequences: multi-token observations (the tile-per-wave
    form, q > 16), single tokens (the step-per-wave form), the memory sliding past mem_len and the ring wrapping (mem_len 40 + 64 rows)."""
    from bdm_db1_amd import RingMemory, TransformerXL, synth
    from bdm_db1_amd.data import NLPTaskInput
    cfg = synth.db1_config("tiny", n_embed=256, n_head=2, n_layer=2, n_position=64, mem_len=40, pre_lnorm=pre_lnorm, fp16=True)
    torch.manual_seed(3)
    model = TransformerXL(cfg, device=torch.device(DEV), compute_dtype=torch.bfloat16)
    model.eval()
    rng = np.random.default_rng(0)
    calls = [rng.integers(0, 32000, (2, q)) for q in (7, 1, 1, 22, 1, 30, 1, 1, 16, 17, 1, 40, 1, 1, 1, 9)]

    def run(ring):
        mems = RingMemory(model, 2) if ring else model.init_mem(2)
        model._dec_state = None
        outs = []
        with torch.no_grad():
            for ids in calls:
                x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=torch.from_numpy(ids).to(DEV), text_len=None)
                logits, _, mems = model([x], compute_loss=False, mems=mems)
                outs.append(logits.float().cpu().numpy())
        return outs
    ref, got = run(False), run(True)
    for step, (a, b) in enumerate(zip(got, ref)):
        err = np.abs(a - b).max() / np.abs(b).max()
        assert err < 1e-2, f"call {step} (q = {calls[step].shape[1]}): rel err {err:.3e}"   # same kernels' maths, different partial-sum grouping


def test_graphed_ring_steps_share_a_memory():
    """GraphedRingStep: an observation call (q = 22) and 1-token calls as two captured graphs over ONE RingMemory reproduce the eager
    K/V-cached calls of the same token stream; reset_memory() starts a new episode"""
    from bdm_db1_amd import GraphedRingStep, TransformerXL, synth
    from bdm_db1_amd.data import NLPTaskInput
    cfg = synth.db1_config("tiny", n_embed=256, n_head=2, n_layer=2, n_position=64, mem_len=40, fp16=True)
    torch.manual_seed(5)
    model = TransformerXL(cfg, device=torch.device(DEV), compute_dtype=torch.bfloat16)
    model.eval()
    rng = np.random.default_rng(1)
    stream = []
    for _ in range(5):   # five transitions: 22 observation tokens, then three single tokens
        stream += [torch.from_numpy(rng.integers(0, 32000, (1, 22))).to(DEV)] + [torch.from_numpy(rng.integers(0, 32000, (1, 1))).to(DEV) for _ in range(3)]
    mems, eager = model.init_mem(1), []
    with torch.no_grad():
        for ids in stream:
            x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
            logits, _, mems = model([x], compute_loss=False, mems=mems)
            eager.append(logits.float().cpu().numpy())
    obs = GraphedRingStep(model, batch_size=1, n_new=22)
    one = GraphedRingStep(model, batch_size=1, n_new=1, memory=obs.memory)
    for rep in range(2):
        for t, ids in enumerate(stream):
            logits, _ = (obs if ids.shape[1] == 22 else one)(ids)
            got = logits.float().cpu().numpy()
            err = np.abs(got - eager[t]).max() / np.abs(eager[t]).max()
            assert err < 1e-2, f"pass {rep} call {t}: rel err {err:.3e}"
        obs.reset_memory()


@pytest.mark.parametrize("M", [1, 5, 16, 22, 64])
def test_linear_decode_equals_the_unfused_launches(M):
    """db1_linear_decode at the DB1-1.3B layer shapes: the attention output projection + residual LayerNorm, the first feed-forward map through
    GEGLU, the second one (split over K) + residual LayerNorm -- against db1_gemm / db1_ffn_act_fwd / db1_layernorm_residual_fwd.
    The unsplit GEGLU launch is bit-equal; the ones split over K add their partial tiles in a fixed order (equal from run to run) and differ from
    the single accumulation chain by fp32 rounding only; the LayerNorm tail equals the LayerNorm launch on the same y.  The ticket counters are zero again after every launch."""
    from bdm_db1_amd import ops
    d, dff, eps, alpha = 2048, 4096, 1e-5, 0.81
    g = torch.Generator(device=DEV).manual_seed(M)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=g) * sc).to(torch.bfloat16)
    x, res = rnd(M, d), rnd(M, d)
    Wo, W1, b1, W2, b2 = rnd(d, d, sc=0.03), rnd(2 * dff, d, sc=0.03), rnd(2 * dff, sc=0.5), rnd(d, dff, sc=0.03), rnd(d, sc=0.5)
    gam, bet = rnd(d) + 1, rnd(d, sc=0.1)
    new = lambda *s: torch.empty(*s, device=DEV, dtype=torch.bfloat16)
    f32 = lambda *s: torch.empty(*s, device=DEV, dtype=torch.float32)

    def unfused_ln(xin, W, bias, r):
        y, out = new(M, W.shape[0]), new(M, W.shape[0])
        ops.gemm(xin, W.t(), y, bias=bias)
        ops.layernorm_residual_fwd(r, y, alpha, gam, bet, out, None, f32(M), f32(M), eps)
        return y, out
    def close(got, ref, what):
        err = (got.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
        assert err < 1e-2, f"{what}: rel err {err:.3e}"
    # 1. o_net + LayerNorm (K = 2048 split over two workgroups per column group)
    y_ref, h_ref = unfused_ln(x, Wo, None, res)
    y, h = new(M, d), new(M, d)
    ops.linear_decode(x, Wo, None, y, ln=(res, alpha, gam, bet, eps, h))
    close(y, y_ref, "o_net"), close(h, h_ref, "o_net LayerNorm")
    h_of_y = new(M, d)
    ops.layernorm_residual_fwd(res, y, alpha, gam, bet, h_of_y, None, f32(M), f32(M), eps)
    assert torch.equal(h, h_of_y)     # the LayerNorm tail is the LayerNorm launch's arithmetic on the y this launch stored
    # 2. CoreNet.0 through GEGLU
    z, a_ref = new(M, 2 * dff), new(M, dff)
    ops.gemm(x, W1.t(), z, bias=b1)
    ops.ffn_act_fwd(z, a_ref, "geglu")
    act = new(M, dff)
    ops.linear_decode(x, W1, b1, act, geglu=True)
    assert torch.equal(act, a_ref)
    # 3. CoreNet.2 (K = 4096: two workgroups per column group) + LayerNorm
    f_ref, o_ref = unfused_ln(a_ref, W2, b2, res)
    f, o = new(M, d), new(M, d)
    ops.linear_decode(act, W2, b2, f, ln=(res, alpha, gam, bet, eps, o))
    f2, o2 = new(M, d), new(M, d)
    ops.linear_decode(act, W2, b2, f2, ln=(res, alpha, gam, bet, eps, o2))
    assert torch.equal(f, f2) and torch.equal(o, o2)
    close(f, f_ref, "CoreNet.2"), close(o, o_ref, "CoreNet.2 LayerNorm")
    assert int(ops.decode_tickets(x.device).abs().sum().item()) == 0
    if M <= 16:
        # 4. the LayerNorm on the way IN: qkv = LN(alpha * res + x) W^T, and GEGLU(LN(.) W_1^T + b_1); the normalised rows are stored as well
        hn_ref = new(M, d)
        ops.layernorm_residual_fwd(res, x, alpha, gam, bet, hn_ref, None, f32(M), f32(M), eps)
        Wq = rnd(3 * d, d, sc=0.03)
        q_ref, q, hn = new(M, 3 * d), new(M, 3 * d), new(M, d)
        ops.gemm(hn_ref, Wq.t(), q_ref)
        ops.linear_decode(x, Wq, None, q, pre=(res, alpha, gam, bet, eps, hn))
        close(hn, hn_ref, "input LayerNorm rows"), close(q, q_ref, "qkv after the input LayerNorm")
        assert (hn.float() - hn_ref.float()).abs().max().item() <= 2.0 ** -6 * hn_ref.float().abs().max().item()   # a bf16 last place at most
        q2 = new(M, 3 * d)
        ops.gemm(hn, Wq.t(), q2)
        assert torch.equal(q, q2)           # the projection used exactly the rows it stored
        ops.ffn_act_fwd(z.copy_(torch.addmm(b1.float(), hn.float(), W1.float().t()).to(torch.bfloat16)), a_ref, "geglu")
        hn2 = new(M, d)
        ops.linear_decode(x, W1, b1, act, geglu=True, pre=(res, alpha, gam, bet, eps, hn2))
        assert torch.equal(hn2, hn)
        close(act, a_ref, "GEGLU after the input LayerNorm")


def test_fused_inference_layer_equals_the_separate_launches():
    """the model's inference path with the fused linear maps (use_decode_fused) against the same path with separate activation / LayerNorm
    launches, over a ring memory: 1-token and multi-token calls, d_model = 512 (LayerNorm rows of one wave), with the residual LayerNorms on
    the way in to the next linear map (<= 16 tokens) and on the way out, eager and as a graph"""
    from bdm_db1_amd import GraphedRingStep, RingMemory, TransformerXL, synth
    from bdm_db1_amd.data import NLPTaskInput
    cfg = synth.db1_config("tiny", n_embed=512, n_head=4, n_layer=2, n_position=64, mem_len=40, fp16=True)
    torch.manual_seed(11)
    model = TransformerXL(cfg, device=torch.device(DEV), compute_dtype=torch.bfloat16)
    model.eval()
    assert model.d_head == 128 and model._decode_fused_ok(1, False, None)
    rng = np.random.default_rng(4)
    calls = [torch.from_numpy(rng.integers(0, 32000, (2, q))).to(DEV) for q in (7, 1, 1, 22, 1, 30, 1, 1, 16, 1)]

    def run(fused, prologue=True):
        model.use_decode_fused, model.use_decode_ln_prologue = fused, prologue
        mems, outs = RingMemory(model, 2), []
        with torch.no_grad():
            for ids in calls:
                x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
                logits, _, mems = model([x], compute_loss=False, mems=mems)
                outs.append(logits.float().cpu().numpy())
        return outs
    ref = run(False)
    for prologue in (True, False, None):
        model.use_decode_attn_partials = prologue is not None      # (None: LayerNorm on the way in, attention with its own merge)
        got = run(True, prologue is not False)
        for step, (a, b) in enumerate(zip(got, ref)):
            err = np.abs(a - b).max() / np.abs(b).max()
            assert err < 1e-2, f"call {step} (LayerNorm on the way {'in' if prologue else 'out'}): rel err {err:.3e}"
    model.use_decode_fused = model.use_decode_ln_prologue = model.use_decode_attn_partials = True
    one = GraphedRingStep(model, batch_size=2, n_new=1)
    ids = calls[1]
    first = one(ids)[0].float().cpu().numpy()
    mems = RingMemory(model, 2)
    with torch.no_grad():
        x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
        eager = model([x], compute_loss=False, mems=mems)[0].float().cpu().numpy()
    assert np.array_equal(first, eager)


@pytest.mark.parametrize("B,q,mlen", [(1, 1, 1024), (2, 1, 300), (1, 2, 1000), (1, 1, 0)])
def test_output_projection_merges_the_attention_partials(B, q, mlen):
    """db1_relattn_decode_ring_fwd with out == NULL + db1_linear_decode_attn (the output projection merges the per-chunk partial results on
    its way in) equals the attention with its own merge (in-launch by the last chunk, and as a separate launch) followed by the plain projection"""
    from bdm_db1_amd import ops
    H, D, cap = 16, 128, 1024 + 64
    d = H * D
    g = torch.Generator(device=DEV).manual_seed(B * 100 + q * 10 + mlen)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=g) * sc).to(torch.bfloat16)
    qkv, u, vb = rnd(B, q, 3, H, D), rnd(H, D, sc=0.3), rnd(H, D, sc=0.3)
    ring0, R, W = rnd(B, cap, 2, H, D), rnd(cap, d), rnd(d, d, sc=0.03)
    state = torch.tensor([777], dtype=torch.int32, device=DEV)
    scale, shift = 1.0 / math.sqrt(D), mlen + q
    outs = []
    for mode in ("in-launch", "separate", "partials"):
        ring = ring0.clone()
        o = torch.empty(B * q, d, device=DEV, dtype=torch.bfloat16)
        if mode == "partials":
            assert ops.linear_decode_attn_supported(B, q, H, D, mlen + q, d)
            part = torch.empty(ops.relattn_decode_ring_part_numel(B, q, mlen + q, H), device=DEV, dtype=torch.float32)
            ops.relattn_decode_ring_fwd(qkv, u, vb, ring, state, R, None, B, q, mlen, H, D, shift, scale, part=part)
            ops.linear_decode_attn(part, mlen + q, B, q, H, D, W, o)
        else:
            av = torch.empty(B, q, H, D, device=DEV, dtype=torch.bfloat16)
            ops.relattn_decode_ring_fwd(qkv, u, vb, ring, state, R, av, B, q, mlen, H, D, shift, scale, fused_merge=mode == "in-launch")
            ops.linear_decode(av.view(B * q, d), W, None, o)
        outs.append((o, ring))
    for o, ring in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(ring, outs[0][1])
    assert int(ops.decode_tickets(qkv.device).abs().sum().item()) == 0


@pytest.mark.parametrize("mem_len", [1024, 1400])
def test_decode_chain_matches_the_separate_launches_at_db1_1p3b_geometry(mem_len):
    """db1_decode_chain (one persistent launch per layer for o_net + LN + ff1 / GEGLU + ff2 + LN + the next layer's qkv projection, fed by the
    attention's chunk partials) against the five launches per layer it replaces, over 1-token calls on a K / V ring at the 1.3B layer
    geometry (3 layers, mem_len 1024 / 1400): same logits to fp32 summation order + bf16 rounding, no stage wait ran into its spin limit, and the
    hipGraph-captured call equals the eager one bit for bit."""
    from bdm_db1_amd import GraphedRingStep, RingMemory, TransformerXL, synth, ops
    from bdm_db1_amd.data import NLPTaskInput
    cfg = synth.db1_config("1.3B", n_layer=3, mem_len=mem_len, n_position=max(1024, mem_len))   # (1400 + 1 keys: 12 chunks, the merge's other unrolling)
    torch.manual_seed(11)
    model = TransformerXL(cfg, device=torch.device(DEV), compute_dtype=torch.bfloat16)
    model.eval()
    rng = np.random.default_rng(4)
    calls = [rng.integers(0, 32000, (1, q)) for q in (22, 1, 1, 1, 9, 1, 1)]

    def run(chain):
        model.use_decode_chain = chain
        mems = RingMemory(model, 1)
        outs = []
        with torch.no_grad():
            for ids in calls:
                x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=torch.from_numpy(ids).to(DEV), text_len=None)
                logits, _, mems = model([x], compute_loss=False, mems=mems)
                outs.append(logits.float().cpu().numpy())
        return outs
    ref, got = run(False), run(True)
    assert model._chain_watch is not None, "the one-token calls did not go through db1_decode_chain"
    model.check_decode_chain(synchronize=True)   # raises if a stage wait of a chain launch ran into its spin limit
    for step, (a, b) in enumerate(zip(got, ref)):
        err = np.abs(a - b).max() / np.abs(b).max()
        assert err < 1e-2, f"call {step} (q = {calls[step].shape[1]}): rel err {err:.3e}"
    assert any(c.shape[1] == 1 for c in calls)
    # graphed = eager, both through the chain launch
    model.use_decode_chain = True
    ids = torch.from_numpy(rng.integers(0, 32000, (1, 1))).to(DEV)
    g = GraphedRingStep(model, 1, 1)
    mem_e = RingMemory(model, 1)
    with torch.no_grad():
        for _ in range(3):
            lg_g, _ = g(ids)
            x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
            lg_e, _, mem_e = model([x], compute_loss=False, mems=mem_e)
            assert torch.equal(lg_g.float(), lg_e.float())
    g.check(synchronize=True)
    model.check_decode_chain(synchronize=True)
    # a captured step owns its scratch and its pinned flag word, both created BEFORE the capture -- no zero-fill node inside the
    # graph, so a failed hand-off's sticky flag survives replays nobody synchronised on (it used to be wiped at the start of every replay);
    # and checking the step's flag leaves the model's own pending watch (its eager calls) alone
    from bdm_db1_amd import lib as db1lib
    off = int(db1lib.load().db1_decode_chain_error_offset())
    g2 = GraphedRingStep(model, 1, 1)
    assert g2._watch is not None and g2._watch[1] is g2._scratch and g2._scratch.data_ptr() != ops.decode_chain_scratch(model.dev).data_ptr()
    eager_watch = model._chain_watch
    with torch.no_grad():
        g2(ids)
        torch.cuda.synchronize()
        g2._scratch[off:off + 4].view(torch.int32).fill_(1)          # (simulated: a poll of this step's launches ran into its limit)
        g2.graph.replay()
        g2.graph.replay()                                            # two replays, no synchronisation, no check in between
        with pytest.raises(db1lib.Db1Error, match="hand-off poll"):
            g2.check(synchronize=True)
    assert model._chain_watch is eager_watch, "checking a captured step's flag must not replace the model's pending watch"
    assert model.use_decode_chain is False
    with pytest.raises(RuntimeError):
        g2(ids)                                                      # the step must not be replayed again
    model.use_decode_chain = True
    g.check(synchronize=True)                                        # the first step's scratch is its own: its flag is clean
    # the product path acts on the flag: a launch that could not hand off (simulated: the sticky flag of the eager stream's
    # scratch is set by hand) makes the NEXT forward over a ring raise, switches the persistent path off, and the per-launch path then
    # serves the same call
    x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
    with torch.no_grad():
        ops.decode_chain_scratch(model.dev)[off:off + 4].view(torch.int32).fill_(1)    # (the scratch of THIS stream: the graph above has its own)
        lg_bad, _, mem_e = model([x], compute_loss=False, mems=mem_e)      # this call's flag copy carries the 1
        lg_bad.float().cpu()                                               # (a caller reading its logits synchronises)
        with pytest.raises(db1lib.Db1Error, match="hand-off poll"):
            model([x], compute_loss=False, mems=mem_e)
        assert model.use_decode_chain is False and model._chain_watch is None
        lg_ok, _, mem_e = model([x], compute_loss=False, mems=mem_e)       # per-launch path
        assert torch.isfinite(lg_ok.float()).all() and model._chain_watch is None
    model.use_decode_chain = True


def test_decode_chain_stages_against_fp32_arithmetic():
    """db1_decode_chain alone, stage by stage: the rows it leaves in its scratch (tagged words: y_o, act, f) and its outputs (h1_out, f_out,
    x_next, qkv_next) against the same chain in fp32 torch arithmetic with bf16 rounding where the launches it replaces store bf16
    (transformer_xl.py:227-243,246-292,136), from random chunk partials; both forms of the launch (with the next layer's projection / last
    layer); every word of the scratch rows carries the launch's tag (slot + 1)."""
    from bdm_db1_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    d, dff, H, D, nunit = 2048, 4096, 16, 128, 9
    if not ops.decode_chain_supported(d, dff, H, D, nunit * 128):
        pytest.skip("db1_decode_chain needs 256 CUs")
    torch.manual_seed(0)
    bf16 = torch.bfloat16
    r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(bf16)
    x = r(1, d)
    Wo, W1, W2, Wq = r(d, d, sc=0.02), r(2 * dff, d, sc=0.02), r(d, dff, sc=0.02), r(3 * d, d, sc=0.02)
    b1, b2 = r(2 * dff, sc=0.1), r(d, sc=0.1)
    g1, be1 = (1 + 0.1 * torch.randn(d, device=dev)).to(bf16), r(d, sc=0.1)
    g2, be2 = (1 + 0.1 * torch.randn(d, device=dev)).to(bf16), r(d, sc=0.1)
    part = torch.full((H, nunit, 64, D + 2), float("nan"), device=dev)     # only query row 0 of every unit is read
    part[:, :, 0, :D] = torch.randn(H, nunit, D, device=dev)
    part[:, :, 0, D] = torch.randn(H, nunit, device=dev)
    part[:, :, 0, D + 1] = torch.rand(H, nunit, device=dev) + 0.5
    part[3, 2, 0, D + 1] = 0.0                                               # a chunk without visible keys contributes nothing
    alpha, eps = 1.3, 1e-5
    h1o, fo = torch.zeros(1, d, device=dev, dtype=bf16), torch.zeros(1, d, device=dev, dtype=bf16)
    xn, qn = torch.zeros(1, d, device=dev, dtype=bf16), torch.zeros(1, 3 * d, device=dev, dtype=bf16)
    rb = lambda t: t.to(bf16).float()
    m, l, o = part[:, :, 0, D], part[:, :, 0, D + 1], part[:, :, 0, :D]
    wt = torch.where(l > 0, torch.exp(m - torch.where(l > 0, m, torch.full_like(m, -1e30)).max(1, keepdim=True).values), torch.zeros_like(m))
    merged = rb(((o * wt[..., None]).sum(1) / (l * wt).sum(1, keepdim=True)).reshape(1, d))

    def ln(s, g, b):
        s = rb(s)
        mu = s.mean(-1, keepdim=True)
        return rb((s - mu) * torch.rsqrt(((s - mu) ** 2).mean(-1, keepdim=True) + eps) * g.float() + b.float())
    y_ref = rb(merged @ Wo.float().t())
    h1 = ln(alpha * x.float() + y_ref, g1, be1)
    z = rb(h1 @ W1.float().t() + b1.float())
    act_ref = rb(z[:, :dff] * torch.nn.functional.gelu(z[:, dff:]))
    f_ref = rb(act_ref @ W2.float().t() + b2.float())
    xn_ref = ln(alpha * h1 + f_ref, g2, be2)
    q_ref = rb(xn_ref @ Wq.float().t())
    err = lambda a, b: ((a.reshape(-1).float() - b.reshape(-1)).abs().max() / b.abs().max()).item()

    def rows():
        w = ops.decode_chain_scratch(dev)[: (2 * d + dff) * 4].view(torch.int32)
        val = (w >> 16).to(torch.int16).view(bf16).float()
        return val[:d], val[d:d + dff], val[d + dff:], (w & 0xffff)
    for slot, wq in ((4, Wq), (5, None)):     # (consecutive launches use different slots)
        last = wq is None
        ops.decode_chain(part, nunit * 128, H, x, Wo, W1, b1, W2, b2, wq, g1, be1, g2, be2, alpha, eps, h1o if last else None, fo if last else None,
                         None if last else xn, None if last else qn, slot, w_o_next=Wo)
        torch.cuda.synchronize()
        assert not ops.decode_chain_error(dev)
        y_o, act, f, tags = rows()
        assert tags.unique().tolist() == [slot + 1]
        assert err(y_o, y_ref) < 4e-3 and err(act, act_ref) < 1e-2 and err(f, f_ref) < 1e-2
        if last:
            assert err(h1o, h1) < 5e-3 and err(fo, f_ref) < 1e-2
        else:
            assert err(xn, xn_ref) < 1e-2 and err(qn, q_ref) < 1e-2


@pytest.mark.parametrize("M", [4, 16])
def test_batched_ring_decode_equals_independent_environments(M):
    """M environments decoded together over one RingMemory(model, M) -- one weight stream per token for all of them (bench.py's `decode`
    block: tokens/s at M = 1 / 4 / 16) -- against the same M token streams decoded one environment at a time (batch 1: the persistent
    one-token launch), at the 1.3B layer geometry with a full memory: the same logits per environment to bf16 rounding of the
    intermediate rows, and the graphed batched call equals the eager one bit for bit."""
    from bdm_db1_amd import GraphedRingStep, RingMemory, TransformerXL, synth
    from bdm_db1_amd.data import NLPTaskInput
    cfg = synth.db1_config("1.3B", n_layer=2)
    torch.manual_seed(3)
    model = TransformerXL(cfg, device=torch.device(DEV), compute_dtype=torch.bfloat16)
    model.eval()
    rng = np.random.default_rng(M)
    calls = [rng.integers(0, 32000, (M, q)) for q in (22, 1, 1, 1)]
    mk = lambda ids: NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=torch.from_numpy(ids).to(DEV), text_len=None)
    with torch.no_grad():
        mems = RingMemory(model, M)
        together = []
        for ids in calls:
            lg, _, mems = model([mk(ids)], compute_loss=False, mems=mems)
            together.append(lg.float().cpu().numpy())
        for env in (0, M // 2, M - 1):
            m1 = RingMemory(model, 1)
            for step, ids in enumerate(calls):
                lg, _, m1 = model([mk(ids[env:env + 1])], compute_loss=False, mems=m1)
                a, b = together[step][env], lg.float().cpu().numpy()[0]
                err = np.abs(a - b).max() / np.abs(b).max()
                assert err < 1e-2, (env, step, err)
        model.check_decode_chain(synchronize=True)
        g = GraphedRingStep(model, batch_size=M, n_new=1)
        mem_e = RingMemory(model, M)
        ids = torch.from_numpy(calls[1]).to(DEV)
        for _ in range(3):
            lg_g, _ = g(ids)
            lg_e, _, mem_e = model([mk(calls[1])], compute_loss=False, mems=mem_e)
            assert torch.equal(lg_g, lg_e)
