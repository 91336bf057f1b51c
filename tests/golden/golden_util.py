"""Shared by make_golden.py (runs where /root/reference exists) and the tests.

Fixtures hold only arrays: inputs and the reference's outputs.  Weights are NOT
stored: both sides regenerate them from a seeded NumPy stream with a fixed draw
order (``make_params``), which keeps the fixtures to a few hundred KB.
"""
from __future__ import annotations

import numpy as np

CFG_DEFAULTS = dict(
    n_embed=128, n_position=256, n_layer=2, n_head=4, n_inner=None, pre_lnorm=False, mem_len=256,
    same_length=True, untie_r=False, text_vocab_size=32000, num_discrete_values=1024,
    num_continuous_bin=1024, overlap_with_text=True, embd_pdrop=0.0, drop=0.0, dropattn=0.0,
    activation_fn="geglu", layer_norm_epsilon=1e-5, share_input_output_embedding=True,
    use_deepnorm=False, fp16=False, vision_patch_size=16, vision_num_input_channels=3,
    vision_position_vocab_size=128, vision_hidden_dropout_prob=0.0,
)

# the fixture cases: name -> config overrides
CASES = {
    # BASELINE.json configs[0]: DB1-tiny text-only seq-len 256
    "tiny_nlp": dict(),
    # mixed RL(+image) / NLP / caption batch, small vocabulary
    "small_mixed": dict(n_embed=64, n_head=2, n_position=48, mem_len=48, text_vocab_size=300,
                        num_continuous_bin=64, num_discrete_values=64),
    # sliding-window mask (0 < mem_len < L)
    "small_window": dict(n_embed=64, n_head=2, n_position=48, mem_len=16, text_vocab_size=300,
                         num_continuous_bin=64, num_discrete_values=64),
    # the other flag values: gelu FF, untied u/v, separate lm_head, deepnorm, no vocabulary overlap
    "small_flags": dict(n_embed=64, n_head=4, n_position=40, mem_len=40, text_vocab_size=200,
                        num_continuous_bin=32, num_discrete_values=16, overlap_with_text=False,
                        activation_fn="gelu", untie_r=True, share_input_output_embedding=False,
                        use_deepnorm=True, n_inner=96),
    "small_prelnorm": dict(n_embed=64, n_head=2, n_position=40, mem_len=40, text_vocab_size=200,
                           num_continuous_bin=32, num_discrete_values=32, pre_lnorm=True, same_length=False),
    # inference with Transformer-XL memory (evaluate_rl.py:157-266 call pattern)
    "small_mems": dict(n_embed=64, n_head=2, n_position=24, mem_len=24, text_vocab_size=300,
                       num_continuous_bin=64, num_discrete_values=64, n_layer=3),
    # VQA input path (_forward_vqa, transformer_xl.py:705-748) next to a text task
    "small_vqa": dict(n_embed=64, n_head=2, n_position=40, mem_len=40, text_vocab_size=300,
                      num_continuous_bin=64, num_discrete_values=64),
    # memory inference at d_head = 128, the head size of DB1-1.3B: what the bf16 K/V-cached decode kernels need
    "mems_d128": dict(n_embed=256, n_head=2, n_position=32, mem_len=32, text_vocab_size=300,
                      num_continuous_bin=64, num_discrete_values=64, n_layer=2),
    # the default mem_len = 0 under same_length: EVERY key hidden -> the reference attends uniformly to all keys (no error)
    "small_memlen0": dict(n_embed=64, n_head=2, n_position=24, mem_len=0, text_vocab_size=300,
                          num_continuous_bin=64, num_discrete_values=64),
}
MEM_CASES = {"small_mems": (5, 1, 7), "mems_d128": (6, 1, 1, 9, 1)}   # query lengths of the consecutive calls with memory


def case_cfg(name: str) -> dict:
    c = dict(CFG_DEFAULTS)
    c.update(CASES[name])
    return c


def param_shapes(cfg: dict):
    """state_dict names/shapes in a fixed order (ic_encoder.* aliases and tied per-layer
    r_*_bias aliases omitted; pos_emb.inv_freq is a buffer and is stored in the fixture)."""
    d, H = cfg["n_embed"], cfg["n_head"]
    D = d // H
    di = 4 * d if cfg["n_inner"] is None else cfg["n_inner"]
    V = cfg["text_vocab_size"] + cfg["num_continuous_bin"] + (0 if cfg["overlap_with_text"] else cfg["num_discrete_values"]) + 1
    ps, C = cfg["vision_patch_size"], cfg["vision_num_input_channels"]
    out = []
    if not cfg["untie_r"]:
        out += [("r_w_bias", (H, D)), ("r_r_bias", (H, D))]
    out += [("word_embedding.weight", (V, d))]
    pe = "vision_encoder.patch_embeddings."
    out += [(pe + "conv1.weight", (64, C, 3, 3)), (pe + "conv1.bias", (64,)),
            (pe + "projection.weight", (d, 64, ps, ps)), (pe + "projection.bias", (d,)),
            (pe + "residual_path.0.weight", (64,)), (pe + "residual_path.0.bias", (64,)),
            (pe + "residual_path.2.weight", (64, 64, 3, 3)), (pe + "residual_path.2.bias", (64,)),
            (pe + "residual_path.3.weight", (64,)), (pe + "residual_path.3.bias", (64,)),
            (pe + "residual_path.5.weight", (64, 64, 3, 3)), (pe + "residual_path.5.bias", (64,)),
            ("vision_encoder.row_position_embeddings.weight", (cfg["vision_position_vocab_size"], d)),
            ("vision_encoder.col_position_embeddings.weight", (cfg["vision_position_vocab_size"], d)),
            ("rl_local_timestep_embedding.weight", (513, d))]
    for i in range(cfg["n_layer"]):
        p = f"h.{i}."
        if cfg["untie_r"]:
            out += [(p + "dec_attn.r_r_bias", (H, D)), (p + "dec_attn.r_w_bias", (H, D))]
        out += [(p + "dec_attn.qkv_net.weight", (3 * d, d)), (p + "dec_attn.o_net.weight", (d, d)),
                (p + "dec_attn.r_net.weight", (d, d)),
                (p + "dec_attn.layer_norm.weight", (d,)), (p + "dec_attn.layer_norm.bias", (d,)),
                (p + "pos_ff.CoreNet.0.weight", (di, d)), (p + "pos_ff.CoreNet.0.bias", (di,)),
                (p + "pos_ff.CoreNet.2.weight", (d, di // 2 if cfg["activation_fn"] == "geglu" else di)),
                (p + "pos_ff.CoreNet.2.bias", (d,)),
                (p + "pos_ff.layer_norm.weight", (d,)), (p + "pos_ff.layer_norm.bias", (d,))]
    if not cfg["share_input_output_embedding"]:
        out += [("lm_head.weight", (V, d))]
    return out


def make_params(cfg: dict, seed: int):
    """Deterministic float32 weights.  Larger std than the reference's 0.02 init and
    non-trivial LN/GN/bias values so every term of the forward/backward is exercised."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape in param_shapes(cfg):
        if name.endswith("layer_norm.weight") or ("residual_path" in name and name.endswith(".weight") and len(shape) == 1):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith(".bias") or name.endswith("r_w_bias") or name.endswith("r_r_bias"):
            a = 0.05 * rng.standard_normal(shape)
        elif "projection.weight" in name:
            a = 0.01 * rng.standard_normal(shape)
        elif "conv" in name or "residual_path" in name:
            a = 0.06 * rng.standard_normal(shape)
        else:
            a = 0.05 * rng.standard_normal(shape)
        params[name] = a.astype(np.float32)
    return params


def sample_idx(n: int, k: int = 256):
    """Fixed subsample of a flat tensor of n elements (<= k entries)."""
    if n <= k:
        return np.arange(n)
    return (np.arange(k, dtype=np.int64) * (n // k)) + (np.arange(k) % 7) % max(1, n // k)


def make_batch(name: str, cfg: dict, seed: int):
    """Synthetic task inputs for a case: list of dicts with the reference's field names."""
    rng = np.random.default_rng(seed + 1000)
    L = cfg["n_position"]
    V_text = cfg["text_vocab_size"]
    sep = cfg["text_vocab_size"] + cfg["num_continuous_bin"] + (0 if cfg["overlap_with_text"] else cfg["num_discrete_values"])
    tasks = []

    def nlp(B):
        ids = rng.integers(0, V_text, size=(B, L + 1))
        lm = (rng.random((B, L)) > 0.2).astype(np.float32)
        lm[:, -1] = 1.0
        return dict(kind="nlp", text_seq=ids[:, :-1].copy(), label=ids[:, 1:].copy(), loss_mask=lm)

    if name == "tiny_nlp":
        ids = rng.integers(0, 32000, size=(8, L + 1))
        tasks.append(dict(kind="nlp", text_seq=ids[:, :-1].copy(), label=ids[:, 1:].copy(),
                          loss_mask=np.ones((8, L), np.float32)))
    elif name in ("small_mixed",):
        # RL: obs = one 32x32 image (4 patches) + 2 tensor tokens, SEP, 1 action -> 8 tokens/transition
        B, ntr = 2, L // 8
        seq = np.zeros((B, L + 1), np.int64)
        for b in range(B):
            row = []
            for t in range(ntr + 1):
                row += [-1] * 4 + list(V_text + rng.integers(0, cfg["num_continuous_bin"], 2)) + [sep] + [int(rng.integers(0, 18))]
            seq[b] = np.array(row[:L + 1])
        inp, lab = seq[:, :-1].copy(), seq[:, 1:].copy()
        step = 8
        within = np.arange(L) % step
        pos = np.where(within <= 6, within + 1, 0)
        lm = np.zeros((B, L), np.float32)
        lm[:, within == 6] = 1.0  # label at the separator position is the action token
        nimg = int((inp[0] == -1).sum()) // 4
        vision = rng.random((B, nimg, 3, 32, 32)).astype(np.float32) * 255.0
        tasks.append(dict(kind="rl", tensor_seq=inp, vision_seq=vision, position_id=np.tile(pos, (B, 1)),
                          label=lab, loss_mask=lm))
        tasks.append(nlp(2))
        # caption: prompt 4 + image 32x48 (6 patches) + text
        Bc, P, nv = 2, 4, 6
        T = L - P - nv
        prompt = rng.integers(0, V_text, size=(Bc, P))
        text = rng.integers(0, V_text, size=(Bc, T))
        img = rng.standard_normal((Bc, 3, 32, 48)).astype(np.float32)
        label = rng.integers(0, V_text, size=(Bc, L))
        lm = np.zeros((Bc, L), np.float32)
        lm[:, P + nv - 1:] = (rng.random((Bc, T + 1)) > 0.3)
        lm[:, P + nv - 1] = 1.0
        tasks.append(dict(kind="ic", prompt_seq=prompt, img_seq=img, text_seq=text, label=label, loss_mask=lm))
    elif name in ("small_window", "small_flags", "small_prelnorm", "small_memlen0"):
        tasks.append(nlp(3))
    elif name == "small_vqa":
        # prompt 3 + image 32x32 (4 patches) + text (question 9 ++ answer): _forward_vqa concatenates exactly like _forward_ic
        Bq, P, nv, ql = 2, 3, 4, 9
        T = L - P - nv
        prompt = rng.integers(0, V_text, size=(Bq, P))
        text = rng.integers(1, V_text, size=(Bq, T))
        img = rng.standard_normal((Bq, 3, 32, 32)).astype(np.float32)
        label = np.zeros((Bq, L), np.int64)
        al = T - ql + 1                                           # answer tokens: right-aligned labels (coco_token_dataset.py:183-192)
        label[:, L - al:] = rng.integers(0, V_text, size=(Bq, al))
        lm = np.zeros((Bq, L), np.float32)
        lm[:, L - al:] = 1.0
        lm[1, L - 3] = 0.0
        tasks.append(dict(kind="vqa", prompt_seq=prompt, img_seq=img, text_seq=text, label=label, loss_mask=lm,
                          ques_len=np.full((Bq,), ql, np.int64)))
        tasks.append(nlp(1))
    elif name in MEM_CASES:
        pass
    return tasks


# ---------------------------------------------------------------------------------------------------------------------
# inputs of the RL / caption sample-builder fixtures (rl_dataset.npz, caption_vqa.npz): regenerated on both sides
# ---------------------------------------------------------------------------------------------------------------------
RL_DS_CASES = {
    # name: (observation kind, action kind, seq_length, constructor options, sample indices, numpy global seed)
    "vec_cont_noprompt": ("vec5", "cont2", 64, dict(use_prompt=False), [0, 7, 40, 10 ** 6], 1),
    "vec_cont_prompt_subseq": ("vec5", "cont2", 64, dict(use_prompt=True, prompt_prob=1.0, prompt_at_final_transition_prob=0.0), [0, 3, 33], 2),
    "vec_cont_prompt_final": ("vec5", "cont2", 64, dict(use_prompt=True, prompt_prob=1.0, prompt_at_final_transition_prob=1.0), [1, 20], 3),
    "vec_cont_prompt_timestep": ("vec5", "cont2", 64, dict(use_prompt=True, prompt_prob=1.0, prompt_at_final_transition_prob=0.0,
                                                            prompt_strategy="stochastic_timestep"), [2, 11], 4),
    "vec_cont_prompt_mixed": ("vec5", "cont2", 96, dict(use_prompt=True, prompt_prob=0.5, prompt_ratio=0.25), [0, 5, 9, 14, 21, 30], 5),
    "img_disc": ("img32", "disc", 48, dict(use_prompt=False), [0, 2, 12], 6),
    "img_disc_nooverlap_prompt": ("img32", "disc", 48, dict(use_prompt=True, prompt_prob=1.0, overlap_with_text=False, num_discrete_values=18), [0, 6], 7),
    "dict_img_vec_disc": ("dict", "disc2", 80, dict(use_prompt=True, prompt_prob=0.7), [0, 4, 8], 8),
    "discobs_cont": ("disc3", "cont1", 40, dict(use_prompt=False, overlap_with_text=False), [0, 9], 9),
}


def rl_trajectories(obs_kind: str, act_kind: str, seed: int = 31):
    """a handful of short synthetic trajectories [(observations, actions)] of unequal lengths"""
    rng = np.random.default_rng(seed)
    out = []
    for n in (9, 4, 13, 2, 7):
        if obs_kind == "vec5":
            obs = (rng.standard_normal((n, 5)) * 3).astype(np.float32)
        elif obs_kind == "disc3":
            obs = rng.integers(0, 16, (n, 3)).astype(np.int64)
        elif obs_kind == "img32":
            obs = (rng.random((n, 3, 32, 32)) * 255).astype(np.float32)
        elif obs_kind == "dict":
            obs = {"rgb": (rng.random((n, 3, 16, 32)) * 255).astype(np.float32), "state": rng.standard_normal((n, 3)).astype(np.float32),
                   "aux": rng.integers(0, 10, (n, 2)).astype(np.int64)}
        else:
            raise ValueError(obs_kind)
        if act_kind.startswith("cont"):
            act = rng.uniform(-1, 1, (n, int(act_kind[4:]))).astype(np.float32)
        elif act_kind == "disc":
            act = rng.integers(0, 18, (n,)).astype(np.int64)
        else:
            act = rng.integers(0, 18, (n, int(act_kind[4:]))).astype(np.int64)
        out.append((obs, act))
    return out


def caption_samples(seed: int = 41, eos: int = 0):
    """samples as the reference's RandomCOCO / VQA readers yield them (dicts of token arrays and an image tensor)"""
    import torch
    rng = np.random.default_rng(seed)
    ic, vqa = [], []
    for k, tlen in enumerate((12, 30, 5)):
        text = rng.integers(1, 300, tlen).astype(np.int32)
        text[rng.integers(0, tlen)] = eos                      # an eos inside / at the end: masked positions
        ic.append(dict(text=text, img=torch.from_numpy(rng.standard_normal((3, 32, 48)).astype(np.float32)),
                       prompt=[5, 6, 7, 8], img_id=100 + k))
    for k, (ql, al) in enumerate(((6, 3), (9, 2), (4, 5))):
        ans = rng.integers(1, 300, al).astype(np.int32)
        ans[-1] = eos
        vqa.append(dict(ques=rng.integers(1, 300, ql).astype(np.int32), ans=ans if k != 1 else ans.tolist(),
                        img=torch.from_numpy(rng.standard_normal((3, 32, 32)).astype(np.float32)), ques_id=7000 + k, img_id=200 + k,
                        prompt=np.array([9, 10, 11], np.int32), ques_len=ql))
    return ic, vqa


# ---------------------------------------------------------------------------------------------------------------------
# action decoding (get_action.npz): the call sequences both sides replay
# ---------------------------------------------------------------------------------------------------------------------
GET_ACTION_CASES = {
    # name: (use memory, discrete action, obs_length, action_length, steps, prompt strategy, use_prompt, fixed prompt length)
    "mem_continuous": (True, False, 4, 2, 3, "moving_prompt", False, 0),
    "mem_discrete_masked": (True, True, 3, 1, 3, "moving_prompt", False, 0),
    "window_continuous": (False, False, 4, 2, 4, "moving_prompt", False, 0),
    "window_fixed_prompt": (False, False, 4, 2, 4, "fixed_prompt", True, 7),
}


def get_action_inputs(name: str, cfg: dict, seed: int = 61):
    """per environment step: the new observation tokens (continuous-bin ids) the episode loop would append, + the separator"""
    mem, disc, ol, al, steps, strat, use_prompt, lfp = GET_ACTION_CASES[name]
    rng = np.random.default_rng(seed + len(name))
    tv, nb = cfg["text_vocab_size"], cfg["num_continuous_bin"]
    sep = tv + nb + (0 if cfg["overlap_with_text"] else cfg["num_discrete_values"])
    obs = [np.concatenate([tv + rng.integers(0, nb, ol), [sep]]).astype(np.int64) for _ in range(steps)]
    prompt = np.concatenate([tv + rng.integers(0, nb, ol), [sep], tv + rng.integers(0, nb, al)]).astype(np.int64) if lfp else None
    masks = [None, np.array([1, 0, 1, 0, 0, 1], np.float64), None] if name == "mem_discrete_masked" else [None] * steps
    return obs, prompt, masks
