#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference).

Run only in the build container (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Nothing from the reference is copied: the fixtures contain input arrays and the
reference's output arrays only.  Weights are regenerated on both sides from
``golden_util.make_params``.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DB1_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import torch  # noqa: E402
from golden_util import CASES, MEM_CASES, case_cfg, make_params, make_batch, sample_idx  # noqa: E402

torch.set_num_threads(8)


def build_ref_model(cfg, params):
    from src.model import TransformerXL
    m = TransformerXL(SimpleNamespace(**cfg))
    sd = m.state_dict()
    new = {}
    for k, v in sd.items():
        k2 = k.replace("ic_encoder.", "vision_encoder.")
        if k2 in params:
            new[k] = torch.from_numpy(params[k2])
        elif not cfg["untie_r"] and (k.endswith("dec_attn.r_w_bias") or k.endswith("dec_attn.r_r_bias")):
            new[k] = torch.from_numpy(params[k.split(".")[-1]])
        else:
            assert k == "pos_emb.inv_freq", k
            new[k] = v
    m.load_state_dict(new)
    m.eval()  # deterministic vision position ids; all dropout p are 0 anyway
    return m


def to_ref_inputs(tasks):
    from src.data.input_specs import NLPTaskInput, RLTaskInput, ICTaskInput, VQATaskInput
    T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
    out = []
    for t in tasks:
        base = dict(position_id=T(t.get("position_id")), attention_mask=None, loss_mask=T(t.get("loss_mask")), label=T(t.get("label")))
        if t["kind"] == "nlp":
            out.append(NLPTaskInput(text_seq=T(t["text_seq"]), text_len=None, **base))
        elif t["kind"] == "rl":
            out.append(RLTaskInput(text_seq=None, vision_seq=T(t["vision_seq"]), tensor_seq=T(t["tensor_seq"]), **base))
        elif t["kind"] == "ic":
            out.append(ICTaskInput(prompt_seq=T(t["prompt_seq"]), img_seq=T(t["img_seq"]), text_seq=T(t["text_seq"]), img_id_seq=None, **base))
        elif t["kind"] == "vqa":
            out.append(VQATaskInput(prompt_seq=T(t["prompt_seq"]), img_seq=T(t["img_seq"]), text_seq=T(t["text_seq"]), img_id_seq=None,
                                    ques_id_seq=None, ques_len=T(t["ques_len"]), **base))
    return out


def gen_model_case(name, seed):
    cfg = case_cfg(name)
    params = make_params(cfg, seed)
    m = build_ref_model(cfg, params)
    out = {"inv_freq": m.pos_emb.inv_freq.numpy().copy()}
    if name in MEM_CASES:
        # consecutive calls with memory (small_mems: qlen 5 / 1 / 7)
        rng = np.random.default_rng(seed + 7)
        mems = m.init_mem(2)
        for step, q in enumerate(MEM_CASES[name]):
            ids = rng.integers(0, cfg["text_vocab_size"], size=(2, q))
            from src.data.input_specs import NLPTaskInput
            x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None,
                             text_seq=torch.from_numpy(ids), text_len=None)
            with torch.no_grad():
                logits, _, mems = m([x], compute_loss=False, mems=mems)
            out[f"ids{step}"] = ids
            out[f"logits{step}"] = logits.numpy()
            out[f"mem_last{step}"] = mems[-1].numpy()
        return out
    tasks = make_batch(name, cfg, seed)
    logits, loss = m(to_ref_inputs(tasks))
    loss.backward()
    lg = logits.detach().numpy()
    out["loss"] = np.float64(loss.item())
    out["logits_shape"] = np.array(lg.shape)
    out["logits_sample"] = lg.reshape(-1)[sample_idx(lg.size, 4096)]
    out["logits_norm"] = np.float64(np.sqrt((lg.astype(np.float64) ** 2).sum()))
    for k, p in m.named_parameters():
        if k.startswith("ic_encoder."):
            continue
        if not cfg["untie_r"] and ("dec_attn.r_w_bias" in k or "dec_attn.r_r_bias" in k):
            continue
        g = p.grad
        g = np.zeros(p.shape, np.float32) if g is None else g.numpy()
        out["gnorm/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out["gsample/" + k] = g.reshape(-1)[sample_idx(g.size)]
    return out


def gen_patch_embed(seed):
    from src.tokenizer.vision_embedding import PatchEmbeddings
    cfg = case_cfg("small_mixed")
    params = make_params(cfg, seed)
    pe = PatchEmbeddings(16, 3, cfg["n_embed"], data_type=torch.float32)
    pe.load_state_dict({k.replace("vision_encoder.patch_embeddings.", ""): torch.from_numpy(v)
                        for k, v in params.items() if k.startswith("vision_encoder.patch_embeddings.")})
    rng = np.random.default_rng(seed + 3)
    img = (rng.random((3, 3, 32, 48)) * 255).astype(np.float32)
    y = pe(torch.from_numpy(img))
    G = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    (y * torch.from_numpy(G)).sum().backward()
    out = {"img": img, "G": G, "y": y.detach().numpy()}
    for k, p in pe.named_parameters():
        out["grad/" + k] = p.grad.numpy()
    return out


def gen_scalar_tokenizer(seed):
    from src.tokenizer.scalar_tokenizer import ContinuousScalarTokenizer
    tok = ContinuousScalarTokenizer()
    rng = np.random.default_rng(seed)
    known_obs = np.array([-300, -1, -.01, 0, .01, .5, 1, 10, 256, 1e4], np.float32)
    known_act = np.array([-1, -.5, 0, .5, .999, 1], np.float32)
    # random sweep over several magnitudes + exact bin boundaries of the action path and their fp32 neighbours
    sweep = np.concatenate([
        rng.standard_normal(40000) * s for s in (0.01, 0.3, 3.0, 50.0, 400.0)]).astype(np.float32)
    edges = (np.arange(0, 1025, dtype=np.float64) / 512.0 - 1.0).astype(np.float32)
    act = np.concatenate([edges, np.nextafter(edges, np.float32(2)), np.nextafter(edges, np.float32(-2)),
                          rng.uniform(-1.2, 1.2, 20000).astype(np.float32)])
    # mu-law bin boundaries mapped back to observation space, and neighbours
    t = edges.astype(np.float64)
    xb = (np.sign(t) * (np.power(25601.0, np.abs(t)) - 1.0) / 100.0).astype(np.float32)
    nb = [xb]
    for _ in range(3):
        nb.append(np.nextafter(nb[-1], np.float32(1e9)))
    lo = xb
    for _ in range(3):
        lo = np.nextafter(lo, np.float32(-1e9))
        nb.append(lo)
    obs = np.concatenate([sweep] + nb)
    D = lambda x, a: tok.discretize(torch.from_numpy(x.copy()), a).numpy().astype(np.int32)
    ids = np.arange(0, 1024, dtype=np.int64)
    return dict(known_obs=known_obs, known_obs_ids=D(known_obs, False), known_act=known_act, known_act_ids=D(known_act, True),
                obs=obs, obs_ids=D(obs, False), act=act, act_ids=D(act, True),
                dec_ids=ids, dec_obs=tok.decode(torch.from_numpy(ids), False).numpy(), dec_act=tok.decode(torch.from_numpy(ids), True).numpy())


def gen_adam(seed):
    rng = np.random.default_rng(seed)
    out = {}
    for mode in ("adam", "adamw"):
        p0 = rng.standard_normal(257).astype(np.float32)
        gs = [rng.standard_normal(257).astype(np.float32) * s for s in (1.0, 0.1, 3.0)]
        p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
        cls = torch.optim.Adam if mode == "adam" else torch.optim.AdamW
        opt = cls([p], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
        out[f"{mode}/p0"] = p0
        for i, g in enumerate(gs):
            p.grad = torch.from_numpy(g.copy())
            norm = torch.nn.utils.clip_grad_norm_([p], 1.0)
            opt.step()
            out[f"{mode}/g{i}"] = g
            out[f"{mode}/norm{i}"] = np.float64(norm.item())
            out[f"{mode}/p{i + 1}"] = p.detach().numpy().copy()
            st = opt.state[p]
            out[f"{mode}/m{i + 1}"] = st["exp_avg"].numpy().copy()
            out[f"{mode}/v{i + 1}"] = st["exp_avg_sq"].numpy().copy()
    return out


def gen_scheduler():
    import src.mpu  # noqa: F401  (print_rank_0)
    from src.train_utils.optimizer_param_scheduler import OptimizerParamScheduler
    out = {}
    steps = np.array([0, 1, 5, 10, 11, 50, 99, 100, 101, 150], np.int64)
    for style in ("constant", "linear", "cosine"):
        for wstyle in ("constant", "linear", "cosine"):
            opt = SimpleNamespace(param_groups=[{"lr": 0.0, "weight_decay": 0.0}])
            s = OptimizerParamScheduler(opt, max_lr=1e-3, min_lr=1e-5, lr_warmup_steps=10, lr_decay_steps=100,
                                        lr_decay_style=style, start_wd=0.01 if wstyle == "constant" else 0.0, end_wd=0.01,
                                        wd_incr_steps=80, wd_incr_style=wstyle)
            lrs, wds = [], []
            prev = 0
            for st in steps:
                s.step(int(st - prev))
                prev = st
                lrs.append(opt.param_groups[0]["lr"])
                wds.append(opt.param_groups[0]["weight_decay"])
            out[f"lr/{style}/{wstyle}"] = np.array(lrs)
            out[f"wd/{style}/{wstyle}"] = np.array(wds)
    out["steps"] = steps
    return out


def gen_rl_packing():
    for n in ("gym", "d4rl", "tree"):
        sys.modules.setdefault(n, types.ModuleType(n))
    try:
        from src.data.rl_dataset import _get_action_flag_and_position_id, _truncate_or_pad_to_match_seq_len
    except Exception as e:  # pragma: no cover
        print("rl_dataset helpers not importable:", repr(e))
        return None
    out = {}
    cases = [(0, 63, 6, 1, 0), (0, 1023, 20, 1, 0), (0, 99, 7, 3, 2), (0, 10, 4, 2, 0), (22, 1046, 20, 1, 3)]
    for i, c in enumerate(cases):
        f, p = _get_action_flag_and_position_id(*c)
        out[f"args{i}"] = np.array(c)
        out[f"flag{i}"] = f
        out[f"pos{i}"] = p
    out["pad_in"] = np.arange(5)
    out["pad8"] = _truncate_or_pad_to_match_seq_len(np.arange(5), 8)
    out["pad3"] = _truncate_or_pad_to_match_seq_len(np.arange(5), 3)
    return out


class _FakeTextTokenizer:
    """the two things the sample builders read from the text tokenizer"""
    vocab_size = 32000
    eos_token_id = 0


def _tree_stub():
    """dm-tree is not installed: the three functions the reference's RL dataset uses, for the two observation forms it supports (one
    array, or a flat dict of arrays; dm-tree visits dict keys in sorted order and rebuilds the dict in the original key order)"""
    m = types.ModuleType("tree")

    def map_structure(fn, *structs):
        s0 = structs[0]
        if isinstance(s0, dict):
            vals = {k: map_structure(fn, *[s[k] for s in structs]) for k in sorted(s0)}
            return {k: vals[k] for k in s0}
        return fn(*structs)

    def flatten(s):
        return [x for k in sorted(s) for x in flatten(s[k])] if isinstance(s, dict) else [s]

    m.map_structure, m.flatten = map_structure, flatten
    return m


def gen_rl_dataset():
    """RLFullDataset.get (src/data/rl_dataset.py:590-752, with prepend_prompt :475-578 and postprocess_obs_and_act :393-473) run by
    the reference itself on in-memory trajectories.  gym / d4rl (a private fork) are absent, so __init__ (which opens a d4rl
    environment and an on-disk cache) is bypassed: the object is created with __new__, given the trajectories, and everything else
    -- type spec, dims, transition counts, index table -- is computed by the reference's own methods and its compiled helpers."""
    from golden_util import RL_DS_CASES, rl_trajectories
    for n in ("gym", "d4rl"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["tree"] = _tree_stub()
    from src.data import rl_dataset as R
    from src.tokenizer.scalar_tokenizer import ContinuousScalarTokenizer
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref"))
    import helpers as ref_helpers
    out = {}
    for name, (ok, ak, L, opts, idxs, seed) in RL_DS_CASES.items():
        trajs = rl_trajectories(ok, ak)
        ds = R.RLFullDataset.__new__(R.RLFullDataset)
        o = dict(overlap_with_text=True, num_discrete_values=1024, prompt_ratio=0.5, prompt_prob=0.25, prompt_at_final_transition_prob=0.5,
                 use_prompt=True, prompt_strategy="stochastic_subseq")
        o.update(opts)
        ds.env, ds.name, ds.output_sequence_length = object(), name, L
        ds.prompt_strategy, ds.use_prompt, ds.vision_patch_size = o["prompt_strategy"], o["use_prompt"], 16
        ds.prompt_prob, ds.prompt_at_final_transition_prob, ds.prompt_ratio = o["prompt_prob"], o["prompt_at_final_transition_prob"], o["prompt_ratio"]
        ds.text_tokenizer, ds.discretizer = _FakeTextTokenizer(), ContinuousScalarTokenizer()
        ds.num_discrete_values, ds.overlap_with_text = o["num_discrete_values"], o["overlap_with_text"]
        ds.is_lazy, ds.cached = False, False
        ds.observations, ds.actions = [t[0] for t in trajs], [t[1] for t in trajs]
        ds.path_lengths = np.array([len(a) for a in ds.actions], dtype=np.int32)
        tmp_obs, tmp_act = ds.get_obs_action_by_path_idx(0)                    # rl_dataset.py:220-236
        ds.obs_type_spec = ds.get_obs_type_spec(tmp_obs)
        ds.observation_dims_for_spec = ds.get_observation_dim(tmp_obs)
        ds.observation_dim = sum(sys.modules["tree"].flatten(ds.observation_dims_for_spec))
        ds.action_dim = ds.get_action_dim(tmp_act[0])
        trans_dim = ds.observation_dim + ds.action_dim
        ds.transition_num = (L + trans_dim) // (trans_dim + 1)
        ds.prompt_transition_num = int(o["prompt_ratio"] * ds.transition_num)
        ds.predicted_transition_num = ds.transition_num - ds.prompt_transition_num
        ds.indices = np.array(ref_helpers.build_rl_sample_idx(ds.path_lengths, ds.transition_num))
        out[f"{name}/meta"] = np.array([ds.observation_dim, ds.action_dim, ds.transition_num, ds.prompt_transition_num, len(ds.indices)])
        np.random.seed(seed)
        for j, idx in enumerate(idxs):
            r = ds.get(idx)
            for f in ("position_id", "loss_mask", "label", "tensor_seq"):
                out[f"{name}/{j}/{f}"] = getattr(r, f).numpy()
            if r.vision_seq is not None:
                out[f"{name}/{j}/vision_seq"] = r.vision_seq.numpy()
        if name == "vec_cont_noprompt":   # demonstrations for evaluation prompts (:812-862)
            np.random.seed(99)
            for j, (strategy, strict) in enumerate((("fixed_prompt", False), ("moving_prompt", True))):
                d = ds.sample_expert_demonstration(strategy, strict, False)
                out[f"{name}/demo{j}/actions"], out[f"{name}/demo{j}/tensor"] = d["actions"], d["obs/tensor"]
    return out


def gen_caption_vqa():
    """ICDataset / VQADataset / get_ltor_masks_and_position_ids / get_loss_mask_vqa (src/data/coco_token_dataset.py:58-210) and the
    caption length fitting of RandomCOCO.__getitem__ (:43-48) run by the reference; torchvision (absent) is only the base class of a
    reader that is not used here."""
    from golden_util import caption_samples
    tv = types.ModuleType("torchvision")
    tv.datasets = types.ModuleType("torchvision.datasets")
    tv.datasets.CocoCaptions = object
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.datasets", tv.datasets)
    from src.data import coco_token_dataset as C
    import torch.nn.functional as F
    ic, vqa = caption_samples()
    out = {}
    args = SimpleNamespace(n_position=48, eod_mask_loss=False)
    tok = _FakeTextTokenizer()
    for name, cls, samples in (("ic", C.ICDataset, ic), ("vqa", C.VQADataset, vqa)):
        ds = cls(args, samples, tok)
        for j in range(len(ds)):
            r = ds[j]
            for f in ("loss_mask", "label", "prompt_seq", "text_seq", "img_seq"):
                out[f"{name}/{j}/{f}"] = getattr(r, f).float().numpy() if f == "img_seq" else getattr(r, f).numpy()
    for j, (n, full) in enumerate(((5, 12), (11, 12), (1, 4))):
        data = np.array([3, 0, 7, 0, 9, 2, 0, 4, 1, 8, 6][:n], np.int32)
        _, lm, pid = C.get_ltor_masks_and_position_ids(data, 0, full)
        out[f"ltor/{j}/data"], out[f"ltor/{j}/full"], out[f"ltor/{j}/loss_mask"], out[f"ltor/{j}/position_ids"] = data, np.int64(full), lm, pid
    for j, seq_length in enumerate((4, 9, 20)):   # RandomCOCO.__getitem__ :43-48 on a 9-token caption
        text = torch.IntTensor([[11, 12, 13, 14, 15, 16, 17, 18, 19]]).squeeze()
        text = text[..., :seq_length] if text.shape[-1] >= seq_length else F.pad(text, (0, seq_length - text.shape[-1]), "constant", 0)
        out[f"fit/{j}"] = text.numpy()
    return out


def gen_gpt_dataset():
    """GPTDataset (src/data/gpt_dataset.py:86-180) on the fixture store written by the reference's builder, its index arrays from the
    reference's own _build_doc_idx / _build_shuffle_idx / _num_epochs and compiled helpers.build_sample_idx (the wrapper
    _build_index_mappings needs a CUDA tensor and an initialised process group for its barrier: its body is replayed here), and
    BlendableDataset (blendable_dataset.py:30-72)."""
    if not hasattr(np, "float"):
        np.float = np.float64
    from src.data import indexed_dataset as ref_idx
    from src.data import gpt_dataset as G
    from src.data.blendable_dataset import BlendableDataset
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref"))
    import helpers as ref_helpers
    ds = ref_idx.MMapIndexedDataset(os.path.join(HERE, "data_fixture"), skip_warmup=True)
    out = {}
    for k, (docs, seq, seed, eod_mask) in enumerate([(np.arange(11), 16, 1234, False), (np.array([0, 2, 3, 5, 6, 8, 9]), 7, 7, True),
                                                     (np.arange(3, 11), 50, 99, False)]):
        sizes = ds.sizes
        num_samples = np.sum(sizes[docs]) // seq
        tpe = G._num_tokens(docs, sizes)
        ne = G._num_epochs(tpe, seq, num_samples)
        rng = np.random.RandomState(seed=seed)
        if ne == 1:
            sep = False
        else:
            before = ((ne - 1) * tpe - 1) // seq
            sep = (num_samples - before) < int(0.80 * ((tpe - 1) // seq))
        doc_idx = G._build_doc_idx(docs, ne, rng, sep)
        sample_idx = np.array(ref_helpers.build_sample_idx(sizes, doc_idx, seq, ne, tpe))
        shuffle_idx = G._build_shuffle_idx(before if sep else sample_idx.shape[0] - 1, sample_idx.shape[0] - 1, rng)
        g = G.GPTDataset.__new__(G.GPTDataset)
        g.name, g.indexed_dataset, g.seq_length, g.eos_token_id = "t", ds, seq, 3
        g.reset_position_ids = g.reset_attention_mask = False
        g.eod_mask_loss = eod_mask
        g.doc_idx, g.sample_idx, g.shuffle_idx = doc_idx, sample_idx, shuffle_idx
        out[f"case{k}/args"] = np.array([seq, seed, int(eod_mask), ne, int(sep)])
        out[f"case{k}/docs"], out[f"case{k}/doc_idx"], out[f"case{k}/sample_idx"], out[f"case{k}/shuffle_idx"] = docs, doc_idx, sample_idx, shuffle_idx
        items = [g[i] for i in range(len(g))]
        out[f"case{k}/text_seq"] = np.concatenate([x.text_seq.numpy() for x in items])
        out[f"case{k}/label"] = np.concatenate([x.label.numpy() for x in items])
        out[f"case{k}/loss_mask"] = np.concatenate([x.loss_mask.numpy() for x in items])
        out[f"case{k}/position_id"] = items[0].position_id.numpy()
    # BlendableDataset: three list-backed datasets, global batch 8, weights 0.5 / 0.3 / 0.2
    dsets = [list(range(100, 110)), list(range(200, 205)), list(range(300, 303))]
    b = BlendableDataset(dsets, [0.5, 0.3, 0.2], global_batch_size=8)
    np.random.seed(5)
    out["blend/offsets"], out["blend/len"] = b.offset_in_batch, np.int64(len(b))
    out["blend/items"] = np.array([b[i] for i in range(40)])
    b2 = BlendableDataset(dsets, [1.0, 1.0, 2.0])
    np.random.seed(6)
    out["blend2/offsets"], out["blend2/items"] = b2.offset_in_batch, np.array([b2[i] for i in range(12)])
    return out


def gen_get_action():
    """get_action (src/evaluation/evaluate_rl.py:157-266) of the reference, on the reference model, replaying an episode's call pattern:
    with Transformer-XL memory (observation tokens, then one token per call, then the memorising call) and without (sliding window,
    fixed prompt).  deepspeed / gym / d4rl / tree are import-time dependencies of that module only: stubbed."""
    import importlib.machinery
    from golden_util import GET_ACTION_CASES, get_action_inputs
    gym = types.ModuleType("gym")
    gym.Wrapper, gym.Env = object, object
    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete = type("Discrete", (), {"__init__": lambda self, n: setattr(self, "n", n)})
    spaces.Box = type("Box", (), {})
    gym.spaces = spaces
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    for n in ("d4rl", "deepspeed"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["tree"] = _tree_stub()
    sys.modules["deepspeed"].DeepSpeedEngine = object
    for n in ("d4rl", "deepspeed", "tree", "gym"):
        sys.modules[n].__spec__ = importlib.machinery.ModuleSpec(n, None)
    from src.evaluation import evaluate_rl as E
    from src.tokenizer.scalar_tokenizer import ContinuousScalarTokenizer
    name = "small_mems"
    cfg = case_cfg(name)
    params = make_params(cfg, 100 + list(CASES).index(name))
    m = build_ref_model(cfg, params)
    m.device = torch.device("cpu")
    tok = ContinuousScalarTokenizer(cfg["num_continuous_bin"])
    out = {}
    for case, (mem, disc, ol, al, steps, strat, use_prompt, lfp) in GET_ACTION_CASES.items():
        args = SimpleNamespace(overlap_with_text=cfg["overlap_with_text"], text_vocab_size=cfg["text_vocab_size"], num_discrete_values=cfg["num_discrete_values"],
                               n_position=cfg["n_position"], use_prompt=use_prompt)
        obs, prompt, masks = get_action_inputs(case, cfg)
        space = spaces.Discrete(6) if disc else None
        memory = m.init_mem(1) if mem else None
        seq = torch.from_numpy(prompt) if prompt is not None else torch.zeros(0, dtype=torch.long)
        with torch.no_grad():
            for st in range(steps):
                seq = torch.from_numpy(obs[st]) if mem else torch.cat([seq, torch.from_numpy(obs[st])])
                act, (seq, _), memory = E.get_action(args, m, seq, None, tok, lfp, 0, ol, al, disc, space, memory, prompt_strategy=strat, action_mask=masks[st])
                out[f"{case}/{st}/act"] = np.asarray(act, dtype=np.float64)
                out[f"{case}/{st}/seq"] = seq.numpy().copy()
                if mem:
                    out[f"{case}/{st}/mem_last"] = memory[-1].numpy().copy()
    return out


SAMPLER_CASES = [  # (total, consumed, micro_batch, rank, world)
    (100, 0, 4, 0, 2), (100, 0, 4, 1, 2), (64, 16, 2, 3, 4), (37, 0, 5, 0, 1), (1000, 256, 8, 5, 8), (96, 192, 4, 1, 2),
]


def gen_samplers():
    from src.data.data_samplers import SequentialPretrainingSampler, RandomPretrainingSampler, my_collate_fn
    from src.data.input_specs import NLPTaskInput, RLTaskInput
    out = {"cases": np.array(SAMPLER_CASES)}
    for i, (tot, cons, mb, rank, world) in enumerate(SAMPLER_CASES):
        if cons < tot:
            s = SequentialPretrainingSampler(tot, cons, mb, rank, world)
            out[f"seq{i}"] = np.array([b for b in s], dtype=np.int64).reshape(-1, mb)
            s2 = SequentialPretrainingSampler(tot, cons, mb, rank, world, drop_last=False)
            bl = [b for b in s2]
            out[f"seq_last{i}"] = np.array(bl[-1], dtype=np.int64)
        for sharding in (True, False):
            r = RandomPretrainingSampler(list(range(tot)), tot, cons, mb, rank, world, sharding)
            out[f"rand{int(sharding)}_{i}"] = np.array([b for b in r], dtype=np.int64).reshape(-1, mb)
            out[f"rand{int(sharding)}_{i}_consumed"] = np.int64(r.consumed_samples)
    # collate: 2 NLP + 1 RL + 1 NLP samples -> [NLP(3 rows), RL(1 row)]
    mk = lambda v: torch.full((1, 6), v, dtype=torch.int64)
    tasks = [NLPTaskInput(position_id=None, attention_mask=None, loss_mask=mk(1).float(), label=mk(10), text_seq=mk(11), text_len=None),
             NLPTaskInput(position_id=None, attention_mask=None, loss_mask=mk(2).float(), label=mk(20), text_seq=mk(21), text_len=None),
             RLTaskInput(position_id=mk(3), attention_mask=None, loss_mask=mk(3).float(), label=mk(30), text_seq=None,
                         vision_seq=torch.full((1, 2, 3, 16, 16), 3.0), tensor_seq=mk(31)),
             NLPTaskInput(position_id=None, attention_mask=None, loss_mask=mk(4).float(), label=mk(40), text_seq=mk(41), text_len=None)]
    merged = my_collate_fn(tasks)
    out["collate_types"] = np.array([type(m).__name__ for m in merged])
    out["collate_nlp_label"] = merged[0].label.numpy()
    out["collate_nlp_text"] = merged[0].text_seq.numpy()
    out["collate_rl_vision_shape"] = np.array(merged[1].vision_seq.shape)
    out["collate_rl_tensor"] = merged[1].tensor_seq.numpy()
    return out


def gen_data_ingest():
    """Golden vectors for libdb1_data.so, produced by the reference itself:
    * a token store written by the reference's MMapIndexedDatasetBuilder (src/data/indexed_dataset.py:566-598) -> data_fixture.idx/.bin
      (data files) plus the items / sizes / doc_idx it reads back;
    * the reference's native index builders (src/data/helpers.cpp compiled by oracle/Makefile into oracle/_ref/)."""
    if not hasattr(np, "float"):
        np.float = np.float64  # the reference's dtype table still uses the alias numpy removed (indexed_dataset.py:109)
    from src.data import indexed_dataset as ref_idx
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref"))
    import helpers as ref_helpers
    rng = np.random.default_rng(21)
    out = {}
    prefix = os.path.join(HERE, "data_fixture")
    b = ref_idx.MMapIndexedDatasetBuilder(prefix + ".bin", dtype=np.uint16)
    lens = [5, 1, 17, 64, 3, 9, 33, 2, 40, 7, 12]
    items = [rng.integers(0, 50000, n).astype(np.uint16) for n in lens]
    for i, it in enumerate(items):
        b.add_item(torch.from_numpy(it.astype(np.int64)))
        if i in (1, 3, 4, 8, 10):
            b.end_document()
    b.finalize(prefix + ".idx")
    ds = ref_idx.MMapIndexedDataset(prefix, skip_warmup=True)
    out["store_sizes"], out["store_doc_idx"] = np.array(ds.sizes), np.array(ds.doc_idx)
    out["store_flat"] = np.concatenate([np.array(ds[i]) for i in range(len(ds))])
    out["store_get_3_10_20"] = np.array(ds.get(3, 10, 20))
    out["store_get_8_5"] = np.array(ds.get(8, 5))
    sl = ds[2:6]
    out["store_slice_2_6_lens"] = np.array([len(x) for x in sl])
    out["store_slice_2_6_flat"] = np.concatenate([np.array(x) for x in sl])
    del ds
    # sample index: documents = the store's items shuffled over 3 epochs, as gpt_dataset.py:262-292 calls it
    sizes = np.array(lens, dtype=np.int32)
    for k, (seq, epochs) in enumerate([(16, 3), (7, 2), (50, 4), (2, 1)]):
        doc_idx = np.concatenate([rng.permutation(len(lens)) for _ in range(epochs)]).astype(np.int32)
        tpe = int(sizes.sum())
        out[f"sample_args{k}"] = np.array([seq, epochs, tpe])
        out[f"sample_doc_idx{k}"] = doc_idx
        out[f"sample_idx{k}"] = np.array(ref_helpers.build_sample_idx(sizes, doc_idx, seq, epochs, tpe))
    out["sample_sizes"] = sizes
    pl = rng.integers(2, 40, 23).astype(np.int32)
    out["rl_path_lengths"] = pl
    for tn in (1, 5, 47):
        out[f"rl_idx_tn{tn}"] = np.array(ref_helpers.build_rl_sample_idx(pl, tn))
    w = np.array([0.5, 0.25, 0.15, 0.1])
    for size in (1, 10, 1000):
        di, dsi = np.zeros(size, np.uint8), np.zeros(size, np.int64)
        ref_helpers.build_blending_indices(di, dsi, w, len(w), size, False)
        out[f"blend_index_{size}"], out[f"blend_sample_{size}"] = di, dsi
    out["blend_weights"] = w
    return out


def gen_init_stats():
    """a14: moments of the reference's own initialisation (transformer_xl.py:444-468 + the nn.Conv2d / nn.GroupNorm defaults it leaves
    in the patch embedder), plain and DeepNorm: per parameter (mean, std, min, max, numel).  Pins the DISTRIBUTION (the values depend on
    torch's RNG stream)."""
    from src.model import TransformerXL
    out = {}
    for tag, extra in (("plain", {}), ("deepnorm", {"use_deepnorm": True})):
        cfg = case_cfg("small_mixed")
        cfg.update(dict(n_embed=256, n_head=4, n_layer=3, text_vocab_size=4000), **extra)
        torch.manual_seed(1234)
        m = TransformerXL(SimpleNamespace(**cfg))
        names = []
        for k, v in m.state_dict().items():
            if k.startswith("ic_encoder.") or k == "pos_emb.inv_freq" or (".dec_attn.r_" in k and not cfg["untie_r"]):
                continue
            v = v.double()
            names.append(k)
            out[f"{tag}/{k}"] = np.array([float(v.mean()), float(v.std(unbiased=False)), float(v.min()), float(v.max()), v.numel()], np.float64)
        out[f"{tag}/names"] = np.array(names)
        if extra:
            out[f"{tag}/alpha_beta"] = np.array([m.deepnorm_alpha, m.deepnorm_beta], np.float64)
    out["cfg_n_embed"], out["cfg_n_head"], out["cfg_n_layer"], out["cfg_text_vocab_size"] = np.int64(256), np.int64(4), np.int64(3), np.int64(4000)
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "init":
        np.savez_compressed(os.path.join(HERE, "init_stats.npz"), **gen_init_stats())
        print("wrote init_stats")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "data":
        np.savez_compressed(os.path.join(HERE, "data_ingest.npz"), **gen_data_ingest())
        print("wrote data_ingest + data_fixture.idx/.bin")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "gpt":
        np.savez_compressed(os.path.join(HERE, "gpt_dataset.npz"), **gen_gpt_dataset())
        print("wrote gpt_dataset")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "get_action":
        np.savez_compressed(os.path.join(HERE, "get_action.npz"), **gen_get_action())
        print("wrote get_action")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "packers":
        np.savez_compressed(os.path.join(HERE, "rl_dataset.npz"), **gen_rl_dataset())
        np.savez_compressed(os.path.join(HERE, "caption_vqa.npz"), **gen_caption_vqa())
        print("wrote rl_dataset + caption_vqa")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "samplers":
        np.savez_compressed(os.path.join(HERE, "samplers.npz"), **gen_samplers())
        print("wrote samplers")
        return
    torch.manual_seed(0)
    only = sys.argv[2:] if len(sys.argv) > 2 and sys.argv[1] == "model" else None
    for i, name in enumerate(CASES):
        if only is not None and name not in only:
            continue
        d = gen_model_case(name, seed=100 + i)
        np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **d)
        print("wrote", name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(d.items())[:4]})
    if only is not None:
        return
    np.savez_compressed(os.path.join(HERE, "patch_embed.npz"), **gen_patch_embed(7))
    np.savez_compressed(os.path.join(HERE, "scalar_tokenizer.npz"), **gen_scalar_tokenizer(11))
    np.savez_compressed(os.path.join(HERE, "adam.npz"), **gen_adam(13))
    np.savez_compressed(os.path.join(HERE, "scheduler.npz"), **gen_scheduler())
    r = gen_rl_packing()
    if r is not None:
        np.savez_compressed(os.path.join(HERE, "rl_packing.npz"), **r)
    print("done")


if __name__ == "__main__":
    main()
