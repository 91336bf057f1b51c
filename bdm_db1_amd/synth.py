"""Synthetic batches of the shapes BASELINE.json's configs name.  Host-side index
building only; the layouts restate the reference's packers:
  * text:     ids uniform [0, 32000), label = ids shifted by one                       (gpt_dataset.py:86-180)
  * RL:       [obs patches (-1 placeholders), SEP, action]* truncated to L+1, split     (rl_dataset.py:44-71, 614-752)
  * caption:  prompt + image patches + text; loss on the last image position and text   (coco_token_dataset.py:58-152)
and the data-parallel sharding rule of SequentialPretrainingSampler (data_samplers.py:152-170)."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from .data import ICTaskInput, NLPTaskInput, RLTaskInput


def db1_config(size: str = "1.3B", **over) -> SimpleNamespace:
    """The released geometry (scripts/evaluate/evaluate_rl_1.2B.sh:16-19,70-83) or the tiny plumbing config."""
    base = dict(n_inner=None, pre_lnorm=False, same_length=True, untie_r=False, text_vocab_size=32000, num_discrete_values=1024,
                num_continuous_bin=1024, overlap_with_text=True, embd_pdrop=0.0, drop=0.0, dropattn=0.0, activation_fn="geglu",
                layer_norm_epsilon=1e-5, share_input_output_embedding=True, use_deepnorm=False, fp16=True, vision_patch_size=16,
                vision_num_input_channels=3, vision_position_vocab_size=128, vision_hidden_dropout_prob=0.0)
    if size == "1.3B":
        base.update(n_embed=2048, n_layer=24, n_head=16, n_position=1024, mem_len=1024)
    elif size == "tiny":
        base.update(n_embed=128, n_layer=2, n_head=4, n_position=256, mem_len=256, fp16=False)
    else:
        raise ValueError(size)
    base.update(over)
    return SimpleNamespace(**base)


def dp_shard(global_rows: int, micro_batch: int, rank: int, world: int):
    """Row range of this rank inside each global chunk of micro_batch*world rows (data_samplers.py:152-155)."""
    assert global_rows % (micro_batch * world) == 0
    return [(c * micro_batch * world + rank * micro_batch, c * micro_batch * world + (rank + 1) * micro_batch)
            for c in range(global_rows // (micro_batch * world))]


def text_batch(B: int, L: int, seed: int, device, vocab: int = 32000) -> NLPTaskInput:
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, vocab, size=(B, L + 1))
    T = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
    return NLPTaskInput(position_id=None, attention_mask=None, loss_mask=T(np.ones((B, L), np.float32)),
                        label=T(ids[:, 1:]), text_seq=T(ids[:, :-1]), text_len=None)


def rl_batch(B: int, L: int, seed: int, device, cfg, img_hw=(64, 80), n_actions: int = 18) -> RLTaskInput:
    """Atari-like transitions: (h/16 * w/16) image patches, SEP, one discrete action."""
    rng = np.random.default_rng(seed)
    p = cfg.vision_patch_size
    npatch = (img_hw[0] // p) * (img_hw[1] // p)
    step = npatch + 2
    sep = cfg.text_vocab_size + cfg.num_continuous_bin + (0 if cfg.overlap_with_text else cfg.num_discrete_values)
    ntr = (L + step - 1) // step  # transition_num (rl_dataset.py:229-231)
    seq = np.empty((B, ntr * step), np.int64)
    for t in range(ntr):
        seq[:, t * step:t * step + npatch] = -1
        seq[:, t * step + npatch] = sep
        seq[:, t * step + npatch + 1] = rng.integers(0, n_actions, B)
    seq = np.concatenate([seq, np.full((B, 1), -1, np.int64)], axis=1)[:, :L + 1]
    inp, lab = seq[:, :-1], seq[:, 1:]
    within = np.arange(L) % step
    pos = np.where(within <= npatch, within + 1, 0).astype(np.int64)          # obs + separator: 1..npatch+1, action: 0
    loss_mask = (within == npatch).astype(np.float32)                          # the label at the separator is the action
    nimg = int(np.ceil((inp[0] == -1).sum() / npatch))
    vision = rng.random((B, nimg, cfg.vision_num_input_channels, img_hw[0], img_hw[1]), dtype=np.float32) * 255.0
    T = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
    return RLTaskInput(position_id=T(np.tile(pos, (B, 1))), attention_mask=None, loss_mask=T(np.tile(loss_mask, (B, 1))),
                       label=T(lab), text_seq=None, vision_seq=T(vision), tensor_seq=T(inp))


def caption_batch(B: int, L: int, seed: int, device, cfg, img_hw=(224, 224), prompt_len: int = 8) -> ICTaskInput:
    rng = np.random.default_rng(seed)
    p = cfg.vision_patch_size
    nv = (img_hw[0] // p) * (img_hw[1] // p)
    Tt = L - prompt_len - nv
    assert Tt > 0
    V = cfg.text_vocab_size
    prompt = rng.integers(0, V, (B, prompt_len))
    text = rng.integers(1, V, (B, Tt))
    img = rng.random((B, cfg.vision_num_input_channels, img_hw[0], img_hw[1]), dtype=np.float32)
    mean = np.array([0.485, 0.456, 0.406], np.float32)[None, :, None, None]
    std = np.array([0.229, 0.224, 0.225], np.float32)[None, :, None, None]
    img = (img - mean[:, :img.shape[1]]) / std[:, :img.shape[1]]
    label = np.zeros((B, L), np.int64)
    label[:, prompt_len + nv - 1:L - 1] = text          # next-token targets start at the last image position
    label[:, L - 1] = 0                                  # eos
    mask = np.zeros((B, L), np.float32)
    mask[:, prompt_len + nv - 1:] = 1.0
    T = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
    return ICTaskInput(position_id=None, attention_mask=None, loss_mask=T(mask), label=T(label), prompt_seq=T(prompt),
                       img_seq=T(img), text_seq=T(text), img_id_seq=None)


def mixture_batch(B: int, L: int, seed: int, device, cfg):
    """config 5 stand-in: 50 % RL, 25 % text, 25 % caption rows (BlendableDataset assigns slots by position)."""
    n_rl = max(1, B // 2)
    n_txt = max(1, (B - n_rl) // 2)
    n_ic = max(1, B - n_rl - n_txt)
    return [rl_batch(n_rl, L, seed, device, cfg), text_batch(n_txt, L, seed + 1, device, cfg.text_vocab_size),
            caption_batch(n_ic, L, seed + 2, device, cfg)]
