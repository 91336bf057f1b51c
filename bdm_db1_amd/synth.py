"""Synthetic batches of the shapes BASELINE.json's configs name.  Only the RAW data is synthetic
(random token ids, random images, random actions); the samples are built by the package's own sample builders and batched by its
collate function, i.e. by the path a real data loader takes:
  * text:     ids uniform [0, 32000), label = ids shifted by one                                  (gpt_dataset.py:86-180)
  * RL:       synthetic image-observation trajectories -> data.rl_dataset.RLFullDataset.get        (rl_dataset.py:590-752)
  * caption:  synthetic (prompt, image, caption) records -> data.coco_token_dataset.ICDataset      (coco_token_dataset.py:104-152)
and the data-parallel sharding rule of SequentialPretrainingSampler (data_samplers.py:152-170)."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from .data import ICTaskInput, NLPTaskInput, RLTaskInput, my_collate_fn
from .data.coco_token_dataset import ICDataset
from .data.rl_dataset import RLFullDataset


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
    """Atari-like trajectories: one (h/16 * w/16)-patch image per observation, one discrete action, packed by
    RLFullDataset: [patch placeholders (-1), SEP, action]* truncated to L + 1 and split into input / label, position ids 1..npatch+1
    on observation + separator and 0 on the action, loss where the label is an action."""
    rng = np.random.default_rng(seed)
    p = cfg.vision_patch_size
    step = (img_hw[0] // p) * (img_hw[1] // p) + 2
    ntr = (L + step - 1) // step  # transition_num (rl_dataset.py:229-231): each synthetic trajectory is exactly one sample long
    trajs = [(rng.random((ntr, cfg.vision_num_input_channels, img_hw[0], img_hw[1]), dtype=np.float32) * 255.0,
              rng.integers(0, n_actions, ntr).astype(np.int64)) for _ in range(B)]
    tokenizers = (SimpleNamespace(vocab_size=cfg.text_vocab_size), SimpleNamespace(num_continuous_bin=cfg.num_continuous_bin))
    ds = RLFullDataset(trajs, L, tokenizers, overlap_with_text=cfg.overlap_with_text, num_discrete_values=cfg.num_discrete_values,
                       vision_patch_size=p, use_prompt=False)
    assert ds.transition_num == ntr
    first = [i for i, row in enumerate(ds.indices) if row[1] == 0]   # the sample that starts at step 0 of each trajectory
    (batch,) = my_collate_fn([ds[i] for i in first])
    batch.loss_mask = batch.loss_mask.to(torch.float32)
    return batch.to(device=device)


def caption_batch(B: int, L: int, seed: int, device, cfg, img_hw=(224, 224), prompt_len: int = 8) -> ICTaskInput:
    """prompt + image patches + caption filling L tokens; loss on the last image position and on the caption (ICDataset)"""
    rng = np.random.default_rng(seed)
    p = cfg.vision_patch_size
    nv = (img_hw[0] // p) * (img_hw[1] // p)
    Tt = L - prompt_len - nv
    assert Tt > 0
    V = cfg.text_vocab_size
    mean = np.array([0.485, 0.456, 0.406], np.float32)[:, None, None]
    std = np.array([0.229, 0.224, 0.225], np.float32)[:, None, None]
    C = cfg.vision_num_input_channels
    records = []
    for b in range(B):
        img = (rng.random((C, img_hw[0], img_hw[1]), dtype=np.float32) - mean[:C]) / std[:C]   # vit_dataset.py:42-43
        caption = np.concatenate([rng.integers(1, V, Tt), [0]]).astype(np.int32)                # Tt tokens + eos
        records.append(dict(text=caption, img=torch.from_numpy(img), prompt=rng.integers(0, V, prompt_len).tolist(), img_id=b))
    ds = ICDataset(SimpleNamespace(n_position=L), records, SimpleNamespace(eos_token_id=0))
    (batch,) = my_collate_fn([ds[i] for i in range(B)])
    batch.label, batch.prompt_seq, batch.text_seq = batch.label.long(), batch.prompt_seq.long(), batch.text_seq.long()
    return batch.to(device=device)


def mixture_batch(B: int, L: int, seed: int, device, cfg):
    """config 5 stand-in: 50 % RL, 25 % text, 25 % caption rows (BlendableDataset assigns slots by position)."""
    n_rl = max(1, B // 2)
    n_txt = max(1, (B - n_rl) // 2)
    n_ic = max(1, B - n_rl - n_txt)
    return [rl_batch(n_rl, L, seed, device, cfg), text_batch(n_txt, L, seed + 1, device, cfg.text_vocab_size),
            caption_batch(n_ic, L, seed + 2, device, cfg)]
