"""Image-caption / VQA sample builders with the reference's names (src/data/coco_token_dataset.py:58-210) over in-memory samples:
the wrapped ``dataset`` yields dicts (``img`` float tensor [3, H, W], token id arrays, ``prompt`` ids) -- where the reference gets
them from torchvision's COCO / VQAv2 readers (:24-55, vqa_dataset.py) is the caller's business.

Sequence layout (the model concatenates ``[prompt, image patches, text]`` to ``n_position`` tokens, transformer_xl.py:674-748):
  * caption: input text = caption[:-1]; label = the caption right-aligned so that the LAST IMAGE position predicts its first token;
    loss on that position and on every text position whose INPUT token is not eos  (:58-83, 104-152);
  * VQA: input text = (question ++ answer)[:-1]; label = the answer right-aligned; loss on the position before the answer and on
    the answer positions whose previous answer token is not eos  (:86-101, 155-210).
"""
from __future__ import annotations

import numpy as np
import torch

from .input_specs import ICTaskInput, VQATaskInput


def fit_caption_length(text, seq_length: int) -> torch.Tensor:
    """token ids truncated, or right-padded with 0, to ``seq_length`` (what ``RandomCOCO.__getitem__`` does to the caption it drew,
    coco_token_dataset.py:43-48; ``seq_length`` there = the text budget minus the prompt length)"""
    text = torch.as_tensor(np.asarray(text), dtype=torch.int32).reshape(-1)
    if text.shape[-1] >= seq_length:
        return text[..., :seq_length]
    return torch.nn.functional.pad(text, (0, seq_length - text.shape[-1]), "constant", 0)


def get_ltor_masks_and_position_ids(data, eod_token_id, full_seq_length):
    """(attention_mask=None, loss_mask float32 [full], position_ids int32 [full]) for the text ``data`` that ends the sequence"""
    n = data.shape[0]
    loss_mask = np.zeros((full_seq_length,), dtype=np.float32)
    loss_mask[full_seq_length - n:] = (data != eod_token_id)
    loss_mask[full_seq_length - n - 1] = 1          # the last image position predicts the first caption token
    position_ids = np.zeros((full_seq_length,), dtype=np.int32)
    position_ids[full_seq_length - n:] = np.arange(n, dtype=np.int32)
    return None, loss_mask, position_ids


def get_loss_mask_vqa(label, eod_token_id, eod_mask_loss, full_seq_length):
    """float32 [full]: 1 on the position before the answer and on answer positions whose previous answer token is not eos"""
    n = len(label) if isinstance(label, list) else label.shape[0]
    # (a Python LIST of ids is never eos-masked by the reference: `list == int` is a plain False there, :95-96; arrays are)
    keep = np.ones((n,), dtype=np.float32) if isinstance(label, list) else (np.asarray(label) != eod_token_id).astype(np.float32)
    loss_mask = np.zeros((full_seq_length,), dtype=np.float32)
    if n > 1:
        loss_mask[full_seq_length - n + 1:] = keep[:-1]
    loss_mask[full_seq_length - n] = 1
    return loss_mask


def _finish(res):
    res.apply(lambda x: torch.tensor(x) if not isinstance(x, torch.Tensor) else x)
    res.apply(lambda x: x[None, ...])
    return res


class ICDataset:
    """``args.n_position`` is the total sequence length; ``tokenizer.eos_token_id`` marks padding in the caption"""

    def __init__(self, args, dataset, tokenizer) -> None:
        self.dataset, self.args = dataset, args
        ICDataset.tokenizer = tokenizer

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index: int):
        L = self.args.n_position
        data = self.dataset[index]
        caption = np.array(data["text"], dtype=np.int32)
        tokens = caption[:-1]
        _, loss_mask, _ = get_ltor_masks_and_position_ids(tokens, ICDataset.tokenizer.eos_token_id, full_seq_length=L)
        labels = np.zeros((L,), dtype=np.int32)
        labels[L - tokens.shape[0] - 1:] = caption
        return _finish(ICTaskInput(position_id=None, attention_mask=None, loss_mask=loss_mask, label=labels,
                                   prompt_seq=np.array(data["prompt"], dtype=np.int32), img_seq=data["img"].to(dtype=torch.half),
                                   text_seq=tokens, img_id_seq=data["img_id"]))


class VQADataset:
    def __init__(self, args, dataset, tokenizer) -> None:
        self.dataset, self.args = dataset, args
        VQADataset.tokenizer = tokenizer

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index: int):
        L = self.args.n_position
        data = self.dataset[index]
        ans = data["ans"]
        tokens = np.concatenate([data["ques"], ans], axis=-1)[:-1]
        labels = np.zeros((L,), dtype=np.int32)
        labels[L - len(ans):] = ans
        loss_mask = get_loss_mask_vqa(ans, VQADataset.tokenizer.eos_token_id, getattr(self.args, "eod_mask_loss", False), full_seq_length=L)
        return _finish(VQATaskInput(position_id=None, attention_mask=None, loss_mask=loss_mask, prompt_seq=data["prompt"],
                                    img_seq=data["img"].to(dtype=torch.half), text_seq=tokens, label=labels, img_id_seq=data["img_id"],
                                    ques_id_seq=data["ques_id"], ques_len=data["ques_len"]))
