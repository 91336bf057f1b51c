"""Text samples from the memory-mapped token store, with the reference's names (src/data/gpt_dataset.py:86-448): a sample = the next
``seq_length + 1`` tokens of the epoch-shuffled, concatenated document stream (documents run into each other, the last token of a sample
is the first of the next), served in a second shuffled order.  The three index arrays are those of the reference --

  doc_idx      the documents of all epochs, shuffled (the last epoch apart when it contributes < 80 % of an epoch's samples)  (:355-369)
  sample_idx   (index into doc_idx, offset) of every sample boundary: ``db1_build_sample_idx`` in libdb1_data.so = helpers.cpp:117-203
  shuffle_idx  the order samples are served in                                                                                  (:424-448)

-- drawn from ``numpy.random.RandomState(seed)`` in the reference's order, so a given (store, seed) yields the reference's samples
(tests/golden/gpt_dataset.npz).  The reference builds them on rank 0, saves ``.npy`` files next to the data and synchronises the ranks
with a hand-made all-reduce barrier (:204-325); here every rank builds them in memory (milliseconds through the C++ builder);
``cache_prefix`` keeps the reference's on-disk cache format for corpora where that matters."""
from __future__ import annotations

import os

import numpy as np
import torch

from .input_specs import NLPTaskInput


def get_ltor_masks_and_position_ids(data, eod_token_id, reset_position_ids, reset_attention_mask, eod_mask_loss):
    """(attention_mask [L, L] bool: True = hidden, loss_mask float32 [L], position_ids int64 [L])  (gpt_dataset.py:29-83; the two reset
    flags are unused there as well)"""
    n = data.shape[0]
    attention_mask = np.tril(np.ones((n, n))) < 0.5
    loss_mask = np.ones(data.shape, dtype=np.float32)
    if eod_mask_loss:
        loss_mask[data == eod_token_id] = 0.0
    return attention_mask, loss_mask, np.arange(n, dtype=np.int64)


def _num_tokens(documents, sizes):
    return np.sum(sizes[documents])


def _num_epochs(tokens_per_epoch, seq_length, num_samples):
    """smallest number of epochs that yields num_samples samples of seq_length + 1 tokens overlapping by one"""
    epochs, total = 0, 0
    while True:
        epochs += 1
        total += tokens_per_epoch
        if (total - 1) // seq_length >= num_samples:
            return epochs


def _build_doc_idx(documents, num_epochs, np_rng, separate_last_epoch):
    if not separate_last_epoch or num_epochs == 1:
        doc_idx = np.tile(np.asarray(documents), num_epochs).astype(np.int32)
        np_rng.shuffle(doc_idx)
        return doc_idx
    return np.concatenate((_build_doc_idx(documents, num_epochs - 1, np_rng, False), _build_doc_idx(documents, 1, np_rng, False)))


def _build_shuffle_idx(num_samples, total_size, np_rng):
    dtype_ = np.int64 if total_size >= (np.iinfo(np.uint32).max - 1) else np.uint32
    first = np.arange(0, num_samples, 1, dtype=dtype_)
    np_rng.shuffle(first)
    if num_samples == total_size:
        return first
    last = np.arange(num_samples, total_size, 1, dtype=dtype_)
    np_rng.shuffle(last)
    return np.concatenate((first, last))


def _build_index_mappings(name, data_prefix, documents, sizes, num_samples, seq_length, seed, cache_prefix=None):
    """-> (doc_idx, sample_idx, shuffle_idx)"""
    from .indexed import build_sample_idx
    tokens_per_epoch = _num_tokens(documents, sizes)
    num_epochs = _num_epochs(tokens_per_epoch, seq_length, num_samples)
    files = None
    if cache_prefix is not None:   # the reference's file names (:204-212)
        stem = f"{cache_prefix}_{name}_indexmap_{num_samples}ns_{seq_length}sl_{seed}s"
        files = [stem + s for s in ("_doc_idx.npy", "_sample_idx.npy", "_shuffle_idx.npy")]
        if all(os.path.isfile(f) for f in files):
            return tuple(np.load(f, allow_pickle=True, mmap_mode="r") for f in files)
    np_rng = np.random.RandomState(seed=seed)
    if num_epochs == 1:
        separate_last_epoch = False
    else:
        before_last = ((num_epochs - 1) * tokens_per_epoch - 1) // seq_length
        last_epoch_samples = num_samples - before_last
        per_epoch = (tokens_per_epoch - 1) // seq_length
        assert 0 <= last_epoch_samples < per_epoch + 1, "last epoch number of samples out of range"
        separate_last_epoch = last_epoch_samples < int(0.80 * per_epoch)
    doc_idx = _build_doc_idx(documents, num_epochs, np_rng, separate_last_epoch)
    sample_idx = build_sample_idx(np.ascontiguousarray(sizes, dtype=np.int32), doc_idx, seq_length, num_epochs, tokens_per_epoch)
    n_first = before_last if separate_last_epoch else sample_idx.shape[0] - 1
    shuffle_idx = _build_shuffle_idx(n_first, sample_idx.shape[0] - 1, np_rng)
    if files is not None:
        for f, arr in zip(files, (doc_idx, sample_idx, shuffle_idx)):
            np.save(f, arr, allow_pickle=True)
    return doc_idx, sample_idx, shuffle_idx


class GPTDataset(torch.utils.data.Dataset):
    """same positional signature as the reference's (also used for RLDataset): name, data_prefix, documents, indexed_dataset,
    num_samples (unused), seq_length, seed"""

    def __init__(self, name, data_prefix, documents, indexed_dataset, unused_num_samples, seq_length, seed, eos_token_id=None,
                 reset_position_ids: bool = False, reset_attention_mask: bool = False, eod_mask_loss: bool = False, cache_prefix=None):
        self.name, self.indexed_dataset, self.seq_length = name, indexed_dataset, seq_length
        self.eos_token_id, self.eod_mask_loss = eos_token_id, eod_mask_loss
        self.reset_position_ids, self.reset_attention_mask = reset_position_ids, reset_attention_mask
        documents = np.asarray(documents)
        sizes = np.asarray(indexed_dataset.sizes)
        assert np.min(documents) >= 0 and np.max(documents) < sizes.shape[0]
        num_samples = np.sum(sizes[documents]) // seq_length
        self.doc_idx, self.sample_idx, self.shuffle_idx = _build_index_mappings(name, data_prefix, documents, sizes, num_samples, seq_length, seed,
                                                                                cache_prefix=cache_prefix)

    def __len__(self):
        return self.sample_idx.shape[0] - 1   # sample i = [sample_idx[i], sample_idx[i + 1])

    def __getitem__(self, idx):
        idx = self.shuffle_idx[idx]
        (doc_f, off_f), (doc_l, off_l) = self.sample_idx[idx], self.sample_idx[idx + 1]
        ds = self.indexed_dataset
        if doc_f == doc_l:
            sample = ds.get(self.doc_idx[doc_f], offset=off_f, length=off_l - off_f + 1)
        else:   # the rest of the first document, whole documents in between, the head of the last one
            parts = [ds.get(self.doc_idx[doc_f], offset=off_f)]
            parts += [ds.get(self.doc_idx[i]) for i in range(doc_f + 1, doc_l)]
            parts.append(ds.get(self.doc_idx[doc_l], length=off_l + 1))
            sample = np.concatenate(parts)
        tokens = sample[:self.seq_length]
        _, loss_mask, position_ids = get_ltor_masks_and_position_ids(tokens, self.eos_token_id, self.reset_position_ids, self.reset_attention_mask,
                                                                     self.eod_mask_loss)
        res = NLPTaskInput(position_id=position_ids, attention_mask=None, loss_mask=loss_mask, label=sample[1:self.seq_length + 1], text_seq=tokens,
                           text_len=None)
        res.apply(lambda x: x.astype(np.int32) if x.dtype == np.uint16 else x)
        res.apply(lambda x: torch.tensor(x))
        res.apply(lambda x: x[None, ...])
        return res
