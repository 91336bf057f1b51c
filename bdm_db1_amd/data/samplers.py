"""Host-side batch plumbing in front of the hot path, with the reference's names and behaviour
(src/data/data_samplers.py:28-277): which sample indices each data-parallel rank draws, and how a list of
per-sample task inputs becomes one batched input object per task type for ``TransformerXL.forward``.
Index arithmetic only; pinned against index sequences produced by the reference (tests/golden/samplers.npz).
"""
from __future__ import annotations

import dataclasses
import random
from typing import List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset


def my_collate_fn(task_list: List):
    """One merged object per task TYPE, in order of first appearance; every non-None field is concatenated on dim 0
    (data_samplers.py:28-42 with GatoInputBase.merge_into_one, input_specs.py:57-69)."""
    groups = {}
    for task in task_list:
        groups.setdefault(type(task).__name__, []).append(task)
    merged = []
    for tasks in groups.values():
        head = tasks[0]
        for f in dataclasses.fields(head):
            vals = [getattr(t, f.name) for t in tasks]
            if vals[0] is None:
                continue
            setattr(head, f.name, torch.cat([v for v in vals if v is not None], dim=0))
        merged.append(head)
    return merged


class SequentialPretrainingSampler:
    """Consecutive indices; of every global chunk of micro_batch*world indices, rank r yields rows [r*mb, (r+1)*mb)
    (data_samplers.py:112-170).  The trailing partial chunk is dropped unless drop_last=False."""

    def __init__(self, total_samples, consumed_samples, micro_batch_size, data_parallel_rank, data_parallel_size, drop_last=True):
        assert total_samples > 0, "no sample to consume: {}".format(total_samples)
        assert consumed_samples < total_samples, "no samples left to consume: {}, {}".format(consumed_samples, total_samples)
        assert micro_batch_size > 0 and data_parallel_size > 0
        assert data_parallel_rank < data_parallel_size, \
            "data_parallel_rank should be smaller than data size: {}, {}".format(data_parallel_rank, data_parallel_size)
        self.total_samples, self.consumed_samples = total_samples, consumed_samples
        self.micro_batch_size, self.data_parallel_rank = micro_batch_size, data_parallel_rank
        self.micro_batch_times_data_parallel_size = micro_batch_size * data_parallel_size
        self.drop_last = drop_last

    def __len__(self):
        return self.total_samples

    def get_start_end_idx(self):
        start = self.data_parallel_rank * self.micro_batch_size
        return start, start + self.micro_batch_size

    def __iter__(self):
        chunk = self.micro_batch_times_data_parallel_size
        lo, hi = self.get_start_end_idx()
        first = self.consumed_samples
        n_full = (self.total_samples - first) // chunk
        for c in range(n_full):
            base = first + c * chunk
            yield list(range(base + lo, base + hi))
        rest = list(range(first + n_full * chunk, self.total_samples))
        if rest and not self.drop_last:
            yield rest[lo:hi]


class RandomSeedDataset(Dataset):
    """Re-seeds torch / random / numpy from (index + epoch seed) before every item (data_samplers.py:173-190)."""

    def __init__(self, args, dataset):
        self.base_seed = self.curr_seed = args.seed
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def set_epoch(self, epoch):
        self.curr_seed = self.base_seed + epoch

    def __getitem__(self, idx):
        seed = idx + self.curr_seed
        torch.manual_seed(seed)
        random.seed(seed)
        np.random.seed(seed)
        return self.dataset[idx]


class RandomPretrainingSampler:
    """Per-epoch random permutation (torch.randperm seeded with the epoch number).  With data_sharding every rank permutes
    its own contiguous bucket; otherwise one global permutation is strided over the ranks (data_samplers.py:193-277)."""

    def __init__(self, dataset, total_samples, consumed_samples, micro_batch_size, data_parallel_rank, data_parallel_size, data_sharding):
        assert total_samples > 0, "no sample to consume: {}".format(total_samples)
        assert micro_batch_size > 0 and data_parallel_size > 0
        assert data_parallel_rank < data_parallel_size, \
            "data_parallel_rank should be smaller than data size: {}, {}".format(data_parallel_rank, data_parallel_size)
        self.dataset = dataset
        self.total_samples, self.consumed_samples = total_samples, consumed_samples
        self.micro_batch_size = micro_batch_size
        self.data_parallel_rank, self.data_parallel_size = data_parallel_rank, data_parallel_size
        self.data_sharding = data_sharding
        self.micro_batch_times_data_parallel_size = micro_batch_size * data_parallel_size
        self.last_batch_size = total_samples % self.micro_batch_times_data_parallel_size

    def __len__(self):
        return self.total_samples

    def __iter__(self):
        chunk = self.micro_batch_times_data_parallel_size
        active = self.total_samples - self.last_batch_size
        self.epoch = self.consumed_samples // active
        done = self.consumed_samples % active
        assert done % chunk == 0
        if isinstance(self.dataset, RandomSeedDataset):
            self.dataset.set_epoch(self.epoch)
        g = torch.Generator()
        g.manual_seed(self.epoch)
        if self.data_sharding:
            bucket = (self.total_samples // chunk) * self.micro_batch_size
            order = self.data_parallel_rank * bucket + torch.randperm(bucket, generator=g)[done // self.data_parallel_size:]
        else:
            full = (self.total_samples // self.micro_batch_size) * self.micro_batch_size
            order = torch.randperm(full, generator=g)[done:][self.data_parallel_rank::self.data_parallel_size]
        order = order.tolist()
        for b in range(len(order) // self.micro_batch_size):  # an incomplete last batch is dropped
            self.consumed_samples += chunk
            yield order[b * self.micro_batch_size:(b + 1) * self.micro_batch_size]


def build_pretraining_data_loader(args, dataset, consumed_samples, total_samples: Optional[int], eval=False):
    """torch DataLoader over one of the samplers above with ``my_collate_fn`` (data_samplers.py:57-109)."""
    from .. import mpu
    if dataset is None:
        return None
    if total_samples is None:
        total_samples = len(dataset)
    rank, world = mpu.get_data_parallel_rank(), mpu.get_data_parallel_world_size()
    if args.dataloader_type == "single":
        sampler = SequentialPretrainingSampler(total_samples, consumed_samples, args.micro_batch_size, rank, world)
    elif args.dataloader_type == "cyclic":
        if eval:
            sampler = SequentialPretrainingSampler(total_samples, consumed_samples, args.micro_batch_size * eval, 0, 1)
        else:
            sampler = RandomPretrainingSampler(dataset, total_samples, consumed_samples, args.micro_batch_size, rank, world, True)
    else:
        raise Exception("{} dataloader type is not supported.".format(args.dataloader_type))
    # Worker processes: datasets that tokenize on the device (RLFullDataset -> ContinuousScalarTokenizer.discretize, the HIP mu-law kernel)
    # cannot run in FORKED workers (HIP does not survive a fork: "Cannot re-initialize CUDA in forked subprocess"), so workers are started
    # with the spawn context -- each initialises its own HIP runtime; num_workers = 0 (the reference's default, src/config.py) stays in-process.
    nw = int(getattr(args, "num_workers", 0) or 0)
    ctx = None
    if nw > 0 and (getattr(dataset, "uses_device_tokenizer", False) or any(getattr(d, "uses_device_tokenizer", False) for d in getattr(dataset, "datasets", []))):
        import multiprocessing
        ctx = multiprocessing.get_context("spawn")
    return torch.utils.data.DataLoader(dataset, batch_sampler=sampler, num_workers=nw, pin_memory=True, collate_fn=my_collate_fn,
                                       multiprocessing_context=ctx)
