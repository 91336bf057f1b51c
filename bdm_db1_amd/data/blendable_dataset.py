"""Mixture of datasets with the reference's rule (src/data/blendable_dataset.py:30-72): inside every run of ``global_batch_size``
consecutive indices the positions are split between the datasets in proportion to their weights (rounded, in dataset order), and the
sample of the chosen dataset is drawn uniformly with ``numpy.random.randint`` (the global stream, like the reference).  This is what
puts a fixed share of RL / text / caption rows into every global batch of the 870-task mixture."""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch


class BlendableDataset(torch.utils.data.Dataset):
    def __init__(self, datasets: List, weights, global_batch_size: Optional[int] = None):
        super().__init__()
        self.datasets = datasets
        w = torch.tensor(weights)
        assert (w > 0).all()
        w = w / w.sum()
        if global_batch_size is None:
            global_batch_size = len(datasets)
        else:
            assert global_batch_size >= len(datasets)
        self.sample_batch_size = global_batch_size
        per_batch = (global_batch_size * w).round()
        ends = per_batch.cumsum(0).int().numpy()
        self.offset_in_batch = np.zeros_like(ends)      # first position of each dataset inside a batch-sized run of indices
        self.offset_in_batch[1:] = ends[:-1]
        self.size = sum(len(d) for d in datasets)

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        which = int(np.argwhere(self.offset_in_batch <= idx % self.sample_batch_size).max())
        return self.datasets[which][int(np.random.randint(low=0, high=len(self.datasets[which])))]
