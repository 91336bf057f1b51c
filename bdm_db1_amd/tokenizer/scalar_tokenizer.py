"""``ContinuousScalarTokenizer`` with the reference's surface (src/tokenizer/scalar_tokenizer.py:20-63: constructor arguments,
``discretize(x, is_action)`` -> int32 tensor, ``decode(x, is_action)`` -> float32 tensor) on the gfx950 kernels
``db1_mulaw_discretize`` / ``db1_mulaw_decode`` (include/db1_hip.h).  Callers: the RL dataset (rl_dataset.py:427-432,459-464), the
evaluation wrapper (evaluation/rl/wrapper.py:140) and the action read-out (evaluate_rl.py:263).

Token ids are bit-identical to the reference's torch-CPU result (float32 op order, correctly rounded log); decoded values are
within 1 ulp.  Inputs may be NumPy arrays, CPU tensors (the reference's case: the result comes back on the CPU) or device tensors
(the result stays on the device: nothing synchronises).  There is no CPU implementation here: without libdb1_hip.so and an
MI355X this raises.  In DataLoader workers the HIP runtime has to be initialised by the worker itself: forked workers raise with
that explanation, ``build_pretraining_data_loader`` starts workers with the spawn context for datasets that use this tokenizer."""
from __future__ import annotations

import numpy as np
import torch

from .. import lib, ops


class ContinuousScalarTokenizer:
    def __init__(self, num_continuous_bin: int = 1024, mu: float = 100.0, M: float = 256.0):
        self.num_continuous_bin = num_continuous_bin
        self.mu = mu
        self.M = M

    @staticmethod
    def _device():
        import os
        if getattr(torch.cuda, "_is_in_bad_fork", lambda: False)():
            raise lib.Db1Error("ContinuousScalarTokenizer was called in a FORKED DataLoader worker (pid %d): the HIP runtime does not survive a fork.  "
                               "Use bdm_db1_amd.data.build_pretraining_data_loader (it starts workers with the spawn context for datasets that "
                               "tokenize on the device), pass multiprocessing_context='spawn', or num_workers=0." % os.getpid())
        if not torch.cuda.is_available():
            raise lib.Db1Error("ContinuousScalarTokenizer runs on the MI355X kernels (libdb1_hip.so); there is no CPU path")
        return torch.device("cuda", torch.cuda.current_device())

    def discretize(self, x, is_action: bool):
        """scalar_tokenizer.py:28-45"""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x.copy()).float()
        home = x.device
        xd = x.to(device=self._device() if not x.is_cuda else x.device, dtype=torch.float32).contiguous()
        ids = torch.empty(xd.shape, dtype=torch.int32, device=xd.device)
        if xd.numel():
            ops.mulaw_discretize(xd, ids, is_action, self.num_continuous_bin, self.mu, self.M)
        return ids.to(home)

    def decode(self, x, is_action: bool):
        """scalar_tokenizer.py:47-63 (out-of-range ids are clipped; the reference's warning is printed when the input lives on the
        host, where reading the flag costs nothing extra)"""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        home = x.device
        if x.dtype not in (torch.int32, torch.int64):
            x = x.to(torch.int64)  # the reference calls .float() on whatever integer-valued tensor it is handed
        xd = x.to(self._device() if not x.is_cuda else x.device).contiguous()
        out = torch.empty(xd.shape, dtype=torch.float32, device=xd.device)
        flag = torch.zeros(1, dtype=torch.int32, device=xd.device)
        if xd.numel():
            ops.mulaw_decode(xd, out, is_action, self.num_continuous_bin, self.mu, self.M, oob_flag=flag)
        if home.type == "cpu" and xd.numel() and int(flag.item()):
            print("Warning of exceeded range of discrete number to recontruct, by default values will be cliped, "
                  "min: {}, max:{}".format(x.min(), x.max()))
        return out.to(home)
