"""Process-group bookkeeping with the reference's ``src.mpu`` surface (src/mpu/__init__.py:19-32,
src/mpu/initialize.py:49-398, src/mpu/utils.py:19-71).  DB1 is data-parallel only
(README.md:129; TP = PP = 1 at every call site), so the tensor/pipeline getters are world-size-1
answers and only the data-parallel communicator (RCCL over xGMI via torch.distributed "nccl") is live."""
import torch

from .initialize import *  # noqa: F401,F403
from .utils import divide, ensure_divisibility, split_tensor_along_last_dim, VocabUtility  # noqa: F401


def print_rank_0(message):
    """If distributed is initialized, print only on rank 0."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        if torch.distributed.get_rank() == 0:
            print(message, flush=True)
    else:
        print(message, flush=True)


def print_with_rank(message):
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        print(f"rank: {torch.distributed.get_rank()}", message, flush=True)
    else:
        print(message, flush=True)
