"""Communicator bookkeeping: every public name of the reference's src/mpu/initialize.py:49-398.

The reference copies Megatron's tensor/pipeline/data group builder but only ever runs it with
tensor = pipeline = 1 (evaluate_rl.py:493; README.md:129 "we only use distributed data parallel").
Here the data-parallel group is the one real communicator: it IS the default (world) process group, so no
second RCCL communicator over the same 8 GPUs is created (the reference's builder makes world + 4 of them).
Model-parallel groups are per-rank singleton groups, created LAZILY the first time a caller asks for one
(DeepSpeed-style ``mpu=`` consumers get a valid group of size 1; a training run that never asks pays for none).
Asking for tensor or pipeline parallel sizes > 1 raises: the reference has no such implementation to reproduce.
"""
import torch

_DATA_PARALLEL_GROUP = None
_MODEL_PARALLEL_GROUP = None
_TENSOR_MODEL_PARALLEL_GROUP = None
_PIPELINE_MODEL_PARALLEL_GROUP = None
_EMBEDDING_GROUP = None
_VIRTUAL_PIPELINE_MODEL_PARALLEL_RANK = None
_VIRTUAL_PIPELINE_MODEL_PARALLEL_WORLD_SIZE = None
_MPU_TENSOR_MODEL_PARALLEL_WORLD_SIZE = None
_MPU_PIPELINE_MODEL_PARALLEL_WORLD_SIZE = None
_MPU_TENSOR_MODEL_PARALLEL_RANK = None
_MPU_PIPELINE_MODEL_PARALLEL_RANK = None
_PIPELINE_GLOBAL_RANKS = None
_INITIALIZED = False


def _dist_on():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def is_unitialized():
    """(sic) the reference spells it this way (initialize.py:49)."""
    return _DATA_PARALLEL_GROUP is None and not _INITIALIZED


def initialize_model_parallel(tensor_model_parallel_size_=1, pipeline_model_parallel_size_=1,
                              virtual_pipeline_model_parallel_size_=None, pipeline_model_parallel_split_rank_=None):
    global _DATA_PARALLEL_GROUP, _MODEL_PARALLEL_GROUP, _TENSOR_MODEL_PARALLEL_GROUP, _PIPELINE_MODEL_PARALLEL_GROUP
    global _EMBEDDING_GROUP, _PIPELINE_GLOBAL_RANKS, _INITIALIZED
    global _VIRTUAL_PIPELINE_MODEL_PARALLEL_RANK, _VIRTUAL_PIPELINE_MODEL_PARALLEL_WORLD_SIZE
    if tensor_model_parallel_size_ != 1 or pipeline_model_parallel_size_ != 1:
        raise NotImplementedError("DB1 is data-parallel only (tensor = pipeline = 1); got "
                                  f"tp={tensor_model_parallel_size_} pp={pipeline_model_parallel_size_}")
    assert _DATA_PARALLEL_GROUP is None and not _INITIALIZED, "data parallel group is already initialized"
    _INITIALIZED = True
    if virtual_pipeline_model_parallel_size_ is not None:
        _VIRTUAL_PIPELINE_MODEL_PARALLEL_RANK = 0
        _VIRTUAL_PIPELINE_MODEL_PARALLEL_WORLD_SIZE = virtual_pipeline_model_parallel_size_
    if not _dist_on():
        _PIPELINE_GLOBAL_RANKS = [0]
        return
    _DATA_PARALLEL_GROUP = torch.distributed.group.WORLD   # all ranks: the default communicator, nothing new is created
    _PIPELINE_GLOBAL_RANKS = [torch.distributed.get_rank()]


def _singleton_group():
    """this rank's size-1 group for the tensor / pipeline / embedding getters: created on first use, by this rank alone
    (``use_local_synchronization``: only the members of a new group take part in its creation)"""
    global _MODEL_PARALLEL_GROUP, _TENSOR_MODEL_PARALLEL_GROUP, _PIPELINE_MODEL_PARALLEL_GROUP, _EMBEDDING_GROUP
    if _MODEL_PARALLEL_GROUP is None and _dist_on():
        g = torch.distributed.new_group([torch.distributed.get_rank()], use_local_synchronization=True)
        _MODEL_PARALLEL_GROUP = _TENSOR_MODEL_PARALLEL_GROUP = _PIPELINE_MODEL_PARALLEL_GROUP = _EMBEDDING_GROUP = g
    return _MODEL_PARALLEL_GROUP


def model_parallel_is_initialized():
    return _INITIALIZED


def _need():
    assert _INITIALIZED, "model parallel groups are not initialized (call initialize_model_parallel())"


def get_model_parallel_group():
    _need()
    return _singleton_group()


def get_tensor_model_parallel_group():
    _need()
    return _singleton_group()


def get_pipeline_model_parallel_group():
    _need()
    return _singleton_group()


def get_data_parallel_group():
    _need()
    return _DATA_PARALLEL_GROUP


def get_embedding_group():
    _need()
    return _singleton_group()


def set_tensor_model_parallel_world_size(world_size):
    global _MPU_TENSOR_MODEL_PARALLEL_WORLD_SIZE
    _MPU_TENSOR_MODEL_PARALLEL_WORLD_SIZE = world_size


def set_pipeline_model_parallel_world_size(world_size):
    global _MPU_PIPELINE_MODEL_PARALLEL_WORLD_SIZE
    _MPU_PIPELINE_MODEL_PARALLEL_WORLD_SIZE = world_size


def get_tensor_model_parallel_world_size():
    return _MPU_TENSOR_MODEL_PARALLEL_WORLD_SIZE if _MPU_TENSOR_MODEL_PARALLEL_WORLD_SIZE is not None else 1


def get_model_parallel_world_size():
    assert get_pipeline_model_parallel_world_size() == 1, "legacy get_model_parallel_world_size is only supported if PP is disabled"
    return get_tensor_model_parallel_world_size()


def get_pipeline_model_parallel_world_size():
    return _MPU_PIPELINE_MODEL_PARALLEL_WORLD_SIZE if _MPU_PIPELINE_MODEL_PARALLEL_WORLD_SIZE is not None else 1


def set_tensor_model_parallel_rank(rank):
    global _MPU_TENSOR_MODEL_PARALLEL_RANK
    _MPU_TENSOR_MODEL_PARALLEL_RANK = rank


def set_pipeline_model_parallel_rank(rank):
    global _MPU_PIPELINE_MODEL_PARALLEL_RANK
    _MPU_PIPELINE_MODEL_PARALLEL_RANK = rank


def get_tensor_model_parallel_rank():
    return _MPU_TENSOR_MODEL_PARALLEL_RANK if _MPU_TENSOR_MODEL_PARALLEL_RANK is not None else 0


def get_model_parallel_rank():
    assert get_pipeline_model_parallel_world_size() == 1, "legacy get_model_parallel_rank is only supported if PP is disabled"
    return get_tensor_model_parallel_rank()


def get_pipeline_model_parallel_rank():
    return _MPU_PIPELINE_MODEL_PARALLEL_RANK if _MPU_PIPELINE_MODEL_PARALLEL_RANK is not None else 0


def is_pipeline_first_stage(ignore_virtual=False):
    if not ignore_virtual and get_virtual_pipeline_model_parallel_world_size() is not None \
            and get_virtual_pipeline_model_parallel_rank() != 0:
        return False
    return get_pipeline_model_parallel_rank() == 0


def is_pipeline_last_stage(ignore_virtual=False):
    if not ignore_virtual:
        vws = get_virtual_pipeline_model_parallel_world_size()
        if vws is not None and get_virtual_pipeline_model_parallel_rank() != (vws - 1):
            return False
    return get_pipeline_model_parallel_rank() == (get_pipeline_model_parallel_world_size() - 1)


def get_virtual_pipeline_model_parallel_rank():
    return _VIRTUAL_PIPELINE_MODEL_PARALLEL_RANK


def set_virtual_pipeline_model_parallel_rank(rank):
    global _VIRTUAL_PIPELINE_MODEL_PARALLEL_RANK
    _VIRTUAL_PIPELINE_MODEL_PARALLEL_RANK = rank


def get_virtual_pipeline_model_parallel_world_size():
    return _VIRTUAL_PIPELINE_MODEL_PARALLEL_WORLD_SIZE


def _my_rank():
    return torch.distributed.get_rank() if _dist_on() else 0


def get_tensor_model_parallel_src_rank():
    return (_my_rank() // get_tensor_model_parallel_world_size()) * get_tensor_model_parallel_world_size()


def get_pipeline_model_parallel_first_rank():
    _need()
    return _PIPELINE_GLOBAL_RANKS[0]


def get_pipeline_model_parallel_last_rank():
    _need()
    return _PIPELINE_GLOBAL_RANKS[get_pipeline_model_parallel_world_size() - 1]


def get_pipeline_model_parallel_next_rank():
    _need()
    return _PIPELINE_GLOBAL_RANKS[(get_pipeline_model_parallel_rank() + 1) % get_pipeline_model_parallel_world_size()]


def get_pipeline_model_parallel_prev_rank():
    _need()
    return _PIPELINE_GLOBAL_RANKS[(get_pipeline_model_parallel_rank() - 1) % get_pipeline_model_parallel_world_size()]


def get_data_parallel_world_size():
    return torch.distributed.get_world_size(group=get_data_parallel_group()) if _dist_on() and _DATA_PARALLEL_GROUP is not None else 1


def get_data_parallel_rank():
    return torch.distributed.get_rank(group=get_data_parallel_group()) if _dist_on() and _DATA_PARALLEL_GROUP is not None else 0


def destroy_model_parallel():
    global _DATA_PARALLEL_GROUP, _MODEL_PARALLEL_GROUP, _TENSOR_MODEL_PARALLEL_GROUP, _PIPELINE_MODEL_PARALLEL_GROUP
    global _EMBEDDING_GROUP, _PIPELINE_GLOBAL_RANKS, _INITIALIZED
    _DATA_PARALLEL_GROUP = _MODEL_PARALLEL_GROUP = _TENSOR_MODEL_PARALLEL_GROUP = None
    _PIPELINE_MODEL_PARALLEL_GROUP = _EMBEDDING_GROUP = _PIPELINE_GLOBAL_RANKS = None
    _INITIALIZED = False
