"""Small integer helpers with the names of the reference's src/mpu/utils.py:19-71."""
import torch


def ensure_divisibility(numerator, denominator):
    assert numerator % denominator == 0, "{} is not divisible by {}".format(numerator, denominator)


def divide(numerator, denominator):
    ensure_divisibility(numerator, denominator)
    return numerator // denominator


def split_tensor_along_last_dim(tensor, num_partitions, contiguous_split_chunks=False):
    size = divide(tensor.size()[tensor.dim() - 1], num_partitions)
    chunks = torch.split(tensor, size, dim=tensor.dim() - 1)
    return tuple(c.contiguous() for c in chunks) if contiguous_split_chunks else chunks


class VocabUtility:
    """[first, last) vocabulary range owned by ``rank`` out of ``world_size`` partitions."""

    @staticmethod
    def vocab_range_from_per_partition_vocab_size(per_partition_vocab_size, rank, world_size):
        first = rank * per_partition_vocab_size
        return first, first + per_partition_vocab_size

    @staticmethod
    def vocab_range_from_global_vocab_size(global_vocab_size, rank, world_size):
        return VocabUtility.vocab_range_from_per_partition_vocab_size(divide(global_vocab_size, world_size), rank, world_size)
