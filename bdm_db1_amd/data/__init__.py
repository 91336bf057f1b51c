from .input_specs import GatoInputBase, RLTaskInput, NLPTaskInput, ICTaskInput, VQATaskInput  # noqa: F401
from .samplers import (my_collate_fn, SequentialPretrainingSampler, RandomPretrainingSampler, RandomSeedDataset,  # noqa: F401
                       build_pretraining_data_loader)
