from .input_specs import GatoInputBase, RLTaskInput, NLPTaskInput, ICTaskInput, VQATaskInput  # noqa: F401
from .samplers import (my_collate_fn, SequentialPretrainingSampler, RandomPretrainingSampler, RandomSeedDataset,  # noqa: F401
                       build_pretraining_data_loader)
from .packers import _get_action_flag_and_position_id, _truncate_or_pad_to_match_seq_len  # noqa: F401


def __getattr__(name):  # the token store / index builders / sample builders need libdb1_data.so: imported on first use
    if name in ("RLFullDataset", "RLDataset"):
        import importlib
        return getattr(importlib.import_module(".rl_dataset", __name__), name)
    if name in ("GPTDataset",):
        import importlib
        return getattr(importlib.import_module(".gpt_dataset", __name__), name)
    if name in ("BlendableDataset",):
        import importlib
        return getattr(importlib.import_module(".blendable_dataset", __name__), name)
    if name in ("ICDataset", "VQADataset", "get_ltor_masks_and_position_ids", "get_loss_mask_vqa", "fit_caption_length"):
        import importlib
        return getattr(importlib.import_module(".coco_token_dataset", __name__), name)
    if name in ("MMapIndexedDataset", "build_sample_idx", "build_rl_sample_idx", "build_blending_indices", "indexed"):
        import importlib
        mod = importlib.import_module(".indexed", __name__)
        return mod if name == "indexed" else getattr(mod, name)
    raise AttributeError(name)
