"""Input contract of TransformerXL.forward: same class and field names as the reference's
src/data/input_specs.py:23-112 (the model dispatches on the CLASS NAME, so the reference's own
dataclasses can be passed in unchanged).  Only the container behaviour the hot path needs is
provided (.to / .apply); dataset-side helpers stay with the reference's loader."""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any, Optional


@dataclass
class GatoInputBase:
    position_id: Optional[Any]
    attention_mask: Optional[Any]
    loss_mask: Optional[Any]
    label: Optional[Any]

    def to(self, **kwargs):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None and hasattr(v, "to"):
                setattr(self, f.name, v.to(**kwargs))
        return self

    def apply(self, fn, *args, **kwargs):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                setattr(self, f.name, fn(v, *args, **kwargs))


@dataclass
class RLTaskInput(GatoInputBase):
    text_seq: Any = None
    vision_seq: Any = None
    tensor_seq: Any = None


@dataclass
class NLPTaskInput(GatoInputBase):
    text_seq: Any = None
    text_len: Any = None


@dataclass
class ICTaskInput(GatoInputBase):
    prompt_seq: Any = None
    img_seq: Any = None
    text_seq: Any = None
    img_id_seq: Any = None


@dataclass
class VQATaskInput(GatoInputBase):
    prompt_seq: Any = None
    img_seq: Any = None
    text_seq: Any = None
    img_id_seq: Any = None
    ques_id_seq: Any = None
    ques_len: Any = None
