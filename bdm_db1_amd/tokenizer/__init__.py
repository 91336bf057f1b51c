"""Host mirror of the reference's ``src/tokenizer`` pieces that sit on the DB1 hot path."""
from .scalar_tokenizer import ContinuousScalarTokenizer

__all__ = ["ContinuousScalarTokenizer"]
