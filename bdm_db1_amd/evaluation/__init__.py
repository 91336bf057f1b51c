"""The inference-side caller of the hot path: action decoding with Transformer-XL memory (src/evaluation/evaluate_rl.py:96-266)."""
from .evaluate_rl import (get_action, get_action_batched, masked_logits_for_action, recover_model_predict_token_to_tokenizer_raw,  # noqa: F401
                          truncate_memory, truncate_sequence_by_stepsize)
