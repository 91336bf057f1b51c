"""The caller of the hot path: the reference's training procedure (src/train_utils/train.py) over the MI355X engine."""
from .train import forward_and_backward_step, train, train_step, evaluate_loss  # noqa: F401
