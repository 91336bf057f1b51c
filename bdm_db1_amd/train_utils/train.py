"""Training procedure with the reference's names, arguments and return values (src/train_utils/train.py:32-243), over the DB1 engine
(``bdm_db1_amd.initialize``) instead of DeepSpeed's.  What is kept: the iteration loop and its bookkeeping on ``args``
(``iteration``, ``train_iters``, ``eval_interval``, ``eval_iters``, ``save_dir``, ``save_interval``), one optimizer step =
``engine.gradient_accumulation_steps()`` micro-steps of ``get_batch_fn -> engine(batch) -> engine.backward(loss) -> engine.step()``
returning the list of micro-step losses, the optional TensorBoard-style ``sm_writer.add_scalar("Train loss", ...)``, checkpointing
every ``save_interval`` iterations.  What is not: the reference's validation pass also rolls out RL simulators and decodes captions /
answers (``evaluate_and_print_results`` :86-207, needs gym / d4rl / COCO tooling); ``evaluate_loss`` is its first part, the
validation LOSS over ``eval_iters`` batches."""
from __future__ import annotations

from typing import Any, Callable, List, Optional

import torch

from ..checkpointing import save_checkpoint


def forward_and_backward_step(args, model, data_iterator, get_batch_fn: Callable, do_backward: bool = True, return_all: bool = False):
    """train.py:210-243.  (WARNING kept from the reference: always three return values.)"""
    loss_list, logits_list, input_data_list = [], [], []
    for _ in range(model.gradient_accumulation_steps()):
        input_data = get_batch_fn(args, data_iterator)
        logits, loss = model(input_data)
        if do_backward:
            model.backward(loss)
            model.step()
        if return_all:
            input_data_list.append(input_data)
            logits_list.append(logits)
        loss_list.append(loss)
    if return_all:
        return loss_list, logits_list, input_data_list
    return loss_list, None, None


def train_step(args, model, data_iterator, get_batch_fn: Callable) -> List[torch.Tensor]:
    """train.py:78-83"""
    model.train()
    losses, _, _ = forward_and_backward_step(args, model, data_iterator, get_batch_fn, do_backward=True)
    return losses


def evaluate_loss(args, model, data_iterator, get_batch_fn: Callable) -> float:
    """mean validation loss over ``args.eval_iters`` optimizer-step-sized groups of batches, no gradients (train.py:97-120: the sum over
    iterations of np.mean(loss_list), divided by eval_iters).  ``data_iterator`` may be the reference's (iterator, dict) pair.  The RL / IC /
    VQA roll-outs that evaluate_and_print_results also starts (simulators, COCO evaluators) are out of scope."""
    if isinstance(data_iterator, tuple):   # the reference hands (iterator, {name: iterator}) to evaluate_and_print_results (train.py:95): take the iterator
        data_iterator = data_iterator[0]
    model.eval()
    total, n = 0.0, 0
    with torch.no_grad():
        for _ in range(int(args.eval_iters)):
            losses, _, _ = forward_and_backward_step(args, model, data_iterator, get_batch_fn, do_backward=False)
            total += float(sum(float(x) for x in losses) / len(losses))
            n += 1
    model.train()
    return total / max(n, 1)


def train(args, model, train_data_iterator, valid_data_iterator: Optional[Any], get_batch_fn: Callable, sm_writer: Any = None):
    """train.py:32-75: ``args.iteration`` .. ``args.train_iters`` optimizer steps"""
    iteration = args.iteration
    while iteration < args.train_iters:
        losses = train_step(args, model, train_data_iterator, get_batch_fn)
        if sm_writer:
            loss = sum(losses) / len(losses)
            sm_writer.add_scalar("Train loss", loss.item(), iteration)
        args.iteration = iteration
        if valid_data_iterator is not None and ((getattr(args, "eval_interval", None) and iteration % args.eval_interval == 0)
                                                or iteration == args.train_iters - 1):
            val = evaluate_loss(args, model, valid_data_iterator, get_batch_fn)
            if sm_writer:
                sm_writer.add_scalar("Valid loss", val, iteration)
        iteration += 1
        if getattr(args, "save_dir", None) and iteration % args.save_interval == 0:
            save_checkpoint(args, iteration, model)
    return iteration
