"""Autoregressive action decoding as the reference's RL evaluation does it (src/evaluation/evaluate_rl.py:96-266), with its function
names, arguments and return values, so that its episode loop (``evalute_one_episode`` :286-410, which needs the gym / d4rl simulators
and is therefore not rebuilt) can call into this package unchanged.  One environment step = ``action_length`` model calls:

  * without memory the whole token window is re-fed every call (grown by the predicted token, cut back by whole transitions when it
    exceeds ``n_position``; a fixed prompt in front of the window is kept);
  * with memory (``model.init_mem``) the first call feeds the new transition's observation tokens, each following call ONE token --
    the case the K/V-cached decode path of ``bdm_db1_amd.TransformerXL`` exists for -- and a final call pushes the last action token
    into the memory.

Logits are restricted to the action vocabulary before the argmax (continuous bins after the text ids, or the first
``action_space.n`` discrete ids, minus an optional environment action mask) and predicted ids are mapped back to tokenizer bins."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..data.input_specs import RLTaskInput
from ..data.packers import _get_action_flag_and_position_id

_OUT = 1e10   # what the reference subtracts from logits that may not be chosen


def masked_logits_for_action(args, logits, discrete_action: bool, action_space, env_action_mask: Optional[np.ndarray] = None):
    """evaluate_rl.py:96-124 (in place on ``logits`` [B, L, V])"""
    text = args.text_vocab_size
    if not discrete_action:   # continuous bins sit after the text ids (and after the discrete ids when those do not overlap the text ids)
        logits[..., :text if args.overlap_with_text else text + args.num_discrete_values] -= _OUT
        logits[..., -1] -= _OUT                  # the separator token
        return logits
    if args.overlap_with_text:
        logits[..., action_space.n:] -= _OUT
    else:
        logits[..., :text] -= _OUT
        logits[..., text + action_space.n:] -= _OUT
    if env_action_mask is not None:              # 1 = allowed
        penalty = torch.from_numpy((np.abs(env_action_mask - 1) * _OUT).reshape(1, -1)).to(logits.device)
        logits[:, -1, :action_space.n] = logits[:, -1, :action_space.n] - penalty
    return logits


def recover_model_predict_token_to_tokenizer_raw(args, preds, discrete_action: bool):
    """token id -> tokenizer bin / discrete value, in place (evaluate_rl.py:127-138)"""
    if args.overlap_with_text:
        if discrete_action:
            assert (preds < args.num_discrete_values).all()
            return preds
        assert (preds >= args.text_vocab_size).all(), preds
        preds -= args.text_vocab_size
        return preds
    preds -= args.text_vocab_size
    if not discrete_action:
        preds -= args.num_discrete_values
    return preds


def truncate_sequence_by_stepsize(current_seq, vision_seq, obs_length, act_length, max_length=None):
    """drop the oldest transition (evaluate_rl.py:141-146)"""
    step = obs_length + act_length + 1
    return current_seq[step:], (vision_seq[1:] if vision_seq is not None else None)


def truncate_memory(mems, obs_len, act_len):
    step = obs_len + act_len + 1
    return [m[:, step:] for m in mems]


def _model_call(model, tokens, vision_seq, position_id, memory):
    x = RLTaskInput(tensor_seq=tokens, vision_seq=vision_seq, text_seq=None, attention_mask=None, loss_mask=None, label=None,
                    position_id=torch.tensor(position_id, dtype=torch.long))
    x.to(device=model.device)
    x.apply(lambda t: t[None, ...])
    return model([x], compute_loss=False, mems=memory)


def get_action(args, model, current_seq, vision_seq, cont_tokenizer, len_fixed_prompt, len_fixed_prompt_img, obs_length, action_length,
               discrete_action: bool, action_space, model_memory, prompt_strategy: str = "fixed_prompt", action_mask: Optional[np.ndarray] = None):
    """-> (action, (current_seq, vision_seq), model_memory)   (evaluate_rl.py:157-266)"""
    trans = action_length + obs_length + 1
    picked = []
    for i_act in range(action_length):
        if i_act == 0 or model_memory is None:
            _, pos_id = _get_action_flag_and_position_id(0, len(current_seq) - 1, obs_length, action_length, 0)
        else:
            pos_id = np.array([0])               # a lone action token
        res = _model_call(model, current_seq, vision_seq, pos_id, model_memory)
        if model_memory is not None:
            model_memory = res[-1]
        logits = masked_logits_for_action(args, res[0], discrete_action, action_space, env_action_mask=action_mask)
        preds = logits[:, -1, :].argmax(-1)
        picked_host = preds.cpu()            # (synchronises: the call's kernels are done)
        chk = getattr(model, "check_decode_chain", None)
        if chk is not None:                  # a persistent one-token launch that could not hand off raises here, before the action is used
            chk()
        if model_memory is None:
            current_seq = torch.cat([current_seq, picked_host], dim=0)
            if len(current_seq) > args.n_position:
                if args.use_prompt and prompt_strategy == "fixed_prompt":   # the window behind the fixed prompt slides by one transition
                    current_seq[len_fixed_prompt:] = torch.roll(current_seq[len_fixed_prompt:], -trans).clone()
                    current_seq = current_seq[:-trans]
                    if vision_seq is not None:
                        # the reference rolls WITHOUT dims (evaluate_rl.py:217-219): the image window is flattened and moved by ONE ELEMENT,
                        # not by one image; reproduced as is (results identical to the reference's on the same inputs)
                        vision_seq[len_fixed_prompt_img:] = torch.roll(vision_seq[len_fixed_prompt_img:], -1).clone()
                        vision_seq = vision_seq[:-1]
                else:
                    current_seq, vision_seq = truncate_sequence_by_stepsize(current_seq, vision_seq, obs_length, action_length, None)
        else:
            assert prompt_strategy != "fixed_prompt"    # the memory slides: a fixed prompt cannot stay in front of it
            current_seq, vision_seq = picked_host.clone(), None
        picked.append(recover_model_predict_token_to_tokenizer_raw(args, preds, discrete_action).cpu())
    if model_memory is not None:                 # the last action token enters the memory too
        model_memory = _model_call(model, current_seq, None, [0], model_memory)[-1]
    if discrete_action:
        return picked[0].item(), (current_seq, vision_seq), model_memory
    return cont_tokenizer.decode(torch.cat(picked), is_action=True).numpy(), (current_seq, vision_seq), model_memory


def get_action_batched(args, model, obs_tokens, cont_tokenizer, obs_length, action_length, discrete_action: bool, action_space, model_memory,
                       action_masks: Optional[np.ndarray] = None):
    """The memory mode of ``get_action`` for M environments of one kind at once: ONE model call per token for all of them.

    The reference evaluates its environments one by one at batch 1 (evaluate_rl.py:452-482 hands a rank up to ~110 of them), so every new
    token streams the 2.4 GB of decoder weights once PER ENVIRONMENT; the weights do not care how many rows they multiply, so M
    environments in one call cost one stream.  Same arithmetic per row as ``get_action`` (evaluate_rl.py:157-266: observation call, one token
    per call, the memorising call; logits restricted to the action vocabulary, argmax, ids mapped back to tokenizer bins) -- pinned to the
    reference's golden episode replayed in every row (tests/test_model_gpu.py).

    ``obs_tokens`` [M, q]: the new transition's observation tokens (+ separator) of every environment (token observations: the image-patch
    form goes through ``get_action``); ``model_memory``: ``model.init_mem(M)`` or a ``RingMemory(model, M)``; ``action_masks`` [M, n] (1 =
    allowed) or None.  -> (actions [M] or [M, action_length], last tokens [M, 1], model_memory)"""
    seq = torch.as_tensor(obs_tokens)
    M = seq.shape[0]
    picked = []
    for i_act in range(action_length):
        if i_act == 0:
            _, pos_id = _get_action_flag_and_position_id(0, seq.shape[1] - 1, obs_length, action_length, 0)
        else:
            pos_id = np.array([0])
        res = _model_call_batched(model, seq, pos_id, model_memory)
        model_memory = res[-1]
        logits = masked_logits_for_action(args, res[0], discrete_action, action_space, env_action_mask=None)
        if discrete_action and action_masks is not None:
            penalty = torch.from_numpy(np.abs(np.asarray(action_masks) - 1) * _OUT).to(logits.device)
            logits[:, -1, :action_space.n] = logits[:, -1, :action_space.n] - penalty
        preds = logits[:, -1, :].argmax(-1)
        host = preds.cpu()                       # (synchronises)
        chk = getattr(model, "check_decode_chain", None)
        if chk is not None:
            chk()
        seq = host[:, None].clone()
        picked.append(recover_model_predict_token_to_tokenizer_raw(args, preds, discrete_action).cpu())
    model_memory = _model_call_batched(model, seq, [0], model_memory)[-1]      # the last action token enters the memory too
    if discrete_action:
        return picked[0].numpy(), seq, model_memory
    acts = cont_tokenizer.decode(torch.stack(picked, dim=1).reshape(-1), is_action=True).reshape(M, action_length)
    return acts.numpy(), seq, model_memory


def _model_call_batched(model, tokens, position_id, memory):
    M = tokens.shape[0]
    pos = torch.tensor(np.asarray(position_id), dtype=torch.long)[None, :].expand(M, -1).contiguous()
    x = RLTaskInput(tensor_seq=tokens, vision_seq=None, text_seq=None, attention_mask=None, loss_mask=None, label=None, position_id=pos)
    x.to(device=model.device)
    return model([x], compute_loss=False, mems=memory)
