"""Sequence-packing conventions of the RL trajectory path, with the reference's names (src/data/rl_dataset.py:44-71, 865-872;
used by evaluate_rl.py:176-178 and the RL dataset).  One transition is [observation tokens (obs_seq_len), separator, action
tokens (act_seq_len)].  Index arithmetic only; pinned against arrays produced by the reference (tests/golden/rl_packing.npz)."""
from __future__ import annotations

import numpy as np


def _get_action_flag_and_position_id(index_l, index_r, obs_seq_len, act_seq_len, prepend_trans_num):
    """(action_flag, position_id), int64 [index_r - index_l + 1]; the window must start on a transition boundary.
    position_id = 1 .. obs_seq_len + 1 over the observation tokens and the separator, 0 on action tokens (prompt transitions are
    not distinguished); action_flag = 1 on the action tokens of the non-prompt transitions."""
    n = index_r - index_l + 1
    step = obs_seq_len + act_seq_len + 1
    within = np.arange(n) % step
    position_id = np.where(within <= obs_seq_len, within + 1, 0).astype(np.int64)
    action_flag = ((within > obs_seq_len) & (np.arange(n) >= prepend_trans_num * step)).astype(np.int64)
    return action_flag, position_id


def _truncate_or_pad_to_match_seq_len(arr: np.ndarray, seq_len: int):
    """first seq_len entries, zero-padded at the end when shorter"""
    if len(arr) > seq_len:
        return arr[:seq_len]
    if len(arr) < seq_len:
        return np.pad(arr, (0, seq_len - len(arr)))
    return arr
