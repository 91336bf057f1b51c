"""RL trajectory sample builder: the sequence packer of the reference's ``RLFullDataset`` (src/data/rl_dataset.py:133-862) as a
component over IN-MEMORY trajectories.  What the reference reads from a d4rl environment and an on-disk cache
(``gym.make``, ``qlearning_dataset_with_timeouts``, ``cache_data``; :160-330) is the caller's job here: it hands over
``[(observations, actions), ...]`` per trajectory; everything from there to the ``RLTaskInput`` a training batch is made of is
built here with the reference's conventions:

  * a sample = ``transition_num = (seq_len + obs_dim + act_dim) // (obs_dim + act_dim + 1)`` consecutive transitions starting at
    every step of every trajectory (index table from ``db1_build_rl_sample_idx`` in libdb1_data.so = helpers.cpp:82-115);
  * a transition = ``[observation tokens, SEP, action tokens]``; text / image / tensor observation parts in that order, dict
    observations by sorted key; image patches are ``-1`` placeholders filled by the model from ``vision_seq``  (:614-672);
  * float values -> mu-law bin + ``text_vocab`` (+ ``num_discrete_values`` unless the discrete ids overlap the text ids),
    discrete values as they are (+ ``text_vocab`` unless overlapping)  (:393-473);
  * optional prompt: with probability ``prompt_prob`` a piece (the end, a random sub-sequence or random time steps) of another
    trajectory is put in front and the sample's own transitions are cut to make room; prompt actions carry no loss  (:475-578);
  * ``position_id`` = 1..obs_dim+1 over observation + separator, 0 on actions; ``loss_mask`` = 1 where the LABEL is an action token;
    everything truncated / zero-padded to ``seq_len + 1`` and split into input / label  (:44-71, 673-745).

Random draws use ``numpy.random``'s global stream in the reference's order (one ``choice`` for the prompt trajectory, ``random`` <
prompt_prob, ``random`` < prompt_at_final_transition_prob, the strategy's ``choice``, the offset ``choice``), so a seeded run
reproduces the reference's samples (tests/golden/rl_dataset.npz).  Pass ``rng=np.random.RandomState(s)`` for a private stream.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .input_specs import RLTaskInput
from .packers import _get_action_flag_and_position_id, _truncate_or_pad_to_match_seq_len

Obs = Union[np.ndarray, Dict[str, np.ndarray]]


def _map(fn: Callable, *structs):
    """apply ``fn`` leaf-wise over parallel structures: an array, or a dict of arrays visited in sorted key order (what the
    reference gets from dm-tree's ``map_structure`` for the two observation forms it supports, rl_dataset.py:396-399)"""
    s0 = structs[0]
    if isinstance(s0, dict):
        return {k: _map(fn, *[s[k] for s in structs]) for k in sorted(s0)}
    return fn(*structs)


def _leaves(struct) -> List[Any]:
    if isinstance(struct, dict):
        return [x for k in sorted(struct) for x in _leaves(struct[k])]
    return [struct]


class RLFullDataset(torch.utils.data.Dataset):
    """``RLFullDataset`` over in-memory trajectories.  ``tokenizers = (text_tokenizer, discretizer)`` as in the reference:
    ``text_tokenizer.vocab_size`` (and ``text_tokenizer(list_of_str, padding=, truncation=, max_length=)["input_ids"]`` for text
    observations), ``discretizer.discretize(x, is_action) -> int tensor`` (``bdm_db1_amd.tokenizer.ContinuousScalarTokenizer``).
    ``env`` may carry the reference's optional hooks ``post_process_fn`` / ``action_mapper`` / ``build_task_input``."""

    def __init__(self, trajectories: Sequence[Tuple[Obs, np.ndarray]], seq_length: int, tokenizers: Sequence,
                 overlap_with_text: bool = True, num_discrete_values: int = 1024, prompt_ratio: float = 0.5,
                 prompt_prob: float = 0.25, prompt_at_final_transition_prob: float = 0.5, mask_prompt_action_loss: bool = True,
                 vision_patch_size: int = 16, use_prompt: bool = True, prompt_strategy: str = "stochastic_subseq",
                 env: Any = None, rng: Any = None, name: str = "in_memory"):
        assert prompt_strategy in ("stochastic_subseq", "stochastic_timestep"), prompt_strategy
        self.name = name
        self.env = env
        self.rng = np.random if rng is None else rng
        self.output_sequence_length = int(seq_length)
        self.prompt_strategy, self.use_prompt = prompt_strategy, bool(use_prompt)
        self.vision_patch_size = int(vision_patch_size)
        self.prompt_prob, self.prompt_at_final_transition_prob = prompt_prob, prompt_at_final_transition_prob
        self.prompt_ratio, self.mask_prompt_action_loss = prompt_ratio, mask_prompt_action_loss
        self.text_tokenizer, self.discretizer = tokenizers
        # DataLoader workers of this dataset need their own HIP runtime when the discretizer is the device tokenizer (samplers.py)
        self.uses_device_tokenizer = type(self.discretizer).__module__.startswith("bdm_db1_amd.")
        self.num_discrete_values, self.overlap_with_text = int(num_discrete_values), bool(overlap_with_text)
        self.observations = [t[0] for t in trajectories]
        self.actions = [np.asarray(t[1]) for t in trajectories]
        self.path_lengths = np.array([len(a) for a in self.actions], dtype=np.int32)
        assert len(self.actions) > 0 and all(len(x) == n for o, n in zip(self.observations, self.path_lengths) for x in _leaves(o))
        obs0, act0 = self.get_obs_action_by_path_idx(0)
        self.obs_type_spec = self.get_obs_type_spec(obs0)
        self.observation_dims_for_spec = self.get_observation_dim(obs0)
        self.observation_dim = int(sum(_leaves(self.observation_dims_for_spec)))
        self.action_dim = int(self.get_action_dim(act0[0]))
        step = self.observation_dim + self.action_dim + 1
        self.transition_num = (self.output_sequence_length + step - 1) // step   # ceil((L + 1) / step), rl_dataset.py:229-231
        self.prompt_transition_num = int(prompt_ratio * self.transition_num)
        self.predicted_transition_num = self.transition_num - self.prompt_transition_num
        from .indexed import build_rl_sample_idx   # libdb1_data.so (include/db1_data.h)
        self.indices = build_rl_sample_idx(self.path_lengths, self.transition_num)

    # ---------------------------------------------------------------- structure of one environment (rl_dataset.py:754-810)
    def _post(self, obs):
        return self.env.post_process_fn(obs) if hasattr(self.env, "post_process_fn") else obs

    def get_obs_type_spec(self, obs):
        def kind(x):
            if x.ndim == 4:
                assert x.shape[1] == 3, "We assume the rgb input should of shape (3, h, w)"
                return "image"
            for tag, name in (("float", "float"), ("str", "text"), ("int", "discrete")):
                if tag in x.dtype.name:
                    return name
            raise ValueError(x.dtype)
        return _map(kind, self._post(obs))

    def get_observation_dim(self, obs):
        def dim(x):
            if "str" in x.dtype.name:
                return max(len(t) for t in self.text_tokenizer(x.tolist())["input_ids"])
            if x.ndim == 4 and x.shape[1] == 3:
                return (x.shape[2] // self.vision_patch_size) * (x.shape[3] // self.vision_patch_size)
            return x[0].size
        return _map(dim, self._post(obs))

    def get_action_dim(self, act):
        if hasattr(self.env, "action_mapper"):
            act = self.env.action_mapper(act)
        return act.shape[0] if len(act.shape) == 1 else 1

    def get_obs_action_by_path_idx(self, path_ind: int, start_ind: Optional[int] = None, end_ind: Optional[int] = None):
        start_ind = 0 if start_ind is None else start_ind
        end_ind = end_ind or len(self.actions[path_ind])
        return _map(lambda x: x[start_ind:end_ind], self.observations[path_ind]), self.actions[path_ind][start_ind:end_ind]

    # ---------------------------------------------------------------- token ids (rl_dataset.py:393-473)
    def _continuous_offset(self) -> int:
        return self.text_tokenizer.vocab_size + (0 if self.overlap_with_text else self.num_discrete_values)

    def postprocess_obs_and_act(self, obs_array: Obs, act_array: np.ndarray):
        """-> ((text ids, image, tensor ids), action ids); each observation part ``None`` when the environment has none"""
        obs_array = self._post(obs_array)
        if hasattr(self.env, "action_mapper"):
            act_array = self.env.action_mapper(act_array)

        def one(x, kind, dim):
            text = image = tensor = None
            if kind == "text":
                text = np.array(self.text_tokenizer(x.tolist(), padding="max_length", truncation=True, max_length=dim)["input_ids"], dtype=np.int32)
            elif kind == "image":
                image = x
            elif kind == "float":
                tensor = self.discretizer.discretize(x, is_action=False).numpy() + self._continuous_offset()
            elif kind == "discrete":
                assert x.min() >= 0 and x.max() < self.num_discrete_values
                tensor = x if self.overlap_with_text else x + self.text_tokenizer.vocab_size
            if tensor is not None and tensor.ndim < 2:
                tensor = tensor[:, None]
            return text, image, tensor

        parts = _map(one, obs_array, self.obs_type_spec, self.observation_dims_for_spec)
        if isinstance(parts, dict):
            o_text, o_image, o_tensor = ({k: v[j] for k, v in parts.items()} for j in range(3))
        else:
            o_text, o_image, o_tensor = parts
        if "float" in act_array.dtype.name:
            act_ids = self.discretizer.discretize(act_array, is_action=True).numpy() + self._continuous_offset()
        else:
            assert act_array.min() >= 0 and act_array.max() < self.num_discrete_values
            act_ids = act_array[:, None] if act_array.ndim == 1 else act_array
            if not self.overlap_with_text:
                act_ids = act_ids + self.text_tokenizer.vocab_size
        return (o_text, o_image, o_tensor), act_ids

    # ---------------------------------------------------------------- prompt (rl_dataset.py:475-578)
    def prepend_prompt(self, path_idx: int, observations: Obs, actions: np.ndarray):
        """-> (observations, actions, number of prompt transitions in front)"""
        assert all(len(x) <= self.transition_num for x in _leaves(observations))
        rng, k = self.rng, self.prompt_transition_num
        if not (path_idx >= 0 and rng.random() < self.prompt_prob):
            return observations, actions, 0
        obs_traj, act_traj = self.get_obs_action_by_path_idx(path_idx)
        path_length = self.path_lengths[path_idx]
        if rng.random() < self.prompt_at_final_transition_prob:       # the goal: how the other episode ends
            pick = lambda x: x[-k:]
        elif self.prompt_strategy == "stochastic_timestep":           # k random time steps, in order
            sel = rng.choice(path_length, k, replace=False)
            sel.sort()
            pick = lambda x: x[sel]
        else:                                                         # a random k-step piece
            start = rng.choice(max(path_length - k, 1))
            pick = lambda x: x[start:start + k]
        p_obs, p_act = _map(pick, obs_traj), pick(act_traj)
        n_prompt = len(p_act)
        room = max(0, len(actions) - self.predicted_transition_num)   # the sample's own part is cut to predicted_transition_num
        off = rng.choice(room) if room > 0 else room
        keep = lambda x: x[off:off + self.predicted_transition_num]

        def join(front, back):
            out = np.zeros((front.shape[0] + back.shape[0],) + back.shape[1:], dtype=back.dtype)
            out[:front.shape[0]] = front
            out[front.shape[0]:] = back
            return out

        return _map(join, p_obs, _map(keep, observations)), join(p_act, keep(actions)), n_prompt

    # ---------------------------------------------------------------- one sample (rl_dataset.py:614-752)
    def __len__(self):
        return len(self.indices)

    def __getitem__(self, idx):
        return self.get(idx, with_raw=False)

    def get(self, idx: int, with_raw: bool = False):
        if idx >= len(self.indices):
            idx = idx % len(self.indices)
        path_ind, start_ind, end_ind = (int(v) for v in self.indices[idx])
        path_length = int(self.path_lengths[path_ind])
        observations, actions = self.get_obs_action_by_path_idx(path_ind, start_ind, end_ind)
        n_prompt = 0
        if self.use_prompt:
            other = self.rng.choice(len(self.path_lengths))
            observations, actions, n_prompt = self.prepend_prompt(other, observations, actions)
        (o_text, o_image, o_tensor), act_ids = self.postprocess_obs_and_act(observations, actions)

        present = lambda part: [] if part is None else [v for v in (_leaves(part)) if v is not None]
        images = present(o_image)
        assert len(images) <= 1, "Currently We only support one image in observation"
        o_image = images[0] if images else None
        cols = list(present(o_text))
        if o_image is not None:
            n, c, h, w = o_image.shape
            cols.append(-np.ones((n, (h // self.vision_patch_size) * (w // self.vision_patch_size))))
            if n < self.transition_num:      # every sample of a batch carries transition_num images (zeros behind the real ones)
                padded = np.zeros((self.transition_num, c, h, w), dtype=np.float32)
                padded[:n] = o_image
                o_image = padded
        cols += present(o_tensor)
        sep = self.discretizer.num_continuous_bin + self._continuous_offset()
        joined = np.concatenate(cols + [sep * np.ones((act_ids.shape[0], 1)), act_ids], axis=1).flatten().astype(np.int64)

        action_flag, position_id = _get_action_flag_and_position_id(0, len(joined) - 1, self.observation_dim, self.action_dim, n_prompt)
        step = self.observation_dim + self.action_dim + 1
        if end_ind > path_length:
            action_flag[(path_length - start_ind) * step:] = 0
        target = self.output_sequence_length + 1
        position_id = _truncate_or_pad_to_match_seq_len(position_id, target)
        action_flag = _truncate_or_pad_to_match_seq_len(action_flag, target)
        joined = _truncate_or_pad_to_match_seq_len(joined, target)
        if o_image is not None:              # the padded images need their placeholders too
            for i in range(act_ids.shape[0], o_image.shape[0]):
                joined[i * step:min(target, i * step + self.observation_dim)] = -1

        fields = dict(position_id=position_id[:-1], attention_mask=None, text_seq=None, vision_seq=o_image, tensor_seq=joined[:-1],
                      loss_mask=action_flag[1:], label=joined[1:])
        res = self.env.build_task_input(**fields) if hasattr(self.env, "build_task_input") else RLTaskInput(**fields)
        res.apply(lambda x: torch.tensor(x))
        res.apply(lambda x: x[None, ...])
        return (res, (observations, actions)) if with_raw else res

    # ---------------------------------------------------------------- demonstrations for evaluation prompts (rl_dataset.py:812-862)
    def sample_expert_demonstration(self, strategy: str, strict_length: bool, sample_peak: bool, returns: Optional[np.ndarray] = None):
        """encoded prompt of ``prompt_transition_num`` (``fixed_prompt``) or ``transition_num`` transitions; ``sample_peak`` draws from
        the top 10 % of trajectories by ``returns`` (the reference keeps per-trajectory returns from the d4rl rewards)"""
        want = self.prompt_transition_num if strategy == "fixed_prompt" else self.transition_num
        if sample_peak:
            assert returns is not None and len(returns) == len(self.path_lengths)
            order = sorted(range(len(returns)), key=lambda i: returns[i], reverse=True)
            candidates = order[:int(len(order) * 0.1)]
        else:
            candidates = np.arange(len(self.path_lengths))
        obs_parts, act_parts, have = [], [], 0
        while True:
            o, a = self.get_obs_action_by_path_idx(self.rng.choice(candidates))
            obs_parts.append(o); act_parts.append(a)
            have += len(a)
            if not strict_length or have >= want:
                break
        obs = _map(lambda *xs: np.concatenate(xs, axis=0)[:want], *obs_parts)
        act = np.concatenate(act_parts, axis=0)[:want]
        (o_text, o_image, o_tensor), act_ids = self.postprocess_obs_and_act(obs, act)
        return {"actions": act_ids, "obs/text": o_text, "obs/image": o_image, "obs/tensor": o_tensor}


class RLDataset(torch.utils.data.Dataset):
    """a subset (``documents`` = indices) of an ``RLFullDataset`` (rl_dataset.py:892-924; same positional signature as GPTDataset)"""

    def __init__(self, unused_name, unused_data_prefix, documents: np.ndarray, underlying_dataset, *unused):
        assert documents.ndim == 1 and documents.min() >= 0 and documents.max() < len(underlying_dataset)
        self.dataset, self.indices = underlying_dataset, documents
        self.uses_device_tokenizer = getattr(underlying_dataset, "uses_device_tokenizer", False)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, idx):
        return self.dataset[self.indices[idx % len(self.indices)]]
