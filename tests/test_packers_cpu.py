"""Sample builders: bdm_db1_amd.data.rl_dataset.RLFullDataset and bdm_db1_amd.data.coco_token_dataset against samples the
REFERENCE built from the same in-memory trajectories / caption records (tests/golden/rl_dataset.npz, caption_vqa.npz; generator:
make_golden.py gen_rl_dataset / gen_caption_vqa).  Integer outputs (ids, position ids, masks) are compared exactly.

The product discretizer is the HIP tokenizer (needs an MI355X); on the CPU leg the packer is handed a discretizer backed by the oracle
(test infrastructure, itself pinned to the reference) -- the packer takes its tokenizers as arguments, like the reference.  The GPU
leg (marked gpu) runs the same comparison with bdm_db1_amd.tokenizer.ContinuousScalarTokenizer."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

torch = pytest.importorskip("torch")
from golden_util import RL_DS_CASES, caption_samples, rl_trajectories  # noqa: E402
from oracle import db1_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


class _Tok:
    vocab_size = 32000
    eos_token_id = 0


class _OracleDiscretizer:
    num_continuous_bin = 1024

    def discretize(self, x, is_action):
        return torch.from_numpy(O.mulaw_discretize(np.asarray(x, np.float32), is_action))


def _check_rl(discretizer):
    from bdm_db1_amd.data.rl_dataset import RLDataset, RLFullDataset
    gold = dict(np.load(os.path.join(GOLD, "rl_dataset.npz")))
    for name, (ok, ak, L, opts, idxs, seed) in RL_DS_CASES.items():
        ds = RLFullDataset(rl_trajectories(ok, ak), L, (_Tok(), discretizer), **opts)
        meta = [ds.observation_dim, ds.action_dim, ds.transition_num, ds.prompt_transition_num, len(ds)]
        assert meta == gold[f"{name}/meta"].tolist(), name
        np.random.seed(seed)
        for j, idx in enumerate(idxs):
            r = ds.get(idx)
            assert type(r).__name__ == "RLTaskInput" and r.text_seq is None and r.attention_mask is None
            for f in ("position_id", "loss_mask", "label", "tensor_seq"):
                got = getattr(r, f)
                assert got.shape == (1, L) and got.dtype == torch.int64, (name, j, f)
                assert np.array_equal(got.numpy(), gold[f"{name}/{j}/{f}"]), (name, j, f)
            if f"{name}/{j}/vision_seq" in gold:
                assert np.array_equal(r.vision_seq.numpy(), gold[f"{name}/{j}/vision_seq"]), (name, j)
            else:
                assert r.vision_seq is None
        if name == "vec_cont_noprompt":
            np.random.seed(99)
            for j, (strategy, strict) in enumerate((("fixed_prompt", False), ("moving_prompt", True))):
                d = ds.sample_expert_demonstration(strategy, strict, False)
                assert np.array_equal(d["actions"], gold[f"{name}/demo{j}/actions"]) and np.array_equal(d["obs/tensor"], gold[f"{name}/demo{j}/tensor"])
            sub = RLDataset("n", "p", np.array([3, 1]), ds)
            assert len(sub) == 2 and torch.equal(sub[0].tensor_seq, ds[3].tensor_seq) and torch.equal(sub[3].label, ds[1].label)


def test_rl_sample_builder_matches_reference_golden():
    _check_rl(_OracleDiscretizer())


@pytest.mark.gpu
def test_rl_sample_builder_with_the_hip_tokenizer():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bdm_db1_amd.tokenizer import ContinuousScalarTokenizer
    _check_rl(ContinuousScalarTokenizer())


def test_caption_and_vqa_sample_builders_match_reference_golden():
    from bdm_db1_amd.data.coco_token_dataset import (ICDataset, VQADataset, fit_caption_length, get_loss_mask_vqa,
                                                     get_ltor_masks_and_position_ids)
    gold = dict(np.load(os.path.join(GOLD, "caption_vqa.npz")))
    ic, vqa = caption_samples()
    args = SimpleNamespace(n_position=48, eod_mask_loss=False)
    for name, cls, samples, kind in (("ic", ICDataset, ic, "ICTaskInput"), ("vqa", VQADataset, vqa, "VQATaskInput")):
        ds = cls(args, samples, _Tok())
        assert len(ds) == 3
        for j in range(3):
            r = ds[j]
            assert type(r).__name__ == kind and r.img_seq.dtype == torch.half and r.img_seq.dim() == 4
            for f in ("loss_mask", "label", "prompt_seq", "text_seq"):
                got = getattr(r, f).numpy()
                assert got.dtype == gold[f"{name}/{j}/{f}"].dtype and np.array_equal(got, gold[f"{name}/{j}/{f}"]), (name, j, f)
            assert np.array_equal(r.img_seq.float().numpy(), gold[f"{name}/{j}/img_seq"])
            # the three segments the model concatenates fill n_position exactly when the image has (L - prompt - text) patches
            assert r.label.shape == (1, 48) and r.loss_mask.shape == (1, 48)
    j = 0
    while f"ltor/{j}/data" in gold:
        att, lm, pid = get_ltor_masks_and_position_ids(gold[f"ltor/{j}/data"], 0, int(gold[f"ltor/{j}/full"]))
        assert att is None and lm.dtype == np.float32 and pid.dtype == np.int32
        assert np.array_equal(lm, gold[f"ltor/{j}/loss_mask"]) and np.array_equal(pid, gold[f"ltor/{j}/position_ids"])
        j += 1
    assert j == 3
    for j, n in enumerate((4, 9, 20)):
        got = fit_caption_length([11, 12, 13, 14, 15, 16, 17, 18, 19], n)
        assert got.dtype == torch.int32 and np.array_equal(got.numpy(), gold[f"fit/{j}"])
    # a list answer is not eos-masked (reference quirk), an array answer is
    assert get_loss_mask_vqa([4, 0, 5], 0, False, 6).tolist() == [0, 0, 0, 1, 1, 1]
    assert get_loss_mask_vqa(np.array([4, 0, 5]), 0, False, 6).tolist() == [0, 0, 0, 1, 1, 0]
