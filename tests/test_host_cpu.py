"""CPU-only checks (run in the build container and on any box without a GPU):
the C-ABI library loads and exports every symbol include/db1_hip.h declares (no compute call is made),
host-side logic (schedule, synthetic packers, sharding rule, mpu surface), and the data-parallel gradient
synchronisation on 2 ranks over gloo."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from oracle import db1_oracle as O  # noqa: E402


def test_library_exports_every_declared_symbol():
    from bdm_db1_amd import lib
    from bdm_db1_amd.build import build_lib
    build_lib()
    names = lib.declared_symbols()
    assert len(names) >= 38 and "db1_gemm_strided" in names and "db1_relattn_flash_bwd" in names and "db1_adam_step" in names
    handle = lib.load()  # getattr on every declared name: AttributeError if one is missing
    for n in names:
        assert hasattr(handle, n), n
    assert handle.db1_version() >= 100
    assert isinstance(handle.db1_last_error(), (bytes, type(None)))


def test_ops_refuse_cpu_tensors_instead_of_falling_back():
    from bdm_db1_amd import lib, ops
    a = torch.zeros(4, 4)
    with pytest.raises(lib.Db1Error):
        ops.gemm(a, a, a)
    with pytest.raises(lib.Db1Error):
        ops.add(a, a, a)


def test_model_construction_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bdm_db1_amd import TransformerXL, lib, synth
    with pytest.raises(lib.Db1Error):
        TransformerXL(synth.db1_config("tiny"))


def test_scheduler_matches_reference_golden():
    from bdm_db1_amd.optim import OptimizerParamScheduler
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "scheduler.npz")))
    for style in ("constant", "linear", "cosine"):
        for wstyle in ("constant", "linear", "cosine"):
            opt = SimpleNamespace(param_groups=[{"lr": 0.0, "weight_decay": 0.0}])
            s = OptimizerParamScheduler(opt, max_lr=1e-3, min_lr=1e-5, lr_warmup_steps=10, lr_decay_steps=100, lr_decay_style=style,
                                        start_wd=0.01 if wstyle == "constant" else 0.0, end_wd=0.01, wd_incr_steps=80, wd_incr_style=wstyle)
            prev, lrs, wds = 0, [], []
            for st in gold["steps"]:
                s.step(int(st - prev))
                prev = st
                lrs.append(opt.param_groups[0]["lr"])
                wds.append(opt.param_groups[0]["weight_decay"])
            np.testing.assert_allclose(lrs, gold[f"lr/{style}/{wstyle}"], rtol=1e-12)
            np.testing.assert_allclose(wds, gold[f"wd/{style}/{wstyle}"], rtol=1e-12)
            sd = s.state_dict()
            s2 = OptimizerParamScheduler(SimpleNamespace(param_groups=[{}]), 1e-3, 1e-5, 10, 100, style,
                                         0.01 if wstyle == "constant" else 0.0, 0.01, 80, wstyle)
            s2.load_state_dict(sd)
            assert s2.num_steps == s.num_steps and s2.get_lr() == s.get_lr()


def test_scheduler_checkpoint_contract_follows_the_reference():
    """ADVICE r5 (optimizer_param_scheduler.py:158-234): older checkpoints' key names are read, a missing mandatory key is a KeyError, the
    step count of the checkpoint is ADDED to the counter, and a schedule that disagrees with the checkpoint's (neither flag set) -- like the
    constructor's own guards -- fails with AssertionError, which is what callers of the reference catch"""
    from bdm_db1_amd.optim import OptimizerParamScheduler
    mk = lambda **kw: OptimizerParamScheduler(SimpleNamespace(param_groups=[{}]), 1e-3, 1e-5, 10, 100, "cosine", 0.0, 0.01, 80, "linear", **kw)
    legacy = {"start_lr": 2e-3, "min_lr": 1e-6, "warmup_iter": 5, "end_iter": 200, "decay_style": "linear", "num_iters": 7}
    s = mk()
    s.load_state_dict(legacy)
    assert (s.max_lr, s.min_lr, s.lr_warmup_steps, s.lr_decay_steps, s.lr_decay_style, s.num_steps) == (2e-3, 1e-6, 5, 200, "linear", 7)
    assert (s.start_wd, s.end_wd, s.wd_incr_steps, s.wd_incr_style) == (0.0, 0.01, 80, "linear")      # no weight-decay keys: the constructor's stay
    s.load_state_dict({"max_lr": 2e-3, "min_lr": 1e-6, "warmup_steps": 5, "decay_steps": 200, "lr_decay_style": "linear", "num_steps": 3})
    assert s.num_steps == 10                                                                           # added, as self.step(increment=...) does
    for missing in ("min_lr", "lr_decay_style", "num_steps"):
        sd = mk().state_dict()
        del sd[missing]
        with pytest.raises(KeyError):
            mk().load_state_dict(sd)
    sd = mk().state_dict()
    del sd["wd_incr_steps"]                     # the weight-decay group is optional as a whole ("start_wd" present: all four are looked up)
    with pytest.raises(KeyError):
        mk().load_state_dict(sd)
    sd = mk().state_dict()
    sd["lr_decay_steps"] = 300
    with pytest.raises(AssertionError):
        mk(use_checkpoint_opt_param_scheduler=False).load_state_dict(sd)
    s = mk(use_checkpoint_opt_param_scheduler=False, override_opt_param_scheduler=True)
    s.load_state_dict(sd)
    assert s.lr_decay_steps == 100              # override: the constructed schedule wins
    for bad in (dict(lr_warmup_steps=100), dict(min_lr=-1.0), dict(end_wd=-0.5)):
        args = dict(max_lr=1e-3, min_lr=1e-5, lr_warmup_steps=10, lr_decay_steps=100, lr_decay_style="linear", start_wd=0.0, end_wd=0.01, wd_incr_steps=80,
                    wd_incr_style="linear")
        args.update(bad)
        with pytest.raises(AssertionError):
            OptimizerParamScheduler(SimpleNamespace(param_groups=[{}]), **args)
    OptimizerParamScheduler(SimpleNamespace(param_groups=[{}]), 1e-3, 1e-5, -3, 100, "linear", 0.0, 0.01, 80, "linear")   # (a negative warm-up is accepted there too)
    with pytest.raises(AssertionError):
        OptimizerParamScheduler(SimpleNamespace(param_groups=[{}]), 1e-3, 1e-5, 10, 100, "linear", 0.0, 0.01, 80, "linear", True, True)


def test_synthetic_rl_layout_matches_reference_packer():
    from bdm_db1_amd import synth
    cfg = synth.db1_config("1.3B")
    b = synth.rl_batch(2, 1024, 5, "cpu", cfg)
    assert b.tensor_seq.shape == (2, 1024) and b.vision_seq.shape == (2, 47, 3, 64, 80)
    flag, pos = O.rl_action_flag_and_position_id(0, 1023, 20, 1, 0)  # pinned to rl_dataset.py:44-71 by golden vectors
    assert np.array_equal(b.position_id[0].numpy(), pos)
    ids = b.tensor_seq[0].numpy()
    assert (ids[:20] == -1).all() and ids[20] == 33024 and 0 <= ids[21] < 18
    # the loss mask marks the positions whose LABEL is an action token (= the separator positions)
    lab = b.label[0].numpy()
    lm = b.loss_mask[0].numpy()
    assert np.array_equal(np.nonzero(lm)[0], np.nonzero(ids == 33024)[0])
    assert ((lab[lm == 1] >= 0) & (lab[lm == 1] < 18)).all()
    assert int((ids == -1).sum()) <= 47 * 20
    c = synth.caption_batch(2, 1024, 7, "cpu", cfg)
    assert c.prompt_seq.shape == (2, 8) and c.img_seq.shape == (2, 3, 224, 224) and c.text_seq.shape == (2, 1024 - 8 - 196)
    t = synth.text_batch(3, 1024, 1, "cpu")
    assert torch.equal(t.text_seq[:, 1:], t.label[:, :-1])


def test_dp_sharding_rule():
    from bdm_db1_amd import synth
    # data_samplers.py:152-155: rank r takes rows [r*mb, (r+1)*mb) of every global chunk
    assert synth.dp_shard(32, 4, 1, 4) == [(4, 8), (20, 24)]
    rows = sorted(r for rank in range(4) for s, e in synth.dp_shard(32, 4, rank, 4) for r in range(s, e))
    assert rows == list(range(32))


def test_mpu_surface_single_process():
    from bdm_db1_amd import mpu
    for name in ("initialize_model_parallel", "model_parallel_is_initialized", "get_data_parallel_group", "get_data_parallel_rank",
                 "get_data_parallel_world_size", "get_model_parallel_group", "get_model_parallel_rank", "get_model_parallel_world_size",
                 "get_tensor_model_parallel_group", "get_tensor_model_parallel_rank", "get_tensor_model_parallel_world_size",
                 "get_pipeline_model_parallel_group", "get_pipeline_model_parallel_rank", "get_pipeline_model_parallel_world_size",
                 "is_pipeline_first_stage", "is_pipeline_last_stage", "get_embedding_group", "destroy_model_parallel", "is_unitialized",
                 "print_rank_0", "print_with_rank", "divide", "split_tensor_along_last_dim", "VocabUtility",
                 "get_tensor_model_parallel_src_rank", "get_virtual_pipeline_model_parallel_rank"):
        assert hasattr(mpu, name), name
    mpu.destroy_model_parallel()
    assert mpu.is_unitialized()
    mpu.initialize_model_parallel()
    assert mpu.model_parallel_is_initialized() and mpu.get_data_parallel_world_size() == 1 and mpu.get_data_parallel_rank() == 0
    assert mpu.get_tensor_model_parallel_world_size() == 1 and mpu.is_pipeline_first_stage() and mpu.is_pipeline_last_stage()
    with pytest.raises(NotImplementedError):
        mpu.destroy_model_parallel()
        mpu.initialize_model_parallel(2, 1)
    mpu.destroy_model_parallel()
    assert mpu.divide(12, 4) == 3 and mpu.VocabUtility.vocab_range_from_global_vocab_size(100, 1, 4) == (25, 50)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bdm_db1_amd import mpu
    from bdm_db1_amd.engine import GradSync
    mpu.initialize_model_parallel()
    assert mpu.get_data_parallel_world_size() == world and mpu.get_data_parallel_rank() == rank
    assert dist.get_world_size(mpu.get_model_parallel_group()) == 1
    # the arena of a 3-layer model: buckets in backward-completion order
    n = 1000
    buckets = [("h.2", 0, 300), ("h.1", 300, 600), ("h.0", 600, 900), ("embeddings", 900, 1000)]
    rng = np.random.default_rng(rank)
    g = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    local = g.clone()
    sync = GradSync(g, buckets, mpu.get_data_parallel_group())
    sync.launch("h.2")  # hooks fire as layers finish their backward ...
    sync.launch("h.1")
    sync.launch("h.1")  # ... a duplicate launch must be a no-op
    sync.finish()       # ... whatever was not launched is reduced here
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, local, group=mpu.get_data_parallel_group())
    expect = sum(gathered)
    ok = torch.allclose(g, expect, atol=1e-6)
    # second step re-uses the object
    g.copy_(local)
    sync.finish()
    ok = ok and torch.allclose(g, expect, atol=1e-6)
    # bf16 staging (what the GPUs send over xGMI): each bucket is cast into its slice of the staging tensor, the STAGING slice is
    # reduced, the fp32 arena stays this rank's own sum; `reduced` is what the optimizer reads
    g.copy_(local)
    stage = torch.zeros(n, dtype=torch.bfloat16)
    sync16 = GradSync(g, buckets, mpu.get_data_parallel_group(), stage=stage, cast=lambda src, dst: dst.copy_(src))
    sync16.launch("h.0")
    sync16.finish()
    expect16 = sum(x.to(torch.bfloat16).float() for x in gathered)
    ok = ok and sync16.reduced is stage and torch.equal(g, local)
    ok = ok and torch.allclose(stage.float(), expect16, atol=2e-2, rtol=1e-2)   # one bf16 rounding of the sum
    all16 = [torch.zeros(n, dtype=torch.bfloat16) for _ in range(world)]
    dist.all_gather(all16, stage)
    ok = ok and all(torch.equal(all16[0], x) for x in all16)                    # every rank holds the same reduced gradients
    # per-bucket norm of the REDUCED gradients, collected behind each bucket's all-reduce (engine: the global-norm clip then needs only the
    # sum of the partial sums after the last bucket): equals the norm of the reduced tensor, on every rank, also on a re-used object
    g.copy_(local)
    stage.zero_()
    sq = lambda buf, out: out.copy_((buf.float() ** 2).sum().reshape(1))
    syncn = GradSync(g, buckets, mpu.get_data_parallel_group(), stage=stage, cast=lambda src, dst: dst.copy_(src), norm_sq=sq)
    for _ in range(2):
        syncn.launch("h.2")
        syncn.finish()
        ok = ok and syncn.norm_parts.numel() == len(buckets)
        ok = ok and torch.allclose(syncn.reduced_norm_sq(), (stage.float() ** 2).sum().reshape(1), rtol=1e-6)
        ok = ok and torch.allclose(syncn.norm_parts[3], (stage[900:1000].float() ** 2).sum(), rtol=1e-6)
    # the data-parallel group is the default communicator itself: no second communicator over the same ranks
    ok = ok and mpu.get_data_parallel_group() is dist.group.WORLD
    # mean-of-ranks via the optimizer's gradient scale (the arena holds the SUM): the update every rank applies is identical
    g.copy_(expect)
    p = torch.ones(n)
    upd = p - 0.1 * g * (1.0 / world)
    allp = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(allp, upd)
    ok = ok and all(torch.equal(allp[0], x) for x in allp)
    q.put((rank, bool(ok)))
    dist.barrier()
    mpu.destroy_model_parallel()
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_input_spec_classes_have_reference_field_names():
    from bdm_db1_amd.data import GatoInputBase, ICTaskInput, NLPTaskInput, RLTaskInput, VQATaskInput
    import dataclasses
    f = lambda c: [x.name for x in dataclasses.fields(c)]
    assert f(GatoInputBase) == ["position_id", "attention_mask", "loss_mask", "label"]            # input_specs.py:23-29
    assert f(RLTaskInput)[4:] == ["text_seq", "vision_seq", "tensor_seq"]                          # :72-77
    assert f(NLPTaskInput)[4:] == ["text_seq", "text_len"]                                         # :79-83
    assert f(ICTaskInput)[4:] == ["prompt_seq", "img_seq", "text_seq", "img_id_seq"]               # :85-96
    assert f(VQATaskInput)[4:] == ["prompt_seq", "img_seq", "text_seq", "img_id_seq", "ques_id_seq", "ques_len"]  # :98-112
    x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=torch.ones(2, 3), label=torch.zeros(2, 3), text_seq=torch.zeros(2, 3), text_len=None)
    x.to(dtype=torch.float64)
    assert x.loss_mask.dtype == torch.float64


def test_samplers_and_collate_match_reference_golden():
    from bdm_db1_amd.data import NLPTaskInput, RLTaskInput
    from bdm_db1_amd.data.samplers import RandomPretrainingSampler, SequentialPretrainingSampler, my_collate_fn
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "samplers.npz")))
    for i, (tot, cons, mb, rank, world) in enumerate(gold["cases"].tolist()):
        if cons < tot:
            got = np.array(list(SequentialPretrainingSampler(tot, cons, mb, rank, world)), dtype=np.int64).reshape(-1, mb)
            assert np.array_equal(got, gold[f"seq{i}"]), ("seq", i)
            last = list(SequentialPretrainingSampler(tot, cons, mb, rank, world, drop_last=False))[-1]
            assert np.array_equal(np.array(last, dtype=np.int64), gold[f"seq_last{i}"]), ("seq_last", i)
        for sharding in (True, False):
            s = RandomPretrainingSampler(list(range(tot)), tot, cons, mb, rank, world, sharding)
            got = np.array(list(s), dtype=np.int64).reshape(-1, mb)
            assert np.array_equal(got, gold[f"rand{int(sharding)}_{i}"]), ("rand", sharding, i)
            assert s.consumed_samples == int(gold[f"rand{int(sharding)}_{i}_consumed"])
    mk = lambda v: torch.full((1, 6), v, dtype=torch.int64)
    tasks = [NLPTaskInput(position_id=None, attention_mask=None, loss_mask=mk(1).float(), label=mk(10), text_seq=mk(11), text_len=None),
             NLPTaskInput(position_id=None, attention_mask=None, loss_mask=mk(2).float(), label=mk(20), text_seq=mk(21), text_len=None),
             RLTaskInput(position_id=mk(3), attention_mask=None, loss_mask=mk(3).float(), label=mk(30), text_seq=None,
                         vision_seq=torch.full((1, 2, 3, 16, 16), 3.0), tensor_seq=mk(31)),
             NLPTaskInput(position_id=None, attention_mask=None, loss_mask=mk(4).float(), label=mk(40), text_seq=mk(41), text_len=None)]
    merged = my_collate_fn(tasks)
    assert [type(m).__name__ for m in merged] == gold["collate_types"].tolist()
    assert np.array_equal(merged[0].label.numpy(), gold["collate_nlp_label"])
    assert np.array_equal(merged[0].text_seq.numpy(), gold["collate_nlp_text"])
    assert list(merged[1].vision_seq.shape) == gold["collate_rl_vision_shape"].tolist()
    assert np.array_equal(merged[1].tensor_seq.numpy(), gold["collate_rl_tensor"])


def test_rl_packers_match_reference_golden():
    """bdm_db1_amd.data._get_action_flag_and_position_id / _truncate_or_pad_to_match_seq_len (rl_dataset.py:44-71,865-872)"""
    from bdm_db1_amd.data import _get_action_flag_and_position_id, _truncate_or_pad_to_match_seq_len
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "rl_packing.npz")))
    i = 0
    while f"args{i}" in gold:
        f, p = _get_action_flag_and_position_id(*[int(x) for x in gold[f"args{i}"]])
        assert f.dtype == np.int64 and (f == gold[f"flag{i}"]).all() and (p == gold[f"pos{i}"]).all(), i
        i += 1
    assert i >= 5
    assert (_truncate_or_pad_to_match_seq_len(gold["pad_in"], 8) == gold["pad8"]).all()
    assert (_truncate_or_pad_to_match_seq_len(gold["pad_in"], 3) == gold["pad3"]).all()
    assert _truncate_or_pad_to_match_seq_len(gold["pad_in"], 5) is gold["pad_in"]


def test_generated_gemm_loops_are_in_sync_with_their_generator(tmp_path):
    """bdm_db1_amd/csrc/gemm_w4_loop_{nt,nn,tn}.inc are generated (tools/gen_gemm_w4.py): the committed files must be what the
    generator emits with its defaults, and every loop must hold 4 k-tile bodies x 128 MFMAs with one accumulator tuple per slot"""
    import re
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    env = {k: v for k, v in os.environ.items() if not k.startswith(("W4_", "W4N_"))}
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_gemm_w4.py"), str(tmp_path)], check=True, env=env, capture_output=True)
    for name in ("nt", "nn", "tn"):
        fresh = open(os.path.join(tmp_path, f"gemm_w4_loop_{name}.inc")).read()
        committed = open(os.path.join(root, "bdm_db1_amd", "csrc", f"gemm_w4_loop_{name}.inc")).read()
        assert fresh == committed, f"gemm_w4_loop_{name}.inc is stale: run python tools/gen_gemm_w4.py"
        mf = re.findall(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):(\d+)\]", committed)
        assert len(mf) == 4 * 128
        for body in range(4):   # every k-tile touches each of the 64 accumulator tuples exactly twice (ks 0 and ks 1)
            starts = sorted(int(a) for a, _ in mf[body * 128:(body + 1) * 128])
            assert starts == sorted(list(range(0, 256, 4)) * 2)
        # the LDS-DMA requests: 2 prologue k-tiles + 2 loop bodies, 16 each; none in the two peeled k-tiles
        assert committed.count("global_load_lds_dwordx4") == 64
        assert committed.count("s_barrier") == 1 + 2 * 3 + 1
        # the 256 x 128-tile form (gemm_w4n_loop_*.inc): 64 MFMAs per k-tile on the accumulator tuples a[4 (8 i + j)], j < 4; 12 requests per k-tile
        fresh = open(os.path.join(tmp_path, f"gemm_w4n_loop_{name}.inc")).read()
        committed = open(os.path.join(root, "bdm_db1_amd", "csrc", f"gemm_w4n_loop_{name}.inc")).read()
        assert fresh == committed, f"gemm_w4n_loop_{name}.inc is stale: run python tools/gen_gemm_w4.py"
        mf = re.findall(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):(\d+)\]", committed)
        assert len(mf) == 4 * 64
        want = sorted([4 * (8 * i + j) for i in range(8) for j in range(4)] * 2)
        for body in range(4):
            assert sorted(int(a) for a, _ in mf[body * 64:(body + 1) * 64]) == want
        assert committed.count("global_load_lds_dwordx4") == 48
        assert committed.count("s_barrier") == 1 + 2 * 3 + 1


def _dp8_worker(rank, world, port, q):
    """eight ranks, the engine's use of GradSync over three optimizer steps with gradient accumulation 3: buckets in backward-completion
    order (decoder layers last-to-first, then the ragged `embeddings` tail bucket), hooks only on the boundary micro-step (earlier
    micro-steps accumulate locally and must not communicate), the layer hooks fired in backward order with `embeddings` left to
    finish(), bf16 staging, the data-parallel mean folded into the update"""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bdm_db1_amd import mpu
    from bdm_db1_amd.engine import GradSync
    mpu.initialize_model_parallel()
    nl, per, tail = 5, 96, 37                       # five "layers" of 96 gradients + a ragged 37-element embedding bucket
    n = nl * per + tail
    buckets = [(f"h.{i}", (nl - 1 - i) * per, (nl - i) * per) for i in reversed(range(nl))] + [("embeddings", nl * per, n)]   # arena order = completion order
    assert [b[0] for b in buckets][0] == f"h.{nl - 1}" and buckets[-1] == ("embeddings", nl * per, n)
    g = torch.zeros(n)
    stage = torch.zeros(n, dtype=torch.bfloat16)
    calls = []

    def cast(src, dst):
        calls.append((src.data_ptr() - g.data_ptr()) // 4)
        dst.copy_(src)
    sync = GradSync(g, buckets, mpu.get_data_parallel_group(), stage=stage, cast=cast)
    ga, ok = 3, True
    p = torch.ones(n)
    for step in range(3):
        g.zero_()
        for micro in range(ga):
            torch.manual_seed(1000 * step + 10 * micro + rank)
            g += torch.randn(n)                                   # this micro-step's local gradients, accumulated in the arena
            boundary = micro == ga - 1
            before = len(calls)
            if boundary:                                          # engine.backward: the hook is sync.launch on the boundary micro-step only
                for i in reversed(range(nl)):
                    sync.launch(f"h.{i}")
            else:
                ok = ok and len(sync.handles) == 0 and len(calls) == before   # nothing is cast or sent before the boundary
        local = g.clone()
        ok = ok and len(sync.handles) == nl                       # five collectives in flight, the tail bucket not yet
        sync.finish()                                             # engine.step: the tail bucket, then wait for all
        ok = ok and calls[-(nl + 1):] == [s for _, s, _ in buckets]                  # cast (= launch) order is the arena order
        gathered = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(gathered, local)
        expect = sum(x.to(torch.bfloat16).float() for x in gathered)
        ok = ok and torch.equal(g, local)                         # the fp32 arena stays this rank's own sum
        ok = ok and torch.allclose(sync.reduced.float(), expect, atol=6e-2, rtol=2e-2)   # eight bf16 terms, bf16 partial sums on the wire
        allst = [torch.zeros(n, dtype=torch.bfloat16) for _ in range(world)]
        dist.all_gather(allst, stage)
        ok = ok and all(torch.equal(allst[0], x) for x in allst)  # every rank reads the same reduced gradients ...
        p = p - 0.1 * sync.reduced.float() * (1.0 / (world * ga))
        allp = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(allp, p)
        ok = ok and all(torch.equal(allp[0], x) for x in allp)    # ... so the replicas stay bit-identical over the steps
    q.put((rank, bool(ok)))
    dist.barrier()
    mpu.destroy_model_parallel()
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_eight_ranks_gloo_with_accumulation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(8)]


def test_generated_flash_loops_are_in_sync_with_their_generators(tmp_path):
    """relattn_flash_fwd{2,3}_loop.inc are generated (tools/gen_flash_fwd.py, tools/gen_flash_fwd3.py): the committed files must be what
    the generators emit with their defaults; every iteration path ends in exactly one barrier, and a steady iteration holds the MFMAs of
    one score phase + one P.V phase"""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    env = {k: v for k, v in os.environ.items() if not k.startswith(("FW2_", "FW3_", "FW_"))}
    for gen, inc, n_mfma_steady in (("gen_flash_fwd.py", "relattn_flash_fwd2_loop.inc", 24), ("gen_flash_fwd3.py", "relattn_flash_fwd3_loop.inc", 50)):
        subprocess.run([sys.executable, os.path.join(root, "tools", gen), str(tmp_path)], check=True, env=env, capture_output=True)
        fresh = open(os.path.join(tmp_path, inc)).read()
        committed = open(os.path.join(root, "bdm_db1_amd", "csrc", inc)).read()
        assert fresh == committed, f"{inc} is stale: run python tools/{gen}"
        assert committed.count("s_barrier") == 6 * 3 + 1          # steady / last / idle of the six unrolled instances + the first iteration
        body = committed.split("L_ns1_%=:")[0].split("L_inst1_%=:")[-1]  # instance 1: from its label to the end of its steady path
        assert body.count("v_mfma_f32_16x16x32_bf16") == n_mfma_steady
