"""Data-parallel training semantics on the GPU kernels: two ranks, each with half of a micro-batch, must take the same
optimizer steps as one rank with the whole micro-batch (per-rank loss normalised by its own mask sum, gradients summed by the
bucketed asynchronous all-reduce launched from the backward, mean + clip folded into the fused Adam scale).  Both ranks share the
one GPU of the test box, so the process group is gloo (RCCL refuses two ranks on one device); the engine code path -- hooks,
bucket slices of the flat arena, handle waits, Adam scaling -- is the one the RCCL run uses."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(name="small_window"):
    from golden_util import CASES, case_cfg, make_params
    from bdm_db1_amd import TransformerXL
    seed = 100 + list(CASES).index(name)
    cfg = case_cfg(name)
    params = make_params(cfg, seed)
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", f"model_{name}.npz")))
    params["pos_emb.inv_freq"] = gold["inv_freq"]
    model = TransformerXL(SimpleNamespace(**cfg), compute_dtype=torch.float32)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    return cfg, model


def _batch(cfg, rows):
    from bdm_db1_amd.data import NLPTaskInput
    rng = np.random.default_rng(77)
    L = cfg["n_position"]
    ids = rng.integers(0, cfg["text_vocab_size"], (4, L + 1))
    sel = ids[rows]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
    return NLPTaskInput(position_id=None, attention_mask=None, loss_mask=T(np.ones((len(rows), L), np.float32)), label=T(sel[:, 1:]),
                        text_seq=T(sel[:, :-1]), text_len=None)


# adam_eps is deliberately large: with the default 1e-8 Adam is invariant to the gradient scale and would hide a wrong mean over ranks
ARGS = dict(lr=2e-3, weight_decay=0.01, clip_grad=0.5, optimizer="adamw", keep_logits=True, adam_eps=1e-3)


def _train(model, cfg, rows, mpu=None, steps=3, reduce_dtype="fp32"):
    from bdm_db1_amd import initialize
    engine, _, _, _ = initialize(SimpleNamespace(grad_reduce_dtype=reduce_dtype, **ARGS), model, mpu=mpu)
    engine.train()
    losses = []
    for _ in range(steps):
        logits, loss = engine([_batch(cfg, rows)])
        engine.backward(loss)
        engine.step()
        losses.append(float(loss))
    return losses


def _worker(rank, world, port, q, backend="gloo", reduce_dtype="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if backend == "nccl":   # RCCL over xGMI: one process per GPU, the engine's own initialiser (high-priority comm stream)
        from bdm_db1_amd.engine import init_distributed
        init_distributed(dist_backend="nccl")
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from bdm_db1_amd import mpu
    mpu.initialize_model_parallel()
    cfg, model = _make()
    losses = _train(model, cfg, [2 * rank, 2 * rank + 1], mpu=mpu, reduce_dtype=reduce_dtype)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    q.put((rank, losses, sd if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks_vs_one(backend, reduce_dtype, tol):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend, reduce_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    cfg, model = _make()
    ref_losses = _train(model, cfg, [0, 1, 2, 3])
    # the full-batch loss is the mean of the two half-batch losses (equal mask sums)
    ltol = 2e-5 if reduce_dtype == "fp32" else 2e-3
    for s in range(len(ref_losses)):
        assert abs(0.5 * (got[0][1][s] + got[1][1][s]) - ref_losses[s]) < ltol * max(1.0, abs(ref_losses[s])), s
    sd = got[0][2]
    worst = 0.0
    for k, v in model.state_dict().items():
        ref = v.detach().cpu().numpy().astype(np.float64)
        worst = max(worst, float(np.abs(sd[k] - ref).max() / (np.abs(ref).max() + 1e-30)))
    assert worst < tol, f"parameters after 3 data-parallel steps differ from the single-rank run: {worst:.2e}"


def test_two_ranks_equal_one_rank_with_the_whole_batch():
    """fp32 on the wire: fp32 sums in a different order (two half-batch gradients added by the all-reduce vs one full-batch
    reduction) -> ~1e-5 after 3 steps; a wrong mean / clip scale would show as >= 1e-3 here"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _two_ranks_vs_one("gloo", "fp32", 2e-4)


def test_two_ranks_bf16_gradient_reduction():
    """the default: each bucket is cast to bf16, the staging copy is all-reduced and Adam reads it.  Three AdamW steps at lr 2e-3
    with 2^-9-relative gradient rounding: parameters within 3e-3 of the fp32 single-rank run (a wrong scale moves them by > 1e-2)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _two_ranks_vs_one("gloo", "bf16", 3e-3)


@pytest.mark.parametrize("reduce_dtype,tol", [("fp32", 2e-4), ("bf16", 3e-3)])
def test_two_ranks_rccl(reduce_dtype, tol):
    """the same check over RCCL (backend "nccl"), one process per GPU: runs wherever two GPUs are visible (the 1-GPU test box skips)"""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    _two_ranks_vs_one("nccl", reduce_dtype, tol)


def test_rccl_one_rank_runs_the_bucket_path():
    """RCCL itself on the one GPU of the test box, ONE rank (`tools/rccl_one_rank.py`, its own process): the engine's initialiser
    (high-priority communication stream, communicator bound to the device), then GradSync's bucket path against the real library -- bf16 staging
    cast, async all-reduce, the per-bucket norm on the side stream behind the collective's stream-level wait, finish(), fp32 buckets, barrier --
    with the world size reported as 2 so that nothing is skipped (a sum over one rank is the identity: the reduced copy must equal the cast
    gradient bit for bit, the per-bucket norms their one-pass sum)"""
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL one-rank path ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert "backend nccl" in r.stdout


def _worker_one_rank_rccl(port, q, reduce_dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from bdm_db1_amd import engine as E, mpu
    E.init_distributed(dist_backend="nccl")
    mpu.initialize_model_parallel()
    cfg, model = _make()
    real = dist.get_world_size
    dist.get_world_size = lambda group=None: 2          # (read while the engine is built: it plans buckets, staging and the 1 / world scale for two ranks)
    try:
        from bdm_db1_amd import initialize
        engine, _, _, _ = initialize(SimpleNamespace(grad_reduce_dtype=reduce_dtype, **ARGS), model, mpu=mpu)
    finally:
        dist.get_world_size = real
    assert engine.dp_world == 2 and engine.sync.world == 2 and engine.sync._side is not None
    plain_backward = model.backward                      # the "other rank's" half: the one-rank sum is g where two ranks would deliver 2 g
    model.backward = lambda grad_scale=1.0, **kw: plain_backward(grad_scale=2.0 * grad_scale, **kw)
    engine.train()
    losses = []
    for _ in range(3):
        logits, loss = engine([_batch(cfg, [0, 1, 2, 3])])
        engine.backward(loss)
        engine.step()
        losses.append(float(loss))
    q.put((losses, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, str(dist.get_backend())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("reduce_dtype,tol", [("fp32", 2e-4), ("bf16", 3e-3)])
def test_engine_steps_over_rccl_with_one_rank(reduce_dtype, tol):
    """three optimizer steps of the engine with its gradients going through RCCL itself (one rank on the box's one GPU, the world size read as 2
    while the engine is built, so the hooks launch every bucket's all-reduce from the backward, the norm is taken per bucket behind the
    collectives and Adam reads the reduced copy).  A sum over one rank is the gradient itself where two ranks would deliver twice that, so the
    worker doubles the loss-gradient scale (exact in binary): clip, norm and Adam (a large eps: not scale-invariant) then see what the plain
    single-rank run sees, and the parameters must follow it (the tolerances of the two-rank tests)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_one_rank_rccl, args=(_free_port(), q, reduce_dtype))
    p.start()
    losses, sd, backend = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0 and backend == "nccl"
    cfg, model = _make()
    ref_losses = _train(model, cfg, [0, 1, 2, 3])
    ltol = 2e-5 if reduce_dtype == "fp32" else 2e-3
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < ltol * max(1.0, abs(b)), (losses, ref_losses)
    worst = 0.0
    for k, v in model.state_dict().items():
        ref = v.detach().cpu().numpy().astype(np.float64)
        worst = max(worst, float(np.abs(sd[k] - ref).max() / (np.abs(ref).max() + 1e-30)))
    assert worst < tol, f"parameters after 3 steps over RCCL differ from the plain single-rank run: {worst:.2e}"


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it must start two ranks itself and print ONE JSON line from rank 0
    (the bare command exited non-zero).  Two GPUs -> RCCL; one GPU -> DB1_DIST_BACKEND=gloo with both ranks on it."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["DB1_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--layers", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["value"] > 0 and "cpu_baseline" not in out
    # BASELINE configs 4 / 5 ride along on the same ranks: RL-trajectory and mixture legs with whole-job tokens/s,
    # per-rank min / max step time and the exposed communication
    for wl in ("rl", "mixture"):
        leg = out[wl]
        assert leg.get("error") is None, leg
        assert leg["n_gpus"] == 2 and leg["tokens_per_s"] > 0 and leg["image_patches_per_step"] > 0
        assert abs(leg["tokens_per_s"] - 2 * leg["sequences_per_gpu"] * 1024 / (leg["ms_per_step"] * 1e-3)) < 1e-2 * leg["tokens_per_s"]
        dp = leg["data_parallel"]
        assert dp["ms_per_step_min"] <= dp["ms_per_step_max"] and dp["exposed_comm_ms_per_step_max"] >= 0.0
    assert "ga16" not in out and "decode" not in out      # (single-GPU legs)


def test_bench_eight_ranks_on_this_box():
    """the command the driver's 8-GPU scaling run issues, `python bench.py --gpus 8`, exercised end to end wherever this test runs:
    eight GPUs -> RCCL, fewer -> DB1_DIST_BACKEND=gloo with the eight ranks sharing the device(s) (2 sequences per rank, the full
    24-layer model in every rank: 8 x 22 GB of parameters / optimizer state + 8 x 2.4 GB staging on one 288 GB device).  Checks: one
    JSON line from rank 0, whole-job tokens, dp8, every rank alive to the end, and that the attention backward chose a mode that fits
    the memory the eight ranks leave (no out-of-memory fallback needed by the caller)."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 8:
        env["DB1_DIST_BACKEND"] = "gloo"
    env["DB1_LAUNCH_TIMEOUT_S"] = "1200"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--batch", "2", "--no-cpu-baseline",
                        "--no-kernel-timing", "--leg-steps", "1"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "dp8" and out["config"]["global_batch"] == 16
    assert out["scaling"] == "weak" and out["value"] > 0 and abs(out["value"] - 8 * 2 * 1024 / (out["ms_per_step"] * 1e-3)) < 1e-2 * out["value"]
    assert "INVALID" not in out and np.isfinite(out["final_loss"]) and 5.0 < out["final_loss"] < 12.0
    assert "forward" in out["config"]["attention_backward"] or "scratch" in out["config"]["attention_backward"] or "recompute" in out["config"]["attention_backward"]
    # the first 8-GPU run must produce BASELINE configs 4 and 5, not only text: all three workloads in ONE line
    for wl in ("rl", "mixture"):
        leg = out[wl]
        assert leg.get("error") is None, leg
        assert leg["n_gpus"] == 8 and leg["tokens_per_s"] > 0 and leg["image_patches_per_step"] > 0 and np.isfinite(leg["final_loss"])
        assert "data_parallel" in leg and leg["data_parallel"]["ms_per_step_min"] <= leg["data_parallel"]["ms_per_step_max"]


def test_self_launch_reports_a_failing_rank(tmp_path):
    """a rank that dies must stop the others and surface ITS stderr (not a silent hang): bench.py with an impossible argument for rank 1"""
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, DB1_DIST_BACKEND="gloo", DB1_BENCH_FAIL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--layers", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "rank 1 stderr" in r.stderr and "DB1_BENCH_FAIL_RANK" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
