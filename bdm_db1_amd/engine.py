"""Training / inference engine with the surface the reference's drivers use on the DeepSpeed engine
(src/train_utils/train.py:216-232, src/evaluation/evaluate_rl.py:192,344,508-512, src/checkpointing.py:17-22):

    engine, optimizer, _, lr_scheduler = initialize(args, model, mpu=mpu)
    logits, loss = engine(inputs); engine.backward(loss); engine.step()

What the external runtime did around the model is done here MI355X-first:
  * gradients live in ONE flat float32 arena laid out in backward-completion order; as each decoder layer
    finishes its backward, its contiguous bucket (46.2 M elements for DB1-1.3B) is all-reduced with RCCL
    (torch.distributed "nccl" on ROCm) asynchronously, overlapping the remaining backward;
  * the data-parallel mean, the global-norm clip and 1/grad-accumulation are folded into the fused Adam
    kernel's gradient scale (no extra pass over 1.2 G gradients), and the clip coefficient is computed on
    the device (no host sync);
  * one fused Adam launch updates fp32 master weights + moments and writes the bf16 working copy.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops
from .optim import OptimizerParamScheduler


class GradSync:
    """Bucketed asynchronous all-reduce (SUM) over slices of one flat gradient tensor.
    Backend-agnostic (RCCL on the GPUs; gloo in the CPU tests of the sharding logic).

    With ``stage`` (a bf16 tensor of the arena's size) each bucket is first cast into its staging slice by ``cast(src, dst)`` on
    the compute stream and the STAGING slice is what travels: 2 bytes per gradient over xGMI (2.42 GB per step for DB1-1.3B, the
    volume SURVEY 8e budgets) instead of 4; the optimizer then reads the reduced staging copy (``reduced``).  torch's process group
    orders the collective behind the cast (it waits for the launching stream's work) and ``finish`` orders the optimizer behind it."""

    def __init__(self, flat_grad: torch.Tensor, buckets: List[Tuple[str, int, int]], group=None, stage: Optional[torch.Tensor] = None,
                 cast=None, norm_sq=None):
        self.flat = flat_grad
        self.buckets = {n: (s, e) for n, s, e in buckets}
        self.order = [n for n, _, _ in buckets]
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.stage = stage
        self.cast = cast
        assert stage is None or (stage.numel() == flat_grad.numel() and cast is not None)
        self.handles = []
        self.launched = set()
        # ``norm_sq(reduced_slice, out1)``: out1[0] = sum of squares of one REDUCED bucket.  Run per bucket as soon as its collective is done
        # (on a side stream ordered behind it, next to the backward that is still running), so that the global-norm clip needs only the
        # sum of ``norm_parts`` after the last bucket -- not another pass over all gradients between the last all-reduce and the optimizer.
        self.norm_sq = norm_sq
        self.norm_parts = torch.zeros(len(self.order), device=flat_grad.device, dtype=torch.float32) if norm_sq is not None else None
        # (only a backend whose wait() is a STREAM-level wait -- RCCL -- can be followed from a side stream; gloo's wait() blocks the host, which
        #  would serialise the backward behind every bucket: there the partial sums are taken in finish())
        nccl = self.world > 1 and str(dist.get_backend(group)).lower() == "nccl"
        self._side = torch.cuda.Stream(device=flat_grad.device) if (norm_sq is not None and flat_grad.is_cuda and nccl) else None
        self._pending_norm = []
        self.time_waits = False    # bench.py --gpus N: HIP-event pairs around the waits of finish() = the exposed communication
        self.wait_events = []

    @property
    def reduced(self) -> torch.Tensor:
        """the tensor that holds the all-reduced gradients after ``finish``"""
        return self.flat if (self.stage is None or self.world == 1) else self.stage

    def launch(self, name: str):
        if self.world == 1 or name in self.launched:
            return
        s, e = self.buckets[name]
        self.launched.add(name)
        buf = self.flat[s:e]
        if self.stage is not None:
            self.cast(buf, self.stage[s:e])
            buf = self.stage[s:e]
        h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.handles.append(h)
        if self.norm_sq is not None:
            i = self.order.index(name)
            if self._side is not None:   # the side stream waits for this collective (a stream-level wait: the host and the compute stream go on)
                self._side.wait_stream(torch.cuda.current_stream(self.flat.device))
                with torch.cuda.stream(self._side), ops.stream_scope():
                    h.wait()
                    self.norm_sq(buf, self.norm_parts[i:i + 1])
            else:                        # host tensors (gloo tests): a wait would block here; done in finish()
                self._pending_norm.append((i, buf))

    def reduced_norm_sq(self) -> torch.Tensor:
        """sum of squares of all reduced gradients (valid after ``finish``): the per-bucket partial sums added in bucket order"""
        return self.norm_parts.sum(dtype=torch.float32).reshape(1)

    def finish(self):
        """launch whatever was not launched by a hook, then wait for everything"""
        for n in self.order:
            self.launch(n)
        ev = None
        if self.time_waits and self.handles and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for h in self.handles:
            h.wait()
        if self._side is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
        for i, buf in self._pending_norm:
            self.norm_sq(buf, self.norm_parts[i:i + 1])
        self._pending_norm = []
        if ev is not None:
            ev[1].record()
            self.wait_events.append(ev)
        self.handles = []
        self.launched = set()


class _Optimizer:
    """What callers see as ``optimizer``: param_groups for the LR scheduler + state for checkpoints."""

    def __init__(self, lr, wd):
        self.param_groups = [{"lr": lr, "weight_decay": wd}]


class DB1Engine:
    def __init__(self, args, model, mpu=None, lr_scheduler=None):
        g = lambda k, d=None: getattr(args, k, d) if args is not None else d
        self.module = model
        self.mpu = mpu
        self.args = args
        self.group = None
        if mpu is not None and dist.is_available() and dist.is_initialized() and mpu.model_parallel_is_initialized():
            self.group = mpu.get_data_parallel_group()
        self.dp_world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        micro = g("micro_batch_size", None)
        glob = g("global_batch_size", None)
        ga = g("gradient_accumulation_steps", None)
        if ga is None:
            ga = max(1, glob // (micro * self.dp_world)) if (micro and glob) else 1
        self._ga = int(ga)
        model.loss_grad_scale = 1.0 / self._ga
        # gradient accumulation with the weight gradients formed ONCE per optimizer step (model.WgradStash: the operands of all micro-steps are
        # kept -- 57 KB per token -- and dW = dy^T x runs over K = ga * T rows on the boundary micro-step): opt-in, `defer_wgrad=True`
        # gradient accumulation with ONE backward per optimizer step (model.BackwardWindow): the micro-steps run forward (+ the fused head / loss
        # sweep, which yields each micro-step's loss and dh) into window-sized activation buffers, `backward(loss)` of the non-boundary micro-steps
        # only records, and the boundary runs the backward of the whole window as one pass over ga x B sequences -- the reference's loop
        # (train.py:216-232) needs a loss per micro-step, gradients only at step().  Opt-in, `defer_backward=True`; takes the place of defer_wgrad
        # (the window's weight gradients are K = ga * T products anyway).  Micro-steps the window cannot hold (fp32, pre-LN, a sliding
        # window, kept logits) fall back to their own backward.
        self.defer_backward = bool(g("defer_backward", False)) and self._ga > 1
        model.bwd_window_ga = self._ga if self.defer_backward else 0
        self.defer_wgrad = bool(g("defer_wgrad", False)) and self._ga > 1 and not self.defer_backward
        model.wgrad_defer_ga = self._ga if self.defer_wgrad else 0
        self.micro_steps = 0
        self.global_steps = 0
        self.beta1, self.beta2 = float(g("adam_beta1", 0.9)), float(g("adam_beta2", 0.999))
        self.eps = float(g("adam_eps", 1e-8))
        self.clip = float(g("clip_grad", 1.0) or 0.0)
        self.adamw = str(g("optimizer", "adam")).lower() == "adamw"
        lr = g("lr", None)
        self.optimizer = _Optimizer(float(lr) if lr is not None else 1e-4, float(g("weight_decay", 0.01)))
        self.lr_scheduler = lr_scheduler
        self.overlap_comm = bool(g("overlap_grad_reduce", True))
        # DeepSpeed's engine hands back logits the caller may keep; `keep_logits=False` (bench.py, memory-tight training) lets the CE backward
        # overwrite the logits buffer with dlogits in place (4.4 GB at 64 x 1024 tokens)
        model.keep_logits = bool(g("keep_logits", True))
        # without kept logits the head + loss (+ the head's backward) run as one chunked sweep and `engine(batch)` returns (None, loss)
        model.fuse_head_loss = not model.keep_logits and bool(g("fuse_head_loss", True))
        ar = model.arena
        if ar.exp_avg is None:
            ar.exp_avg = torch.zeros_like(ar.master)
            ar.exp_avg_sq = torch.zeros_like(ar.master)
        self._norm_sq = torch.zeros(1, device=model.device, dtype=torch.float32)
        self._acc_segments = model.accumulator_segments()   # the small accumulators of the gradient arena (cleared after every step)
        # gradients cross xGMI in bf16 (the reference's DeepSpeed fp16 engine reduces 2-byte gradients too: 2.42 GB per step,
        # SURVEY 8e); "fp32" keeps the 4-byte arena on the wire (bit-comparable with a single-rank run up to summation order)
        rdt = str(g("grad_reduce_dtype", "bf16")).lower()
        assert rdt in ("bf16", "fp32"), rdt
        stage = None
        if self.dp_world > 1 and rdt == "bf16":
            stage = torch.zeros(ar.numel, device=model.device, dtype=torch.bfloat16)
        # (more than one rank, clipping on: the norm of the reduced gradients is collected bucket by bucket behind each all-reduce)
        per_bucket_norm = ops.grad_norm_sq if (self.dp_world > 1 and self.clip > 0 and os.environ.get("DB1_PER_BUCKET_NORM", "1") != "0") else None
        self.sync = GradSync(ar.grad, model.grad_buckets(), self.group, stage=stage, cast=ops.cast, norm_sq=per_bucket_norm)
        self.last_grad_norm_sq = None

    # ---- what the reference's drivers call
    def __call__(self, *a, **k):
        self.module._wg_slot = self.micro_steps % self._ga
        return self.module(*a, **k)

    def __getattr__(self, name):  # attribute passthrough (model.init_mem, evaluate_rl.py:344)
        mod = self.__dict__.get("module")
        if mod is not None and hasattr(mod, name):
            return getattr(mod, name)
        raise AttributeError(name)

    @property
    def device(self):
        return self.module.device

    def train(self, mode: bool = True):
        self.module.train(mode)
        return self

    def eval(self):
        self.module.eval()
        return self

    def gradient_accumulation_steps(self):
        return self._ga

    def is_gradient_accumulation_boundary(self):
        return (self.micro_steps + 1) % self._ga == 0

    def backward(self, loss=None):
        """DeepSpeed semantics: gradient of loss / grad_accumulation_steps; on the boundary micro-step the
        per-layer buckets are all-reduced while the backward of the earlier layers is still running."""
        boundary = self.is_gradient_accumulation_boundary()
        hook = self.sync.launch if (boundary and self.overlap_comm and self.dp_world > 1) else None
        self.module.backward(grad_scale=1.0 / self._ga, layer_done_hook=hook, flush_wgrads=boundary, window_boundary=boundary)
        return loss

    def step(self):
        with torch.cuda.device(self.module.device):
            return self._step()

    def _step(self):
        boundary = self.is_gradient_accumulation_boundary()
        self.micro_steps += 1
        if not boundary:
            return
        self.sync.finish()
        ar = self.module.arena
        grad = self.sync.reduced      # the fp32 arena, or the all-reduced bf16 staging copy
        gscale = 1.0 / self.dp_world  # it holds the SUM over ranks
        if self.clip > 0 and self.sync.norm_parts is not None and self.dp_world > 1:
            self._norm_sq.copy_(self.sync.reduced_norm_sq())   # per-bucket sums taken behind each all-reduce: only this scalar sum is left
        elif self.clip > 0:
            ops.grad_norm_sq(grad, self._norm_sq)   # overwrites; fixed-order partial sums: the clip coefficient is reproducible run to run
        else:
            self._norm_sq.zero_()
        self.global_steps += 1
        grp = self.optimizer.param_groups[0]
        ops.adam_step(ar.master, grad, ar.exp_avg, ar.exp_avg_sq, None if ar.work is ar.master else ar.work,
                      grp["lr"], self.beta1, self.beta2, self.eps, grp["weight_decay"], self.adamw, self.global_steps,
                      gscale=gscale, clip=self.clip, norm_sq=self._norm_sq if self.clip > 0 else None)
        # the weight gradients (99.9 % of the arena) are not cleared: the next backward's GEMMs write them with beta = 0; only the
        # scattered accumulators (LayerNorm / bias / u, v / embedding-table gradients) are, in one launch
        ops.zero_segments(ar.grad, self._acc_segments)
        self.module._grad_fresh = True
        self.module.mark_weights_changed()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step(1)

    @property
    def time_comm(self) -> bool:
        return self.sync.time_waits

    @time_comm.setter
    def time_comm(self, on: bool):
        self.sync.time_waits = bool(on)

    def exposed_comm_ms(self) -> float:
        """total time the compute stream spent waiting for bucket all-reduces since time_comm was switched on (synchronises)"""
        torch.cuda.synchronize(self.module.device)
        ms = float(sum(a.elapsed_time(b) for a, b in self.sync.wait_events))
        self.sync.wait_events = []
        return ms

    def get_global_grad_norm(self) -> float:
        """host-synchronising convenience (not used in the step path)"""
        return float(self._norm_sq.sqrt().item()) / self.dp_world

    # ---- checkpoints (DeepSpeed file naming so the released loaders' paths make sense)
    def save_checkpoint(self, save_dir, tag=None, client_state=None):
        tag = tag or f"global_step{self.global_steps}"
        path = os.path.join(save_dir, str(tag))
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        if rank == 0:
            os.makedirs(path, exist_ok=True)
            ar = self.module.arena
            state = {"module": {k: v.detach().cpu() for k, v in self.module.state_dict().items()},
                     "optimizer": {"exp_avg": ar.exp_avg.cpu(), "exp_avg_sq": ar.exp_avg_sq.cpu(), "step": self.global_steps},
                     "lr_scheduler": self.lr_scheduler.state_dict() if self.lr_scheduler is not None else None,
                     "global_steps": self.global_steps, "micro_steps": self.micro_steps,
                     # the dropout stream (counter-based Philox on (seed, site, step)): a resumed run continues it instead of replaying the
                     # masks of the first iterations
                     "dropout": {"seed": int(self.module.dropout_seed),
                                 "step": int(self.module._drop_step_dev.item()) if self.module._drop_step_dev is not None else int(self.module._drop_step)}}
            state.update(client_state or {})
            torch.save(state, os.path.join(path, "mp_rank_00_model_states.pt"))
            with open(os.path.join(save_dir, "latest"), "w") as f:
                f.write(str(tag))
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        return True

    def load_checkpoint(self, load_dir, tag=None, load_optimizer_states=True, **_):
        if tag is None:
            with open(os.path.join(load_dir, "latest")) as f:
                tag = f.read().strip()
        path = os.path.join(load_dir, str(tag), "mp_rank_00_model_states.pt")
        state = torch.load(path, map_location="cpu", weights_only=False)
        self.module.load_state_dict(state["module"], strict=False)
        if load_optimizer_states and state.get("optimizer") and "exp_avg" in state["optimizer"]:
            ar = self.module.arena
            ar.exp_avg.copy_(state["optimizer"]["exp_avg"])
            ar.exp_avg_sq.copy_(state["optimizer"]["exp_avg_sq"])
            self.global_steps = int(state["optimizer"].get("step", 0))
        if self.lr_scheduler is not None and state.get("lr_scheduler"):
            self.lr_scheduler.load_state_dict(state["lr_scheduler"])
        # partially accumulated gradients are not part of a checkpoint: resume at the last accumulation boundary, on clean accumulators
        self.micro_steps = int(state.get("micro_steps", 0)) // self._ga * self._ga
        with torch.cuda.device(self.module.device):
            ops.zero_segments(self.module.arena.grad, self._acc_segments)
        self.module._grad_fresh = True
        self.module._ctx = None
        if getattr(self.module, "_win", None) is not None:      # (forwards recorded for a deferred backward belong to the interrupted window)
            self.module._win.ctxs, self.module._win.n = [None] * self.module._win.ga, 0
        self.sync.handles, self.sync.launched = [], set()
        if state.get("dropout"):
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            if rank == 0:   # (the other ranks keep their own seed: seed + rank at construction; the step counter is common)
                self.module.dropout_seed = int(state["dropout"]["seed"])
            self.module._drop_step = int(state["dropout"]["step"])
            if self.module._drop_step_dev is not None:   # a hipGraph-captured step is active (graphed_train.py): its device counter follows
                self.module._drop_step_dev.fill_(int(self.module._drop_step))
        client = {k: v for k, v in state.items() if k not in ("module", "optimizer", "lr_scheduler", "dropout")}
        return path, client


def init_distributed(dist_backend: str = "nccl", distributed_port: Optional[int] = None, **_):
    """``deepspeed.init_distributed`` stand-in (evaluate_rl.py:492): one process per GPU, RCCL over xGMI.

    The RCCL communicator runs its kernels on a HIGH-PRIORITY stream: a bucket's all-reduce is a few hundred microseconds of
    xGMI traffic that must not queue behind the seconds of GEMM work already enqueued on the compute stream; bound to the GPU
    at creation (``device_id``) so the communicator is built eagerly, once, before the first step."""
    if dist.is_available() and dist.is_initialized():
        return
    if "RANK" not in os.environ:
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL's intra-node transport on this driver)
    if distributed_port is not None:
        os.environ.setdefault("MASTER_PORT", str(distributed_port))
    if dist_backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group(backend="nccl", pg_options=opts, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend=dist_backend)


def initialize(args=None, model=None, mpu=None, lr_scheduler=None, **_):
    """``deepspeed.initialize`` stand-in: returns the same 4-tuple shape (engine, optimizer, dataloader, lr_scheduler)."""
    engine = DB1Engine(args, model, mpu=mpu, lr_scheduler=None)
    g = lambda k, d=None: getattr(args, k, d) if args is not None else d
    if lr_scheduler is None and g("lr", None) is not None and g("lr_decay_iters", None):
        lr_scheduler = OptimizerParamScheduler(
            engine.optimizer, max_lr=g("lr"), min_lr=g("min_lr", 0.0) or 0.0,
            lr_warmup_steps=g("lr_warmup_iters", 0) or 0, lr_decay_steps=g("lr_decay_iters"),
            lr_decay_style=g("lr_decay_style", "linear"), start_wd=g("start_weight_decay", g("weight_decay", 0.01)),
            end_wd=g("end_weight_decay", g("weight_decay", 0.01)), wd_incr_steps=g("lr_decay_iters"),
            wd_incr_style=g("weight_decay_incr_style", "constant"))
    elif lr_scheduler is not None and hasattr(lr_scheduler, "optimizer"):
        lr_scheduler.optimizer = engine.optimizer
        lr_scheduler.step(0)
    engine.lr_scheduler = lr_scheduler
    return engine, engine.optimizer, None, lr_scheduler
