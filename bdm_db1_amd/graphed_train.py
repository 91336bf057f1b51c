"""Forward + backward of ONE training micro-step as a hipGraph replay.

At the reference's own batch geometry -- micro-batch 4, gradient accumulation 16 (scripts/evaluate/evaluate_rl_1.2B.sh:28-42,
src/train_utils/train.py:216-232) -- a micro-step is ~1040 short launches for ~28 ms of GPU work, and issuing them from Python / ctypes takes
~44 ms: the step is host-bound (profiles/r03a_bench_b4.json: 92.9 k tokens/s against 149.1 k with 64 sequences per micro-step).  The library is
capture-legal by construction (no allocation / synchronisation behind any entry point, scratch from the caller), so the micro-step's
forward + backward is captured once per (batch shape, first / accumulating) and replayed: same kernels, same results, one launch.

What makes it legal to replay:
  * inputs live in static tensors (the batch of every call is copied into them);
  * dropout: the keep decisions are a counter-based function of (seed, site, STEP); the step is read from a device counter the graph
    itself bumps (``drop_step_dev`` of include/db1_hip.h), so every replay draws new masks -- the same stream of masks as the eager engine
    (step 1, 2, 3, ...);
  * gradient accumulation: the first micro-step after an optimizer step WRITES the weight gradients (beta = 0), later ones accumulate:
    two graphs;
  * per-weight-version caches (permuted convolution weights) are rebuilt inside the graph, into static buffers;
  * the optimizer step, the data-parallel all-reduce and its hooks stay outside: with more than one rank the boundary micro-step (the one
    that launches the bucket all-reduces from inside the backward) runs eagerly.
"""
from __future__ import annotations

import copy
import dataclasses
from typing import List, Sequence

import torch


def _tensor_fields(task):
    if dataclasses.is_dataclass(task):
        names = [f.name for f in dataclasses.fields(task)]
    else:
        names = [k for k in vars(task)]
    return [n for n in names if torch.is_tensor(getattr(task, n, None))]


class GraphedTrainStep:
    def __init__(self, engine, example_batch: Sequence):
        """``engine``: bdm_db1_amd.engine.DB1Engine in train() mode with keep_logits=False (the fused head + loss sweep);
        ``example_batch``: a list of task inputs with the shapes / dtypes every later batch will have"""
        model = engine.module
        if not model.training or model.keep_logits or not model.fuse_head_loss:
            raise ValueError("GraphedTrainStep captures the training micro-step with the fused head + loss (engine.train(), keep_logits=False)")
        if engine.micro_steps % engine.gradient_accumulation_steps() != 0:
            raise ValueError("GraphedTrainStep must be built on a gradient-accumulation boundary: its warm-up clears the gradient accumulators")
        self.engine, self.model = engine, model
        dev = model.dev
        self.static: List = []
        for t in example_batch:
            st = copy.copy(t)
            for n in _tensor_fields(t):
                setattr(st, n, getattr(t, n).to(dev).clone())
            self.static.append(st)
        # vision position ids: in training mode the reference draws them per step on the host (vision_embedding.py:150-169) -- a host-side
        # draw + copy cannot be captured, so a task that does not carry them gets static device tensors the graph reads, refilled with a
        # fresh draw before every replay
        self._vis = []
        for st in self.static:
            geo = self._vision_geometry(st)
            if geo is not None and getattr(st, "vision_row_ids", None) is None:
                r, c = model._vision_position_ids(geo[1], geo[2], geo[0])
                st.vision_row_ids, st.vision_col_ids = r.to(dev), c.to(dev)
                self._vis.append((st, geo))
        self._fields = [_tensor_fields(t) for t in self.static]
        # the dropout step counter moves to the device; it continues the eager engine's count
        self.step_dev = torch.full((1,), int(model._drop_step), dtype=torch.int32, device=dev)
        model._drop_step_dev = self.step_dev
        model._graph_static = True
        self.graphs = {}
        self.loss = {}
        # eager warm-up on the static inputs (allocates every workspace / cached table, decides the attention-backward mode), then the
        # accumulators it touched are cleared again
        self.window = bool(getattr(engine, "defer_backward", False)) and model.bwd_window_ga > 1
        ga = engine.gradient_accumulation_steps()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if self.window:      # one whole window: allocates the window's buffers (outside any graph pool) and every workspace of the big backward
                for k in range(ga):
                    self.step_dev.add_(1)
                    model._wg_slot = k
                    _, loss = model(self.static)
                    model.backward(grad_scale=model.loss_grad_scale, layer_done_hook=None, window_boundary=(k == ga - 1))
                if model._win is None or model._win.sig is None:
                    self.window = False      # (this configuration does not take the window: every micro-step keeps its own backward)
            for _ in range(0 if self.window else 2):
                self.step_dev.add_(1)
                model._wg_slot = 0
                _, loss = model(self.static)
                model.backward(grad_scale=model.loss_grad_scale, layer_done_hook=None, flush_wgrads=False)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        model.zero_grad()
        self.step_dev.fill_(int(model._drop_step))
        # one graph per (first / accumulating micro-step) -- and, with deferred weight gradients (engine option defer_wgrad), per micro-step
        # of the accumulation window: each writes its operands into its own rows of the stash.  All graphs share one memory pool (they never
        # run at the same time), so their intermediate buffers cost what one micro-step costs.
        self.defer = bool(getattr(engine, "defer_wgrad", False)) and model.wgrad_defer_ga > 1
        self.pool = torch.cuda.graph_pool_handle()
        self.win_ctx = {}
        if self.window:
            # deferred backward: one graph per micro-step of the window, FORWARD only (each writes its own row block of the window's buffers);
            # the backward of the whole window runs eagerly on the boundary (so the bucket all-reduces of more than one rank start from inside
            # it as usual).  Every graph has its OWN pool: what a forward leaves for the boundary outside the window's buffers (the embedding
            # contexts: ids, the image-patch embedder's activations) must survive the replays of the other micro-steps.
            for slot in range(ga):
                model._grad_fresh = slot == 0
                model._wg_slot = slot
                model._win.n = slot
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.step_dev.add_(1)
                    _, loss = model(self.static)
                self.graphs[(slot == 0, slot)], self.loss[(slot == 0, slot)] = g, loss
                self.win_ctx[slot] = model._win.ctxs[slot]
                model._ctx = None
            model._win.ctxs, model._win.n = [None] * ga, 0
        slots = () if self.window else (range(ga) if self.defer else (0,))
        for slot in slots:
            for fresh in ((True,) if (self.defer and slot == 0) else ((False,) if self.defer else (True, False))):
                model._grad_fresh = fresh
                model._wg_slot = slot
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool):
                    self.step_dev.add_(1)
                    _, loss = model(self.static)
                    model.backward(grad_scale=model.loss_grad_scale, layer_done_hook=None, flush_wgrads=False)
                self.graphs[(fresh, slot)], self.loss[(fresh, slot)] = g, loss
        model._grad_fresh = True
        model._ctx = None

    def _vision_geometry(self, t):
        img = getattr(t, "vision_seq", None)
        if img is None:
            img = getattr(t, "img_seq", None)
        if img is None or not torch.is_tensor(img) or img.dim() < 3:
            return None
        p = self.model.patch_size
        n_img = 1
        for s_ in img.shape[:-3]:
            n_img *= int(s_)
        return n_img, img.shape[-2] // p, img.shape[-1] // p

    def __call__(self, batch: Sequence):
        """one micro-step: returns the loss (a static device scalar, overwritten by the next call); follow it with ``engine.step()``"""
        eng, model = self.engine, self.model
        assert len(batch) == len(self.static)
        for st, t, names in zip(self.static, batch, self._fields):
            for n in names:
                src = getattr(t, n, None)
                dst = getattr(st, n)
                if src is None:          # (the position ids this object supplies itself, below)
                    continue
                if src.data_ptr() != dst.data_ptr():
                    dst.copy_(src, non_blocking=True)
        for st, geo in self._vis:
            if not any(getattr(t, "vision_row_ids", None) is not None for t in batch if self._vision_geometry(t) == geo):
                r, c = model._vision_position_ids(geo[1], geo[2], geo[0])
                st.vision_row_ids.copy_(r, non_blocking=True)
                st.vision_col_ids.copy_(c, non_blocking=True)
        boundary = eng.is_gradient_accumulation_boundary()
        if eng.dp_world > 1 and boundary and not self.window:
            # the bucket all-reduces are launched from inside this backward: eager, on the same device counter
            self.step_dev.add_(1)
            model._wg_slot = eng.micro_steps % eng.gradient_accumulation_steps()
            _, loss = model(self.static)
            eng.backward(loss)
            model._drop_step += 1      # host mirror of the device counter (what engine.save_checkpoint stores)
            return loss
        fresh = bool(model._grad_fresh)
        if self.window:
            slot = eng.micro_steps % eng.gradient_accumulation_steps()
            key = (slot == 0, slot)
            win = model._win
            if key not in self.graphs or win.n != slot or (slot == 0 and not fresh):
                raise RuntimeError(f"GraphedTrainStep (deferred backward): micro-step {slot} of the window with {win.n} forwards recorded and fresh={fresh}: "
                                   "the first micro-step of a window must follow an optimizer step (or zero_grad), the others must follow it in order")
            self.graphs[key].replay()
            win.ctxs[slot], win.n = self.win_ctx[slot], slot + 1
            model._drop_step += 1
            if boundary:
                hook = eng.sync.launch if (eng.overlap_comm and eng.dp_world > 1) else None
                with torch.cuda.device(model.dev):
                    from . import ops
                    with ops.stream_scope():
                        model._backward_window(model.loss_grad_scale, hook)
            return self.loss[key]
        slot = eng.micro_steps % eng.gradient_accumulation_steps() if self.defer else 0
        key = (fresh, slot)
        if key not in self.graphs:
            raise RuntimeError(f"GraphedTrainStep: no captured graph for micro-step {slot} of the window with fresh={fresh}: with deferred weight gradients the "
                               "first micro-step of a window must follow an optimizer step (or zero_grad) and the others must not")
        if self.defer:
            st = model.wgrad_stash
            st.slot = slot
            if slot == 0:
                st.beta = 0.0 if fresh else 1.0
        self.graphs[key].replay()
        model._grad_fresh = False
        model._drop_step += 1          # (host mirror of the device counter: checkpoints / a later switch back to eager steps)
        if self.defer and boundary:    # the stashed weight gradients of the whole window: K = ga * T, outside the graph
            model.flush_deferred_wgrads()
        return self.loss[key]

    def resync_dropout_step(self):
        """after engine.load_checkpoint(): the device counter follows the restored host counter"""
        self.step_dev.fill_(int(self.model._drop_step))

    def close(self):
        """back to eager steps: the dropout counter returns to the host"""
        self.model._drop_step = int(self.step_dev.item())
        self.model._drop_step_dev = None
        self.model._graph_static = False
