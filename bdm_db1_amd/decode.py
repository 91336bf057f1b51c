"""Launch-bound inference with memory as ONE hipGraph replay per call.

A DB1-1.3B call with a full memory (evaluate_rl.py:157-266: batch 1, 1 .. ~50 new tokens, mem_len 1024) is ~340 short
kernels; launched one by one from Python the call is host-bound (4.8 ms wall for 3.2 ms of kernels).  ``GraphedMemoryStep``
captures the model's own forward for a fixed (batch, new-token count) once and replays it: same kernels, same results,
no per-kernel launch cost.  It is opt-in because a graph works on static buffers: the memory it returns is always the SAME
list of tensors, updated in place by the replay (the reference returns fresh tensors every call); callers that only pass the
memory back in -- like evaluate_rl's loop -- see no difference.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List

import torch


class GraphedMemoryStep:
    def __init__(self, model, batch_size: int, n_new: int, make_input=None):
        """``make_input(ids)`` builds the task input for static token ids [batch, n_new] (default: a text input)."""
        from .data import NLPTaskInput
        if model.compute_dtype != torch.bfloat16 or not model.use_decode:
            raise ValueError("GraphedMemoryStep needs the bf16 K/V-cached decode path (model.use_decode)")
        self.model, self.B, self.q = model, batch_size, n_new
        dev = model.dev
        self.ids = torch.zeros(batch_size, n_new, dtype=torch.long, device=dev)
        make_input = make_input or (lambda ids: NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None,
                                                             text_seq=ids, text_len=None))
        self.x = make_input(self.ids)
        self.mems: List[torch.Tensor] = model.init_mem(batch_size)
        self.graph = None
        self.logits = None
        # eager warm-up on the static buffers: allocates every workspace, builds the R table and the K/V cache of self.mems
        with torch.no_grad():
            for _ in range(2):
                _, _, m = model([self.x], compute_loss=False, mems=self.mems)
                st = model._dec_state
                for dst, src in zip(self.mems, m):
                    dst.copy_(src)
                self.kv = [k.contiguous().clone() for k in st.kv]
                model._dec_state = model._dec_pack(self.mems, self.kv)
        self._version = model._wversion
        torch.cuda.synchronize()
        self._capture()
        self.reset_memory()  # the warm-up calls advanced the memory

    def _capture(self):
        model = self.model
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            logits, _, m = model([self.x], compute_loss=False, mems=self.mems)
            new_kv = model._dec_state.kv
            for dst, src in zip(self.mems, m):
                dst.copy_(src)
            for dst, src in zip(self.kv, new_kv):
                dst.copy_(src)
        self.graph, self.logits = g, logits
        model._dec_state = model._dec_pack(self.mems, self.kv)

    def reset_memory(self):
        """start a new episode: zero memory (init_mem) and the matching K/V cache (rebuilt from the zero hidden states)"""
        model = self.model
        with torch.no_grad():
            for dst, src in zip(self.mems, model.init_mem(self.B)):
                dst.copy_(src)
            model._dec_state = None
            dec = model._decode_begin(self.mems, self.B, self.q, self.mems[0].shape[1])
            for dst, src in zip(self.kv, dec.kv):
                dst.copy_(src)
            model._dec_state = model._dec_pack(self.mems, self.kv)

    def __call__(self, ids: torch.Tensor):
        """ids [batch, n_new] -> (logits [batch, n_new, vocab] (static buffer, overwritten by the next call), memory list)"""
        if self.model._wversion != self._version:
            raise RuntimeError("the weights changed after the graph was captured: build a new GraphedMemoryStep")
        self.ids.copy_(ids)
        self.graph.replay()
        return self.logits, self.mems


class RingMemory:
    """Transformer-XL memory for hipGraph-replayed inference: per layer the projected keys / values of the last ``mem_len`` tokens in a
    ring [B, mem_len + 64, 2, H, D], appended IN PLACE by the attention launch of each call (db1_relattn_decode_ring_fwd), and a device
    scalar with the ring's origin.  Pass it as ``mems`` (``model(x, compute_loss=False, mems=ring)`` returns it back as the new memory).
    What it does not keep is the reference's memory CONTENT -- the hidden states (transformer_xl.py:470-504): a caller that reads or edits
    ``mems[i]`` needs the list form (``model.init_mem``); evaluate_rl's loop only hands the memory back to the model."""

    def __init__(self, model, batch_size: int):
        if model.compute_dtype != torch.bfloat16 or not model.use_decode or model.d_head != 128 or not model.mem_len:
            raise ValueError("RingMemory needs the bf16 K/V-cached decode path (d_head 128, mem_len > 0)")
        self.model, self.B = model, batch_size
        self.cap = int(model.mem_len) + 64
        dev = model.dev
        self.kv = [torch.zeros(batch_size, self.cap, 2, model.n_head, model.d_head, device=dev, dtype=torch.bfloat16) for _ in range(model.n_layer)]
        self.state = torch.zeros(1, dtype=torch.int32, device=dev)
        self.reset()

    def reset(self):
        """a new episode: the keys / values of the zero memory (init_mem), origin 0"""
        model, mlen = self.model, int(self.model.mem_len)
        with torch.no_grad():
            saved, model._dec_state = model._dec_state, None
            dec = model._decode_begin(model.init_mem(self.B), self.B, 1, mlen)
            model._dec_state = saved
            for ring, kv in zip(self.kv, dec.kv):
                ring[:, :mlen].copy_(kv.reshape(self.B, mlen, 2, model.n_head, model.d_head))
            self.state.zero_()


class GraphedRingStep:
    """One inference call with memory (batch ``batch_size``, ``n_new`` new tokens) as ONE hipGraph replay over a RingMemory: nothing is
    concatenated, copied or re-projected per call.  Several steps (e.g. the observation call and the 1-token calls of evaluate_rl) can
    share one memory: ``GraphedRingStep(model, 1, 1, memory=obs_step.memory)``."""

    def __init__(self, model, batch_size: int, n_new: int, memory: RingMemory = None, make_input=None):
        from .data import NLPTaskInput
        self.model, self.B, self.q = model, batch_size, n_new
        self.memory = memory if memory is not None else RingMemory(model, batch_size)
        assert self.memory.B == batch_size
        dev = model.dev
        self.ids = torch.zeros(batch_size, n_new, dtype=torch.long, device=dev)
        make_input = make_input or (lambda ids: NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None))
        self.x = make_input(self.ids)
        from . import ops
        saved_state = self.memory.state.clone()
        saved_kv = None if memory is None else [k.clone() for k in self.memory.kv]
        # the persistent one-token launches of THIS step hand over through a scratch of its own, with its own pinned copy of the error
        # flag -- both created here, eagerly: a scratch first touched under capture would be allocated and zero-filled by a graph node, i.e.
        # every replay would wipe the sticky flag before anyone read it (and all steps captured on torch's capture stream would share it)
        self._scratch, self._flag_host = ops.new_chain_scratch(dev)
        pending = model._chain_watch       # (the model's own watch of its eager calls stays what it was)
        with torch.no_grad():   # eager warm-up (workspaces, the R table), then the capture
            with ops.chain_scratch_scope(self._scratch, self._flag_host):
                for _ in range(2):
                    model([self.x], compute_loss=False, mems=self.memory)
                torch.cuda.synchronize()
                model._chain_watch = None
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    logits, _, _ = model([self.x], compute_loss=False, mems=self.memory)
                watch = model._chain_watch
        self.graph, self.logits = g, logits
        self._watch = watch            # the captured call's copy of the persistent launches' error flag (None: per-launch path)
        model._chain_watch = pending
        self._version = model._wversion
        if saved_kv is None:
            self.memory.reset()
        else:   # a shared memory keeps the state its owner left
            for dst, src in zip(self.memory.kv, saved_kv):
                dst.copy_(src)
            self.memory.state.copy_(saved_state)

    def reset_memory(self):
        self.memory.reset()

    def __call__(self, ids: torch.Tensor):
        """ids [batch, n_new] (or None / ``self.ids`` itself: the tokens are already in the static input buffer) ->
        (logits [batch, n_new, vocab] (static buffer, overwritten by the next call), the RingMemory)"""
        if self.model._wversion != self._version:
            raise RuntimeError("the weights changed after the graph was captured (or a persistent launch of this graph failed): build a new GraphedRingStep")
        self.check()   # the replays the stream has finished (free: a pinned host word)
        if ids is not None and ids.data_ptr() != self.ids.data_ptr():   # (a sampler that writes the next token into ``self.ids`` skips this copy)
            self.ids.copy_(ids)
        self.graph.replay()
        return self.logits, self.memory

    def check(self, synchronize: bool = False):
        """raise if a replayed persistent launch reported a failed hand-off (TransformerXL.check_decode_chain): call with
        ``synchronize=True`` before trusting logits that were not read back through a synchronising copy"""
        if self._watch is not None:
            try:
                self.model.check_decode_chain(synchronize, watch=self._watch)   # (this step's own flag copy; the model's pending watch is not touched)
            except Exception:
                # the captured graph still contains the persistent launches: this step must not be replayed again (the model has switched the
                # chain off, so a NEW GraphedRingStep captures the per-launch path)
                self._version = -1
                self._watch = None
                raise
