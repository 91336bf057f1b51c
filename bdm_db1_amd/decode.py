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
