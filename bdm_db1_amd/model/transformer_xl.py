"""MI355X-native TransformerXL for DB1: the reference's ``src/model/transformer_xl.py`` API
(constructor attributes, ``forward(tasks_input, compute_loss, mems)``, ``init_mem``, state-dict names)
over hand-written gfx950 kernels called through the C ABI (``include/db1_hip.h``).

Design (MI355X-first, not a translation of the eager reference):
  * one flat float32 parameter arena (+ Adam m/v, + float32 gradient arena laid out in
    BACKWARD-COMPLETION order so per-layer all-reduce buckets are contiguous), and a bf16 working copy
    written by the fused Adam kernel; ``nn.Parameter``s are views into the master arena so
    ``state_dict()`` / ``load_state_dict()`` keep the reference's names and shapes;
  * forward and backward are explicit sequences of kernel launches on the current HIP stream
    (no autograd graph); activations are kept (288 GB of HBM: no recompute at DB1-1.3B sizes);
  * the relative-position term uses the closed form score[i,j] = ((q_i+u).k_j + (q_i+v).R[i-j])/sqrt(d)
    (transformer_xl.py:98-110,160-173); the attention mask is the index predicate
    ``i - shift < j <= i + mlen`` (:551-567) and is never materialised;
  * the vocabulary is padded to a multiple of 128 rows INSIDE the arena (33 025 -> 33 280); the padded
    logits columns never enter the softmax.
There is no CPU or torch-op fallback: without libdb1_hip.so or without a gfx950 device this raises.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import lib, ops


def print_with_rank(message):
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        print(f"rank: {torch.distributed.get_rank()}", message, flush=True)
    else:
        print(message, flush=True)


class _Node(nn.Module):
    """Anonymous container: only there to reproduce the reference's dotted parameter names."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("container module")


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class ParamArena:
    """Flat device storage for parameters, gradients and optimizer state."""

    def __init__(self, entries: List[Tuple[str, Tuple[int, ...], int]], device, work_dtype):
        # entries: (name, logical shape, allocated element count >= prod(shape))
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...], int]] = {}
        off = 0
        for name, shape, alloc in entries:
            self.offsets[name] = (off, shape, alloc)
            off += _round_up(alloc, 8)
        self.numel = _round_up(off, 8)
        self.master = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.work = self.master if work_dtype == torch.float32 else torch.zeros(self.numel, device=device, dtype=work_dtype)
        self.work_dtype = work_dtype
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        self._views: Dict[int, Tuple[torch.Tensor, dict]] = {}

    def view(self, buf: torch.Tensor, name: str, full: bool = False) -> torch.Tensor:
        # (views are cached: building one costs ~10 us of host time, and an inference layer asks for a dozen per call.  The cache lives on
        # the ARENA, keyed by the buffer's identity -- never on the tensor itself: a view holds its base, so tensor -> dict -> view -> base
        # is a cycle through C++ that Python's collector cannot see, and the buffer would outlive the model.  A buffer that is no longer one
        # of the arena's (reallocated: dtype / device move, a rebuilt optimizer state) loses its entry the next time a new buffer shows up,
        # so nothing here keeps a replaced buffer alive, and everything dies with the arena.)
        ent = self._views.get(id(buf))
        if ent is None or ent[0] is not buf:
            live = {id(t) for t in self.__dict__.values() if isinstance(t, torch.Tensor)}
            for k in [k for k in self._views if k not in live]:
                del self._views[k]
            ent = (buf, {})
            self._views[id(buf)] = ent
        cache = ent[1]
        key = (name, full)
        v = cache.get(key)
        if v is None:
            off, shape, alloc = self.offsets[name]
            v = buf[off:off + alloc] if full else buf[off:off + int(np.prod(shape))].view(*shape)
            cache[key] = v
        return v

    def sync_work(self):
        if self.work is not self.master:
            ops.cast(self.master, self.work)


class WgradStash:
    """Operands of the weight-gradient products of every decoder layer, kept for ALL micro-steps of an optimizer step (gradient
    accumulation): for each of the four big linear maps of a layer its input x and its output gradient dy, [ga * T, width] each, micro-step
    k in rows [k T, (k + 1) T).  The weight gradients are then formed ONCE per optimizer step, dW = dy^T x over K = ga * T rows, instead of
    ga times over K = T with a read-modify-write of the fp32 accumulator each time: at the reference's geometry (micro-batch 4 x 1024 tokens,
    16 micro-steps, scripts/evaluate/evaluate_rl_1.2B.sh:28-42) those K = 4096 products are a quarter of the step and run at half the rate of
    the K = 65 536 ones.  57 KB per token: 90 GB of the 288 at 16 x 4096 tokens -- memory this part has and the reference's GPUs did not."""
    KINDS = ("qkv", "o", "ff1", "ff2")

    def __init__(self, model, T: int, ga: int, nd: int = 0):
        d, di, dff = model.d_model, model.d_inner, model.d_ff
        self.T, self.ga, self.n_layer, self.nd = T, ga, model.n_layer, nd
        # the fifth map, r_net (R = r_net(position table), transformer_xl.py:138): its input is the position table of the micro-step (ONE
        # table for all layers: nd rows -- the embedding dropout redraws it every micro-step), its output gradient dR [nd, d] per layer.
        # 12 ms of K = 1024 products per optimizer step at 16 x 4096 tokens become 24 products over K = ga * nd rows.
        self.rin = torch.empty(ga * nd, d, device=model.dev, dtype=model.compute_dtype) if nd else None
        self.dr = [torch.empty(ga * nd, d, device=model.dev, dtype=model.compute_dtype) for _ in range(model.n_layer)] if nd else None
        xw = {"qkv": d, "o": d, "ff1": d, "ff2": dff}
        yw = {"qkv": 3 * d, "o": d, "ff1": di, "ff2": d}
        new = lambda w: torch.empty(ga * T, w, device=model.dev, dtype=model.compute_dtype)
        self.x = [{k: new(xw[k]) for k in self.KINDS} for _ in range(model.n_layer)]
        self.dy = [{k: new(yw[k]) for k in self.KINDS} for _ in range(model.n_layer)]
        self.slot = 0            # micro-step of the accumulation window the next forward / backward belongs to (set by the engine)
        self.first = 0           # first micro-step of the window whose operands are in THIS stash (> 0: the token count changed mid-window)
        self.r_used = False      # the forwards of this window put their position tables into rin (False: r_net's gradient is formed per micro-step)
        self.parts, self.ln_parts, self.uv_parts, self.b1_parts = False, [], [], []       # alloc_parts
        self.beta = 0.0          # beta of the flush: 0 when the gradient arena was fresh at slot 0

    def nbytes(self) -> int:
        n = sum(t.numel() * t.element_size() for L in (self.x, self.dy) for dct in L for t in dct.values())
        n += sum(t.numel() * t.element_size() for t in [self.rin] + self.dr) if self.nd else 0
        return n + (sum(t.numel() * 4 for L in (self.ln_parts, self.uv_parts, self.b1_parts) for e in L for t in (e if isinstance(e, list) else [e])) if self.parts else 0)

    def parts_plan(self, model) -> int:
        """floats of a LayerNorm partial-sum block if EVERY producer of a small reduction has its partials-only form under the model's
        CURRENT switches (bf16, register-resident LayerNorm width, fused GEGLU backward, flash + dq_r stream), else 0"""
        T, d, dff, H, D = self.T, model.d_model, model.d_ff, model.n_head, model.d_head
        dt = model.compute_dtype
        nln = ops.layernorm_bwd_parts_numel(T, d, dt) if dt == torch.bfloat16 else 0
        ok = (model.use_partial_stash and nln > 0 and nln % (2 * d) == 0 and model.activation_fn == "geglu" and model.use_geglu_epilogue and not model.untie_r and
              model.use_flash and model.use_flash_bwd and model.dropattn == 0 and self.nd and
              ops.relattn_flash_supported(T // self.nd, self.nd, H, D, dt) and
              ops.gemm_nn_geglu_bwd_fused(T, dff, d, dt) and T % 128 == 0 and self.nd and ops.relattn_dqr_supported(T // self.nd, self.nd, H, D, dt))
        return nln if ok else 0

    def alloc_parts(self, model) -> bool:
        """the small reductions of a micro-step's backward -- LayerNorm parameter gradients (two per layer), the u / v column sums of the dq_r
        stream, the first feed-forward bias's column sums -- leave their PARTIAL sums here, per micro-step, and are added up once per layer at
        the flush (db1_colsum_acc over all micro-steps' rows): ~120 launches of 5 us less per micro-step.  Only where every producer has its
        partials-only form (parts_plan).  Re-decided at the first micro-step of every window (revalidate_parts): a switch flipped while the
        stash is alive (a test or an A/B toggling use_flash_bwd / use_geglu_epilogue ...) drops or rebuilds the partial buffers instead of
        failing inside a backward."""
        ga, d, dff, H, D, T = self.ga, model.d_model, model.d_ff, model.n_head, model.d_head, self.T
        nln = self.parts_plan(model)
        if not nln:
            self.parts, self.ln_parts, self.uv_parts, self.b1_parts = False, [], [], []
            return False
        new = lambda *shape: torch.empty(*shape, device=model.dev, dtype=torch.float32)
        self.ln_parts = [[new(ga, nln // (2 * d), 2 * d) for _ in range(2)] for _ in range(self.n_layer)]       # [layer][0: pos_ff LN, 1: dec_attn LN]
        self.uv_parts = [new(ga, 2, ops.relattn_dqr_parts_rows(H), H * D) for _ in range(self.n_layer)]
        self.b1_parts = [new(ga, T // 128, 2 * dff) for _ in range(self.n_layer)]
        self.parts = True
        return True

    def revalidate_parts(self, model):
        """slot 0 of a window: the partial-sum plan against the model's switches as they are NOW"""
        nln = self.parts_plan(model) if self.T == (self.T // self.nd) * self.nd and self.nd else 0
        have = self.ln_parts[0][0].shape[1] * self.ln_parts[0][0].shape[2] if self.parts else 0
        if nln != have:
            self.alloc_parts(model)

    def rins(self) -> torch.Tensor:
        return self.rin[self.slot * self.nd:(self.slot + 1) * self.nd]

    def drs(self, i: int) -> torch.Tensor:
        return self.dr[i][self.slot * self.nd:(self.slot + 1) * self.nd]

    def xs(self, i: int, kind: str) -> torch.Tensor:
        return self.x[i][kind][self.slot * self.T:(self.slot + 1) * self.T]

    def dys(self, i: int, kind: str) -> torch.Tensor:
        return self.dy[i][kind][self.slot * self.T:(self.slot + 1) * self.T]


class BackwardWindow:
    """Activations of ALL micro-steps of a gradient-accumulation window, for ONE backward at the window's boundary (engine option
    ``defer_backward``).  The reference's loop (src/train_utils/train.py:216-232: ``loss = engine(x); engine.backward(loss); engine.step()`` per
    micro-step) needs a LOSS per micro-step, not a backward per micro-step: gradients are only observable at ``step()``.  So the forwards of a
    window write what the backward needs into row block [m n0, (m + 1) n0) of window-sized buffers (every kept tensor has its batch / token
    dimension first), and the boundary runs the backward once over ga x B sequences -- at the reference's 4 x 16 geometry the same launches as
    the 64-sequence step instead of 16 x (one 256 x 128 tile per CU, per-workgroup fixed costs over 4 sequences).  Memory: what a 64-sequence
    step keeps (this part's 288 GB hold it; the reference's GPUs did not)."""

    def __init__(self, ga: int):
        self.ga = int(ga)
        self.bufs: Dict[tuple, torch.Tensor] = {}
        self.ctxs: List[Optional["_Ctx"]] = [None] * self.ga
        self.n = 0               # forwards of the current window recorded so far
        self.slot = 0            # row block the running forward writes
        self.sig = None          # (B, L, nd) of the micro-steps in the buffers

    def take(self, model, key: tuple, shape: Tuple[int, ...], dtype) -> torch.Tensor:
        buf = self.bufs.get(key)
        n0 = int(shape[0])
        if buf is None:
            buf = torch.empty((self.ga * n0,) + tuple(int(x) for x in shape[1:]), device=model.dev, dtype=dtype)
            self.bufs[key] = buf
        if buf.shape[0] != self.ga * n0 or tuple(buf.shape[1:]) != tuple(shape[1:]) or buf.dtype != dtype:
            raise RuntimeError(f"backward window: buffer {key} was built for another shape ({tuple(buf.shape)} vs {self.ga} x {tuple(shape)})")
        return buf[self.slot * n0:(self.slot + 1) * n0]

    def full(self, key: tuple, n: int) -> torch.Tensor:
        """rows of the first n micro-steps"""
        buf = self.bufs[key]
        return buf[:n * (buf.shape[0] // self.ga)]

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())

    def reset(self):
        self.bufs, self.ctxs, self.n, self.slot, self.sig = {}, [None] * self.ga, 0, 0, None


class _PendingLN(SimpleNamespace):
    """a layer output whose closing residual LayerNorm has not been applied yet: LN(alpha * res + y) * gamma + beta (inference path)"""


class _Ctx:
    pass


class TransformerXL(nn.Module):
    """Drop-in for ``src.model.TransformerXL`` (transformer_xl.py:356-748)."""

    def __init__(self, config, device: Optional[torch.device] = None, compute_dtype: Optional[torch.dtype] = None):
        super().__init__()
        if not torch.cuda.is_available():
            raise lib.Db1Error("bdm_db1_amd.TransformerXL needs an MI355X (gfx950) device; there is no CPU path")
        lib.load()
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if lib.load().db1_device_is_gfx950() != 1:
            raise lib.Db1Error("libdb1_hip.so is built for gfx950 only")
        g = lambda k, dflt=None: getattr(config, k, dflt)
        # ---- the attributes the reference constructor reads (transformer_xl.py:357-439)
        self.n_embed = config.n_embed
        self.n_position = config.n_position
        self.n_layer = config.n_layer
        self.n_head = config.n_head
        self.d_head = self.n_embed // self.n_head
        assert self.d_head * self.n_head == self.n_embed, (self.d_head, self.n_head, self.n_embed)
        self.d_model = self.n_embed
        self.d_inner = 4 * self.d_model if g("n_inner") is None else config.n_inner
        self.pre_lnorm = bool(config.pre_lnorm)
        self.mem_len = config.mem_len if g("mem_len") is not None else 0
        self.same_length = bool(config.same_length)
        self.clamp_len = self.n_position
        self.untie_r = bool(config.untie_r)
        self.text_vocab_size = config.text_vocab_size
        self.discrete_vocab_size = config.num_discrete_values
        self.continuous_vocab_size = config.num_continuous_bin
        self.discrete_overlap_with_text = bool(config.overlap_with_text)
        tv = self.text_vocab_size + self.continuous_vocab_size + (0 if self.discrete_overlap_with_text else self.discrete_vocab_size)
        self.total_vocab_size = tv + 1
        self.rl_separator_token_id = tv
        self.activation_fn = config.activation_fn
        if self.activation_fn not in ("geglu", "gelu", "relu"):
            raise NotImplementedError(f"activation_fn={self.activation_fn!r}: the HIP path implements geglu / gelu / relu")
        if self.activation_fn == "geglu":
            assert self.d_inner % 2 == 0
        self.d_ff = self.d_inner // 2 if self.activation_fn == "geglu" else self.d_inner
        self.layer_norm_epsilon = float(config.layer_norm_epsilon)
        self.share_input_output_embedding = bool(config.share_input_output_embedding)
        self.use_deepnorm = bool(g("use_deepnorm", False))
        self.deepnorm_alpha = (2 * self.n_layer) ** 0.25 if self.use_deepnorm else None
        self.deepnorm_beta = (8 * self.n_layer) ** -0.25 if self.use_deepnorm else None
        # dropout (training mode only): embeddings + position table (embd_pdrop, :409,545,575), attention and feed-forward outputs
        # (drop, :229,262-269).  Keep decisions are counter-based (Philox on (seed, step, site, element); db1_dropout in the header),
        # regenerated in the backward: no mask tensors.  The reference's released configuration has dropattn = 0 (config.py:167).
        self.embd_pdrop = float(g("embd_pdrop", 0.0) or 0.0)
        self.drop_p = float(g("drop", 0.0) or 0.0)
        # dropattn (:211; 0 in the released configuration, config.py:167): dropout on the attention probabilities.  Supported through the
        # MATERIALISED attention path only (the probabilities exist there as a tensor; the flash kernels do not draw masks) -- correct and
        # pinned to the oracle, not fast.
        self.dropattn = float(g("dropattn", 0.0) or 0.0)
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        self.dropout_seed = ((torch.initial_seed() * 0x9E3779B97F4A7C15) + rank) & 0xFFFFFFFFFFFFFFFF   # every data-parallel rank draws its own masks
        self._drop_step = 0                # bumped by every training forward: a new mask per micro-step
        self._drop_step_dev = None         # hipGraph-captured training steps (graphed_train.py): the counter lives on the device instead
        self._graph_static = False         # ... and per-weight-version caches are rebuilt inside every forward (static buffers)
        self.patch_size = int(g("vision_patch_size", 16))
        self.vision_channels = int(g("vision_num_input_channels", 3))
        self.vision_position_vocab_size = int(g("vision_position_vocab_size", 128))
        if compute_dtype is None:
            compute_dtype = g("compute_dtype", None)
        if compute_dtype is None:
            compute_dtype = torch.bfloat16 if bool(g("fp16", False)) else torch.float32
        if isinstance(compute_dtype, str):
            compute_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}[compute_dtype]
        assert compute_dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = compute_dtype
        self.vocab_pad = _round_up(self.total_vocab_size, 256)  # whole 256x256 GEMM tiles for the tied head
        self.keep_logits = True          # False: the CE backward overwrites the logits buffer (training engines)
        # training without a logits tensor: head GEMM, masked CE and the head's two gradient GEMMs in ONE sweep over 16 384-row chunks
        # (db1_lmhead_ce_fwd_bwd) during the forward; forward then returns (None, loss).  Set by the engine when keep_logits is False.
        self.fuse_head_loss = False
        self.loss_grad_scale = 1.0       # d(loss * this) is what backward() accumulates: 1 / gradient-accumulation steps (set by the engine)
        self.use_flash = True            # fused attention when the shape is supported
        self.use_flash_bwd = True        # fused backward kernels (False: recompute through the materialised path)
        # what the flash backward recomputes (include/db1_hip.h, db1_relattn_flash_bwd): "forward" = nothing, the forward keeps its
        # unnormalised probabilities per layer (B*H*L*L bf16 + B*H*L*L/32 floats each: 2.2 GiB per layer at 64 x 1024 tokens);
        # "scratch" = the query side recomputes and leaves P / dS in one scratch buffer for the key side; "recompute" = both sides
        self.flash_probs_mode = "forward"
        self.flash_probs_budget = 0.25   # "forward" only while the kept probabilities of all layers fit in this fraction of the device memory, else "scratch"
        self._probs_mode_cache = {}
        self.use_headbias_epilogue = True  # q + r_w_bias / q + r_r_bias written by the qkv projection's epilogue (large bf16 batches)
        self.wgrad_stash: Optional[WgradStash] = None   # gradient accumulation: weight gradients formed once per optimizer step (engine option defer_wgrad)
        self.wgrad_defer_ga = 0          # > 1: training forwards stash the weight-gradient operands of this many micro-steps (set by the engine)
        self._wg_slot = 0                # micro-step index inside the accumulation window (set by the engine before every forward)
        # > 1: training forwards keep their activations in window-sized buffers and the backward of the whole window runs ONCE, on the boundary
        # micro-step (BackwardWindow; set by the engine, option defer_backward)
        self.bwd_window_ga = 0
        self._win: Optional[BackwardWindow] = None
        self._win_on = False             # the running forward writes into the window
        self._drop_rps = 0               # window backward: rows per micro-step of the tensors the LayerNorm backward regenerates dropout for
        self.head_chunk_rows = int(os.environ.get("DB1_HEAD_CHUNK", "0"))   # rows of logits alive at a time in the fused head + loss sweep (0: the library's 16 384)
        self.use_partial_stash = os.environ.get("DB1_PARTIAL_STASH", "1") != "0"   # gradient accumulation: the small reductions once per optimizer step (WgradStash.alloc_parts)
        self.use_rnet_batched = os.environ.get("DB1_RNET_BATCHED", "1") != "0"   # r_net of all layers as one batched launch per forward
        self._R_all = None
        self.use_geglu_epilogue = os.environ.get("DB1_GEGLU_EPI", "1") != "0"   # GEGLU and its backward inside the feed-forward GEMMs' epilogues (large bf16 batches)
        self.use_channels_last = True    # bf16 image-patch embedder in channels-last layout (False: the NCHW kernels of the fp32 path)
        self.use_implicit_conv = True    # 64 -> 64 channel convolutions without a column matrix (conv_implicit.hip)
        self._conv_ops = {}              # (weight name, weight version) -> tap-major GEMM operand
        self.use_conv1_fused = os.environ.get("DB1_CONV1_FUSED", "1") != "0"      # 3 -> 64 channel convolution as one streaming kernel
        self.use_conv_res_epilogue = os.environ.get("DB1_CONV_RES", "1") != "0"   # residual sum of the patch block in the last convolution's epilogue
        self.use_proj_cl = os.environ.get("DB1_PROJ_CL", "1") != "0"   # channels-last patch embedder: projection against a column-permuted weight copy (no activation shuffles)
        self.use_decode = True           # inference with memory: K/V-cached path + fused decode attention when the shape allows
        self.use_decode_fused = True     # ... and, for <= 64 new tokens, linear maps as W streams that finish with GEGLU (post-LN)
        self.use_decode_ln_prologue = True   # ... <= 16 tokens: the residual LayerNorms ride on the way IN to the next linear map
        self.decode_ln_prologue_max_tokens = 4 # ... (more rows: LayerNorm launches; the prologue is redone by every workgroup of the linear map)
        self.use_decode_attn_partials = True   # ... <= 2 tokens (ring memory): the output projection merges the attention's chunk partials
        self.use_decode_chain = os.environ.get("DB1_DECODE_CHAIN", "1") != "0"   # ... ONE token (ring memory): the linear maps between two attention launches as one persistent launch (db1_decode_chain)
        self._chain_watch = None         # (pinned copy of the chain's error flag, its scratch) of the last chain call: check_decode_chain()
        self._wversion = 0               # bumped whenever the weights change (invalidates the inference caches)
        self._dec_state = None           # K/V cache of the memory returned by the last forward (see _decode_begin)
        self._dec_R = None               # (version, [R_i = r_net_i(sinusoid(dist)) for dist < mem_len + 64])
        self._ctx: Optional[_Ctx] = None
        # True: the gradient arena is logically zero -> the weight-gradient GEMMs of the next backward WRITE (beta = 0) instead of
        # accumulating, so the 4.8 GB of weight gradients are neither cleared after a step nor read back by their first writer
        self._grad_fresh = True
        self._tables: Dict[Tuple[int, int], torch.Tensor] = {}

        # ---- parameters: arena in backward-completion order (last layer first, embeddings last)
        d, H, D, di, dff = self.d_model, self.n_head, self.d_head, self.d_inner, self.d_ff
        ent: List[Tuple[str, Tuple[int, ...], int]] = []
        add = lambda n, s, alloc=None: ent.append((n, tuple(s), int(np.prod(s)) if alloc is None else alloc))
        for i in reversed(range(self.n_layer)):
            p = f"h.{i}."
            add(p + "pos_ff.layer_norm.weight", (d,)); add(p + "pos_ff.layer_norm.bias", (d,))
            add(p + "pos_ff.CoreNet.2.weight", (d, dff)); add(p + "pos_ff.CoreNet.2.bias", (d,))
            add(p + "pos_ff.CoreNet.0.weight", (di, d)); add(p + "pos_ff.CoreNet.0.bias", (di,))
            add(p + "dec_attn.layer_norm.weight", (d,)); add(p + "dec_attn.layer_norm.bias", (d,))
            add(p + "dec_attn.o_net.weight", (d, d)); add(p + "dec_attn.r_net.weight", (d, d))
            add(p + "dec_attn.qkv_net.weight", (3 * d, d))
            if self.untie_r:
                add(p + "dec_attn.r_r_bias", (H, D)); add(p + "dec_attn.r_w_bias", (H, D))
        if not self.untie_r:
            add("r_w_bias", (H, D)); add("r_r_bias", (H, D))
        add("rl_local_timestep_embedding.weight", (513, d))
        pe = "vision_encoder.patch_embeddings."
        C, ps = self.vision_channels, self.patch_size
        add(pe + "projection.weight", (d, 64, ps, ps)); add(pe + "projection.bias", (d,))
        add(pe + "residual_path.5.weight", (64, 64, 3, 3)); add(pe + "residual_path.5.bias", (64,))
        add(pe + "residual_path.3.weight", (64,)); add(pe + "residual_path.3.bias", (64,))
        add(pe + "residual_path.2.weight", (64, 64, 3, 3)); add(pe + "residual_path.2.bias", (64,))
        add(pe + "residual_path.0.weight", (64,)); add(pe + "residual_path.0.bias", (64,))
        add(pe + "conv1.weight", (64, C, 3, 3)); add(pe + "conv1.bias", (64,))
        add("vision_encoder.row_position_embeddings.weight", (self.vision_position_vocab_size, d))
        add("vision_encoder.col_position_embeddings.weight", (self.vision_position_vocab_size, d))
        if not self.share_input_output_embedding:
            add("lm_head.weight", (self.total_vocab_size, d), self.vocab_pad * d)
        add("word_embedding.weight", (self.total_vocab_size, d), self.vocab_pad * d)
        self.arena = ParamArena(ent, self.dev, compute_dtype)
        self._register_parameters()
        inv_freq = 1 / (10000 ** (torch.arange(0.0, d, 2.0) / d))  # transformer_xl.py:40 (same float32 op order, on the host)
        self.pos_emb = _Node()
        self.pos_emb.register_buffer("inv_freq", inv_freq.to(self.dev))
        self._init_weights()
        self.arena.sync_work()
        self.train()

    # ------------------------------------------------------------------ parameter plumbing
    def _register_parameters(self):
        shared: Dict[str, nn.Parameter] = {}
        for name in self.arena.offsets:
            prm = nn.Parameter(self.arena.view(self.arena.master, name), requires_grad=True)
            prm.grad = self.arena.view(self.arena.grad, name)
            shared[name] = prm
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if not hasattr(node, part):
                    node.add_module(part, _Node())
                node = getattr(node, part)
            node.register_parameter(parts[-1], prm)
        if not self.untie_r:  # tied u / v appear under every layer in the reference's state dict (:421-422)
            for i in range(self.n_layer):
                att = getattr(getattr(self.h, str(i)), "dec_attn")
                att.register_parameter("r_r_bias", shared["r_r_bias"])
                att.register_parameter("r_w_bias", shared["r_w_bias"])
        self.ic_encoder = self.vision_encoder  # transformer_xl.py:403-404
        self.lm_head_is_tied = self.share_input_output_embedding

    def _init_weights(self):
        """Distribution of the reference init (transformer_xl.py:444-468); values differ (different RNG)."""
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(torch.initial_seed() % (2 ** 31))
        with torch.no_grad():
            for name in self.arena.offsets:
                p = self.arena.view(self.arena.master, name)
                leaf = name.split(".")[-1]
                if "layer_norm" in name or ".residual_path.0." in name or ".residual_path.3." in name:   # LayerNorm / GroupNorm: (1, 0)
                    p.fill_(1.0 if leaf == "weight" else 0.0)
                elif "patch_embeddings" in name:  # nn.Conv2d default init (kaiming_uniform, a = sqrt(5))
                    wname = name.rsplit(".", 1)[0] + ".weight"
                    wshape = self.arena.offsets[wname][1]
                    bound = 1.0 / math.sqrt(int(np.prod(wshape[1:])))
                    p.copy_((torch.rand(p.shape, device=self.dev, generator=gen) * 2 - 1) * bound)
                elif leaf == "bias":
                    p.zero_()
                else:
                    p.copy_(torch.randn(p.shape, device=self.dev, generator=gen) * 0.02)
            if self.use_deepnorm:  # _deepnorm_init :444-454
                for i in range(self.n_layer):
                    for nm, gain in ((f"h.{i}.pos_ff.CoreNet.0.weight", self.deepnorm_beta), (f"h.{i}.pos_ff.CoreNet.2.weight", self.deepnorm_beta),
                                     (f"h.{i}.dec_attn.o_net.weight", self.deepnorm_beta)):
                        nn.init.xavier_uniform_(self.arena.view(self.arena.master, nm), gain=gain)
                    w = self.arena.view(self.arena.master, f"h.{i}.dec_attn.qkv_net.weight")
                    nn.init.xavier_uniform_(w, gain=1)
                    nn.init.xavier_uniform_(w[2 * self.d_model:, :], gain=self.deepnorm_beta)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = dict(state_dict)
        res = super().load_state_dict(sd, strict=strict, **kw)
        self.arena.sync_work()
        self.mark_weights_changed()
        return res

    def mark_weights_changed(self):
        """called after every parameter update (load_state_dict, optimizer step): drops the inference caches"""
        self._wversion += 1
        self._dec_state, self._dec_R = None, None

    def sync_work_params(self):
        """Refresh the bf16 working copy after the float32 master parameters were edited by hand."""
        self.arena.sync_work()
        self.mark_weights_changed()

    def W(self, name: str) -> torch.Tensor:
        """parameter in the compute dtype"""
        return self.arena.view(self.arena.work, name)

    def G(self, name: str) -> torch.Tensor:
        return self.arena.view(self.arena.grad, name)

    @property
    def device(self):
        return self.dev

    # ------------------------------------------------------------------ helpers
    SITE_EMBED, SITE_POS = 0xE0000000, 0xE0000001

    def _drop_args(self, p: float, site: int, step: Optional[int]):
        """(p, seed, site, step[, device step counter]) of one dropout site, or ops.NO_DROP outside training / at p = 0"""
        if step is None or p <= 0.0:
            return ops.NO_DROP
        if self._drop_rps:     # (the backward of a whole accumulation window: row r of the tensor belongs to step + r // rows_per_step)
            return (p, self.dropout_seed, site, step, self._drop_step_dev, self._drop_rps)
        return (p, self.dropout_seed, site, step, self._drop_step_dev) if self._drop_step_dev is not None else (p, self.dropout_seed, site, step)

    def _new(self, *shape, dtype=None):
        return torch.empty(*shape, device=self.dev, dtype=self.compute_dtype if dtype is None else dtype)

    def _keep(self, key: tuple, *shape, dtype=None):
        """a tensor the backward will read: this micro-step's row block of the accumulation window's buffer (BackwardWindow), or a fresh tensor"""
        if self._win_on:
            return self._win.take(self, key, shape, self.compute_dtype if dtype is None else dtype)
        return self._new(*shape, dtype=dtype)

    def _probs_mode(self, B: int, L: int) -> str:
        """flash_probs_mode, demoted from "forward" to "scratch" when keeping B*H*L*L bf16 (+ L/32 floats per row) for every layer would
        take more than flash_probs_budget of the device memory (decided once per shape)"""
        if self.flash_probs_mode != "forward":
            return self.flash_probs_mode
        key = (B, L, self.flash_probs_budget)
        mode = self._probs_mode_cache.get(key)
        if mode is None:
            need = self.n_layer * B * self.n_head * (ops.relattn_flash_probs_tiles(L) * 1024 + L * (L // 32) * 4)
            total = torch.cuda.get_device_properties(self.dev).total_memory
            # ... and only if it FITS beside what this process (weights, optimizer state, the data-parallel staging arena) and any other
            # process on the device already hold, plus the activations this shape keeps (~64 KB per token and layer) and the
            # head's logits chunk: the fallback is the scratch mode, not an out-of-memory error in the middle of the first step
            free, _ = torch.cuda.mem_get_info(self.dev)
            free += torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev)
            acts = int(1.15 * self.n_layer * B * L * 32 * self.d_model) + (6 << 30)
            mode = "forward" if (need <= self.flash_probs_budget * total and need + acts <= free) else "scratch"
            self._probs_mode_cache[key] = mode
        return mode

    def _window(self, qlen: int, mlen: int) -> int:
        """shift of the visibility predicate i - shift < j <= i + mlen (transformer_xl.py:551-567)."""
        klen = qlen + mlen
        if self.same_length:
            mask_len = klen - self.mem_len
            return qlen - mask_len if mask_len > 0 else qlen
        return klen

    def _sinusoid(self, klen: int) -> torch.Tensor:
        """R_in[dist] = cat(sin, cos)(min(dist, clamp) * inv_freq), dist = 0..klen-1 (transformer_xl.py:43-45,569-574).
        A constant table: built once per klen with the reference's float32 op order and cached on the device."""
        key = (klen, 0)
        if key not in self._tables:
            dist = torch.arange(0, klen, 1.0, dtype=torch.float32)
            if self.clamp_len > 0:
                dist.clamp_(max=self.clamp_len)
            inv = self.pos_emb.inv_freq.detach().float().cpu()
            s = torch.ger(dist, inv)
            tab = torch.cat([s.sin(), s.cos()], dim=-1)
            self._tables[key] = tab.to(self.dev).to(self.compute_dtype).contiguous()
        return self._tables[key]

    def _dev_ids(self, t) -> torch.Tensor:
        if not torch.is_tensor(t):
            t = torch.as_tensor(np.asarray(t))
        r = t.to(device=self.dev, dtype=torch.int64).contiguous()
        if self.bwd_window_ga > 1 and self.training and r.data_ptr() == t.data_ptr():
            r = r.clone()    # (deferred backward: the ids are read at the window's boundary -- the caller may have refilled its tensor by then)
        return r

    # ------------------------------------------------------------------ vision encoder (vision_embedding.py:65-180)
    def _vision_position_ids(self, h0: int, w0: int, n_img: int):
        """eval: midpoint rule; train: uniform pick in [low, high) per (image, position) (vision_embedding.py:134-172)."""
        vocab = self.vision_position_vocab_size
        seq = torch.arange(h0 * w0)
        row = torch.div(seq, w0, rounding_mode="trunc")
        col = seq % w0
        col_hi = ((col + 1) / w0 * vocab).to(torch.int32)
        col_lo = (col / w0 * vocab).to(torch.int32)
        row_hi = ((row + 1) / h0 * vocab).to(torch.int32)
        row_lo = (row / h0 * vocab).to(torch.int32)
        if self.training:
            r = (torch.rand(n_img, h0 * w0) * (row_hi - row_lo) + row_lo).floor().to(torch.int64)
            c = (torch.rand(n_img, h0 * w0) * (col_hi - col_lo) + col_lo).floor().to(torch.int64)
        else:
            r = ((row_lo + row_hi) / 2).int().to(torch.int64).unsqueeze(0).expand(n_img, -1)
            c = ((col_lo + col_hi) / 2).int().to(torch.int64).unsqueeze(0).expand(n_img, -1)
        return r.contiguous(), c.contiguous()

    def _conv3x3_fwd(self, x_nchw, wname, bname, N, Cin):
        """per-patch 3x3 conv as im2col + GEMM; returns NHWC output [N*256, 64] and the column matrix"""
        hw = self.patch_size * self.patch_size
        K = Cin * 9
        Kp = _round_up(K, 8)  # conv1: 27 -> 32 zero-padded columns so its weight gradient can use the MFMA tile kernel (split-K)
        cols = self._new(N * hw, Kp)
        ops.im2col3x3(x_nchw, cols, N, Cin, self.patch_size)
        out = self._new(N * hw, 64)
        ops.gemm(cols, self._conv_weight(wname, K, Kp).t(), out, bias=self.W(bname))
        return out, cols

    def _conv_weight(self, wname, K, Kp):
        """[64, Kp] view of a 3x3 conv weight in the compute dtype (zero-padded copy when Cin*9 is not a multiple of 8)"""
        w = self.W(wname).view(64, K)
        if Kp == K:
            return w
        wp = torch.zeros(64, Kp, device=self.dev, dtype=self.compute_dtype)
        ops.add2d(w, wp[:, :K], wp[:, :K])
        return wp

    # ---- channels-last pipeline (bf16, 16x16 patches): activations [N, 256, 64], tap-major column matrices, no layout shuffles
    # between the convolutions (vision.hip).  The fp32 parity path below keeps the reference's NCHW order end to end.
    def _conv_operand_cl(self, wname, Cin):
        """GEMM operand [64, kpad] (tap-major columns, zero padded to a multiple of 8) of a 3x3 conv weight, per weight version"""
        if self._graph_static:   # captured training step: the permuted copy is rebuilt by every replay, always into the same buffer
            key = (wname, "static")
            if key not in self._conv_ops:
                self._conv_ops[key] = torch.empty(64, _round_up(9 * Cin, 8), device=self.dev, dtype=self.compute_dtype)
            ops.conv_weight_permute(self.W(wname), self._conv_ops[key], 64, Cin)
            return self._conv_ops[key]
        key = (wname, self._wversion)
        if key not in self._conv_ops:
            self._conv_ops = {k: v for k, v in self._conv_ops.items() if k[1] in (self._wversion, "static")}
            wp = torch.empty(64, _round_up(9 * Cin, 8), device=self.dev, dtype=self.compute_dtype)
            ops.conv_weight_permute(self.W(wname), wp, 64, Cin)
            self._conv_ops[key] = wp
        return self._conv_ops[key]

    def _proj_operand_cl(self):
        """the patch projection weight [d, 64 * hw] with its columns in (pixel, channel) order, per weight version: the channels-last
        convolution output [N * hw, 64] IS [N, hw * 64], so the projection (and its data gradient) needs no layout shuffle of the
        activations -- a 67 MB copy of the weight per optimizer step instead of two passes over [N, 16 384] per batch"""
        wname = "vision_encoder.patch_embeddings.projection.weight"
        hw, d = self.patch_size * self.patch_size, self.d_model
        if self._graph_static:
            key = (wname + "^cl", "static")
            if key not in self._conv_ops:
                self._conv_ops[key] = torch.empty(d, hw * 64, device=self.dev, dtype=self.compute_dtype)
            ops.nchw_to_nhwc(self.W(wname), self._conv_ops[key], d, 64, hw)
            return self._conv_ops[key]
        key = (wname + "^cl", self._wversion)
        if key not in self._conv_ops:
            self._conv_ops = {k: v for k, v in self._conv_ops.items() if k[1] in (self._wversion, "static")}
            wp = torch.empty(d, hw * 64, device=self.dev, dtype=self.compute_dtype)
            ops.nchw_to_nhwc(self.W(wname), wp, d, 64, hw)
            self._conv_ops[key] = wp
        return self._conv_ops[key]

    def _conv_operand_t_cl(self, wname):
        """data-gradient operand [c_in, tap*64 + c_out] of a 64 -> 64 conv weight, per weight version"""
        if self._graph_static:
            key = (wname + "^T", "static")
            if key not in self._conv_ops:
                self._conv_ops[key] = torch.empty(64, 576, device=self.dev, dtype=self.compute_dtype)
            ops.conv_weight_permute_t(self.W(wname), self._conv_ops[key], 64, 64)
            return self._conv_ops[key]
        key = (wname + "^T", self._wversion)
        if key not in self._conv_ops:
            wt = torch.empty(64, 576, device=self.dev, dtype=self.compute_dtype)
            ops.conv_weight_permute_t(self.W(wname), wt, 64, 64)
            self._conv_ops[key] = wt
        return self._conv_ops[key]

    def _conv3x3_fwd_cl(self, x_cl, wname, bname, N, Cin, out=None, res=None):
        """returns (output [N*256, 64], what the backward needs: the input itself for the implicit 64-channel convs, else the
        column matrix); ``out`` / ``res``: write into this buffer / add this residual in the epilogue (implicit convolutions only)"""
        hw = self.patch_size * self.patch_size
        wp = self._conv_operand_cl(wname, Cin)
        if Cin == 64 and self.use_implicit_conv:  # implicit GEMM: the shifted pixels are gathered by the LDS-DMA, no column matrix
            out = self._new(N * hw, 64) if out is None else out
            ops.conv3x3_implicit_fwd(x_cl, wp, self.W(bname), out, N, sign=1, res=res)
            return out, x_cl
        assert out is None and res is None
        out = self._new(N * hw, 64)
        if Cin == 3 and wp.shape[1] == 32 and hw == 256 and self.use_conv1_fused:   # one streaming kernel: column matrix + convolution
            cols = self._new(N * hw, 32)
            ops.conv1_fused_fwd(x_cl, wp, self.W(bname), cols, out, N)
            return out, cols
        cols = self._new(N * hw, wp.shape[1])
        ops.im2col3x3_nhwc(x_cl, cols, N, Cin, self.patch_size)
        ops.gemm(cols, wp.t(), out, bias=self.W(bname))
        return out, cols

    def _conv3x3_bwd_cl(self, dy, cols, wname, bname, N, Cin, need_dx):
        wp = self._conv_operand_cl(wname, Cin)
        gp = torch.zeros(64, wp.shape[1], device=self.dev, dtype=torch.float32)
        implicit = Cin == 64 and cols.shape[1] == 64  # `cols` is the conv input
        if implicit:   # (the bias gradient -- column sums of dy -- comes out of the same kernel)
            ops.conv3x3_implicit_wgrad(dy, cols, gp, N, gbias_acc=self.G(bname))
        else:
            ops.gemm(dy.t(), cols, gp, beta=1.0)
            ops.colsum_acc(dy, self.G(bname))
        ops.conv_wgrad_unpermute(gp, self.G(wname), 64, Cin)
        if not need_dx:
            return None
        if implicit:
            dx = self._new(N * self.patch_size * self.patch_size, Cin)
            ops.conv3x3_implicit_fwd(dy, self._conv_operand_t_cl(wname), None, dx, N, sign=-1)
            return dx
        dcols = self._new(cols.shape[0], wp.shape[1])
        ops.gemm(dy, wp, dcols)
        dx = self._new(N * self.patch_size * self.patch_size, Cin)
        ops.col2im3x3_nhwc(dcols, dx, N, Cin, self.patch_size)
        return dx

    def _vision_fwd_cl(self, pixels, c, n_img, C, Hh, Ww):
        p, d = self.patch_size, self.d_model
        hw = p * p
        N = n_img * (Hh // p) * (Ww // p)
        pe = "vision_encoder.patch_embeddings."
        patches = self._new(N * hw, C)
        ops.patch_normalize_nhwc(pixels, patches, p)
        c.c1, c.cols1 = self._conv3x3_fwd_cl(patches, pe + "conv1.weight", pe + "conv1.bias", N, C)
        a0 = self._new(N * hw, 64)
        c.m0, c.r0 = self._new(N * 32, dtype=torch.float32), self._new(N * 32, dtype=torch.float32)
        ops.groupnorm_gelu_nhwc_fwd(c.c1, self.W(pe + "residual_path.0.weight"), self.W(pe + "residual_path.0.bias"), a0, c.m0, c.r0, N, 64, hw)
        c.c2, c.cols2 = self._conv3x3_fwd_cl(a0, pe + "residual_path.2.weight", pe + "residual_path.2.bias", N, 64)
        a1 = self._new(N * hw, 64)
        c.m1, c.r1 = self._new(N * 32, dtype=torch.float32), self._new(N * 32, dtype=torch.float32)
        ops.groupnorm_gelu_nhwc_fwd(c.c2, self.W(pe + "residual_path.3.weight"), self.W(pe + "residual_path.3.bias"), a1, c.m1, c.r1, N, 64, hw)
        if self.use_proj_cl and self.use_implicit_conv and self.use_conv_res_epilogue:
            # the last convolution adds the residual in its epilogue and writes straight into the projection's (row-padded) operand
            Np = _round_up(N, 256) if N >= 512 else N
            ypad = self._new(Np * hw, 64)
            _, c.cols3 = self._conv3x3_fwd_cl(a1, pe + "residual_path.5.weight", pe + "residual_path.5.bias", N, 64, out=ypad[:N * hw], res=c.c1)
            if Np > N:
                ypad[N * hw:].zero_()
            c.y, c.y_cl, c.Np = ypad.view(Np, hw * 64), True, Np
            emb_pad = self._new(Np, d)
            ops.gemm(c.y, self._proj_operand_cl().t(), emb_pad, bias=self.W(pe + "projection.bias"))
            return emb_pad[:N], N
        c3, c.cols3 = self._conv3x3_fwd_cl(a1, pe + "residual_path.5.weight", pe + "residual_path.5.bias", N, 64)
        if self.use_proj_cl:
            # (y, x, c) flattening against the column-permuted projection weight.  The patch count of a mixed batch is whatever the data gives
            # (4116, 20 680 ...): rows are padded with zeros to a multiple of 256 so that the K = 16 384 projection and its two gradients take
            # the 256 x 256 kernels (the 128-tile / generic kernels ran them at 0.24 PFLOP/s) -- the residual add writes into the padded buffer
            Np = _round_up(N, 256) if N >= 512 else N
            ypad = self._new(Np * hw, 64)
            ops.add(c.c1, c3, ypad[:N * hw])                   # residual
            if Np > N:
                ypad[N * hw:].zero_()
            c.y, c.y_cl, c.Np = ypad.view(Np, hw * 64), True, Np
            emb_pad = self._new(Np, d)
            ops.gemm(c.y, self._proj_operand_cl().t(), emb_pad, bias=self.W(pe + "projection.bias"))
            return emb_pad[:N], N
        ops.add(c.c1, c3, c3)                                  # residual
        emb = self._new(N, d)
        c.y, c.y_cl = self._new(N, 64 * hw), False             # (c, y, x) flattening = the projection weight's layout
        ops.nhwc_to_nchw(c3, c.y, N, 64, hw)
        ops.gemm(c.y, self.W(pe + "projection.weight").view(d, 64 * hw).t(), emb, bias=self.W(pe + "projection.bias"))
        return emb, N

    def _vision_bwd_cl(self, dy_cl, c, N):
        """dy_cl [N*256, 64]: gradient w.r.t. the residual sum, channels-last"""
        pe = "vision_encoder.patch_embeddings."
        hw = self.patch_size * self.patch_size
        da1 = self._conv3x3_bwd_cl(dy_cl, c.cols3, pe + "residual_path.5.weight", pe + "residual_path.5.bias", N, 64, True)
        dc2 = self._new(N * hw, 64)
        ops.groupnorm_gelu_nhwc_bwd(da1, c.c2, self.W(pe + "residual_path.3.weight"), self.W(pe + "residual_path.3.bias"), c.m1, c.r1, dc2,
                                    self.G(pe + "residual_path.3.weight"), self.G(pe + "residual_path.3.bias"), N, 64, hw)
        da0 = self._conv3x3_bwd_cl(dc2, c.cols2, pe + "residual_path.2.weight", pe + "residual_path.2.bias", N, 64, True)
        dc1 = self._new(N * hw, 64)
        ops.groupnorm_gelu_nhwc_bwd(da0, c.c1, self.W(pe + "residual_path.0.weight"), self.W(pe + "residual_path.0.bias"), c.m0, c.r0, dc1,
                                    self.G(pe + "residual_path.0.weight"), self.G(pe + "residual_path.0.bias"), N, 64, hw,
                                    res=dy_cl.view(N * hw, 64))    # + the residual branch's gradient, in the same pass
        self._conv3x3_bwd_cl(dc1, c.cols1, pe + "conv1.weight", pe + "conv1.bias", N, c.C, False)

    def _vision_fwd(self, pixels: torch.Tensor, row_ids=None, col_ids=None):
        pixels = pixels.to(device=self.dev, dtype=torch.float32).contiguous()
        n_img, C, Hh, Ww = pixels.shape
        p, d = self.patch_size, self.d_model
        hw = p * p
        h0, w0 = Hh // p, Ww // p
        N = n_img * h0 * w0
        pe = "vision_encoder.patch_embeddings."
        c = _Ctx()
        c.cl = self.compute_dtype == torch.bfloat16 and hw == 256 and self.use_channels_last
        if c.cl:
            emb, N = self._vision_fwd_cl(pixels, c, n_img, C, Hh, Ww)
            return self._vision_finish(emb, c, N, C, n_img, h0, w0, row_ids, col_ids)
        patches = self._new(N, C, p, p)
        ops.patch_normalize(pixels, patches, p)
        c1, c.cols1 = self._conv3x3_fwd(patches, pe + "conv1.weight", pe + "conv1.bias", N, C)
        c.c1n = self._new(N, 64, hw)
        ops.nhwc_to_nchw(c1, c.c1n, N, 64, hw)
        a0 = self._new(N, 64, hw)
        c.m0, c.r0 = self._new(N * 32, dtype=torch.float32), self._new(N * 32, dtype=torch.float32)
        ops.groupnorm_gelu_fwd(c.c1n, self.W(pe + "residual_path.0.weight"), self.W(pe + "residual_path.0.bias"), a0, c.m0, c.r0, N, 64, hw)
        c2, c.cols2 = self._conv3x3_fwd(a0, pe + "residual_path.2.weight", pe + "residual_path.2.bias", N, 64)
        c.c2n = self._new(N, 64, hw)
        ops.nhwc_to_nchw(c2, c.c2n, N, 64, hw)
        a1 = self._new(N, 64, hw)
        c.m1, c.r1 = self._new(N * 32, dtype=torch.float32), self._new(N * 32, dtype=torch.float32)
        ops.groupnorm_gelu_fwd(c.c2n, self.W(pe + "residual_path.3.weight"), self.W(pe + "residual_path.3.bias"), a1, c.m1, c.r1, N, 64, hw)
        c3, c.cols3 = self._conv3x3_fwd(a1, pe + "residual_path.5.weight", pe + "residual_path.5.bias", N, 64)
        ops.add(c1, c3, c3)                                   # residual (NHWC)
        c.y = self._new(N, 64 * hw)                            # NCHW flatten = projection weight layout
        ops.nhwc_to_nchw(c3, c.y, N, 64, hw)
        emb = self._new(N, d)
        ops.gemm(c.y, self.W(pe + "projection.weight").view(d, 64 * hw).t(), emb, bias=self.W(pe + "projection.bias"))
        return self._vision_finish(emb, c, N, C, n_img, h0, w0, row_ids, col_ids)

    def _vision_finish(self, emb, c, N, C, n_img, h0, w0, row_ids, col_ids):
        d = self.d_model
        if row_ids is None:
            row_ids, col_ids = self._vision_position_ids(h0, w0, n_img)
        c.row_ids, c.col_ids = self._dev_ids(row_ids).reshape(-1), self._dev_ids(col_ids).reshape(-1)
        assert c.row_ids.numel() == N
        # emb += row_position_embeddings[row_ids] + col_position_embeddings[col_ids] (vision_embedding.py:170-178) in one pass over emb
        ops.vision_pos_add(emb.view(N, d), self.W("vision_encoder.row_position_embeddings.weight"), self.W("vision_encoder.col_position_embeddings.weight"),
                           c.row_ids, c.col_ids)
        c.N, c.C, c.n_img = N, C, n_img
        return emb.view(n_img, h0 * w0, d), c

    def _conv3x3_bwd(self, dy_nhwc, cols, wname, bname, N, Cin, need_dx):
        K, Kp = Cin * 9, cols.shape[1]
        if Kp == K:
            ops.gemm(dy_nhwc.t(), cols, self.G(wname).view(64, K), beta=1.0)
        else:  # padded columns: reduce into a [64, Kp] float32 scratch, then add its first K columns to the gradient
            gp = torch.zeros(64, Kp, device=self.dev, dtype=torch.float32)
            ops.gemm(dy_nhwc.t(), cols, gp, beta=1.0)
            g = self.G(wname).view(64, K)
            ops.add2d(gp[:, :K], g, g)
        ops.colsum_acc(dy_nhwc, self.G(bname))
        if not need_dx:
            return None
        dcols = self._new(cols.shape[0], Kp)
        ops.gemm(dy_nhwc, self._conv_weight(wname, K, Kp), dcols)
        dx = self._new(N, Cin, self.patch_size * self.patch_size)
        ops.col2im3x3(dcols, dx, N, Cin, self.patch_size)
        return dx

    def _vision_bwd(self, demb: torch.Tensor, c: _Ctx):
        """demb [N, d] (compute dtype, contiguous)"""
        p, d = self.patch_size, self.d_model
        hw, N = p * p, c.N
        pe = "vision_encoder.patch_embeddings."
        ops.embed_scatter_add(demb, c.row_ids, self.G("vision_encoder.row_position_embeddings.weight"))
        ops.embed_scatter_add(demb, c.col_ids, self.G("vision_encoder.col_position_embeddings.weight"))
        if c.cl and getattr(c, "y_cl", False):
            # the weight gradient comes out with (pixel, channel) columns: shuffled back per weight row and added (fp32, two passes over 134 MB),
            # the data gradient [N, hw * 64] is channels-last already
            Np = c.Np
            dpad = demb
            if Np > N:                                             # zero rows for the padded patches
                dpad = self._new(Np, d)
                dpad[:N].copy_(demb)
                dpad[N:].zero_()
            gp = torch.empty(d, hw * 64, device=self.dev, dtype=torch.float32)
            ops.gemm(dpad.t(), c.y, gp)
            gpt = torch.empty(d, 64 * hw, device=self.dev, dtype=torch.float32)
            ops.nhwc_to_nchw(gp, gpt, d, 64, hw)
            gw = self.G(pe + "projection.weight").view(d, 64 * hw)
            ops.add(gpt, gw, gw)
            ops.colsum_acc(demb, self.G(pe + "projection.bias"))
            dy_nhwc = self._new(Np * hw, 64)
            ops.gemm(dpad, self._proj_operand_cl(), dy_nhwc.view(Np, hw * 64))
            return self._vision_bwd_cl(dy_nhwc[:N * hw], c, N)
        ops.gemm(demb.t(), c.y, self.G(pe + "projection.weight").view(d, 64 * hw), beta=1.0)
        ops.colsum_acc(demb, self.G(pe + "projection.bias"))
        dy = self._new(N, 64 * hw)
        ops.gemm(demb, self.W(pe + "projection.weight").view(d, 64 * hw), dy)
        dy_nhwc = self._new(N * hw, 64)
        ops.nchw_to_nhwc(dy, dy_nhwc, N, 64, hw)
        if c.cl:
            return self._vision_bwd_cl(dy_nhwc, c, N)
        da1 = self._conv3x3_bwd(dy_nhwc, c.cols3, pe + "residual_path.5.weight", pe + "residual_path.5.bias", N, 64, True)
        dc2n = self._new(N, 64, hw)
        ops.groupnorm_gelu_bwd(da1, c.c2n, self.W(pe + "residual_path.3.weight"), self.W(pe + "residual_path.3.bias"), c.m1, c.r1, dc2n,
                               self.G(pe + "residual_path.3.weight"), self.G(pe + "residual_path.3.bias"), N, 64, hw)
        dc2 = self._new(N * hw, 64)
        ops.nchw_to_nhwc(dc2n, dc2, N, 64, hw)
        da0 = self._conv3x3_bwd(dc2, c.cols2, pe + "residual_path.2.weight", pe + "residual_path.2.bias", N, 64, True)
        dc1n = self._new(N, 64, hw)
        ops.groupnorm_gelu_bwd(da0, c.c1n, self.W(pe + "residual_path.0.weight"), self.W(pe + "residual_path.0.bias"), c.m0, c.r0, dc1n,
                               self.G(pe + "residual_path.0.weight"), self.G(pe + "residual_path.0.bias"), N, 64, hw)
        dc1 = self._new(N * hw, 64)
        ops.nchw_to_nhwc(dc1n, dc1, N, 64, hw)
        ops.add(dc1, dy_nhwc, dc1)                             # residual branch
        self._conv3x3_bwd(dc1, c.cols1, pe + "conv1.weight", pe + "conv1.bias", N, c.C, False)

    # ------------------------------------------------------------------ per-modality embedding (transformer_xl.py:621-748)
    def _embed_task(self, task, compute_loss: bool):
        kind = type(task).__name__
        d = self.d_model
        E = self.W("word_embedding.weight")
        c = _Ctx()
        c.kind = kind
        if kind == "NLPTaskInput":
            ids = self._dev_ids(task.text_seq)
            B, L = ids.shape
            emb = self._new(B, L, d)
            ops.embed_gather(E, ids.view(-1), emb.view(B * L, d))
            c.ids = ids
        elif kind == "RLTaskInput":
            ids = self._dev_ids(task.tensor_seq)
            B, L = ids.shape
            pos = self._dev_ids(task.position_id)
            vis = None
            if task.vision_seq is not None:
                img = task.vision_seq
                img = img.reshape(-1, *img.shape[-3:])
                v, c.vis = self._vision_fwd(img, getattr(task, "vision_row_ids", None), getattr(task, "vision_col_ids", None))
                vis = v.reshape(B, -1, d).contiguous()
                c.vis_shape = vis.shape
            rl_label = None
            if compute_loss and task.label is not None:
                rl_label = self._dev_ids(task.label).clone()  # "-1 -> 0" (:644-645) is applied to a private copy by the kernel
            emb = self._new(B, L, d)
            # "-1 -> 0" on the labels only when there are image placeholders to point at (transformer_xl.py:630-645)
            ops.rl_assemble_fwd(E, self.W("rl_local_timestep_embedding.weight"), vis, ids, pos, rl_label if vis is not None else None, emb)
            c.ids, c.pos = ids, pos
        elif kind in ("ICTaskInput", "VQATaskInput"):
            prompt, text = self._dev_ids(task.prompt_seq), self._dev_ids(task.text_seq)
            v, c.vis = self._vision_fwd(task.img_seq, getattr(task, "vision_row_ids", None), getattr(task, "vision_col_ids", None))
            B, P_, nv, Tt = prompt.shape[0], prompt.shape[1], v.shape[1], text.shape[1]
            L = P_ + nv + Tt
            emb = self._new(B, L, d)
            e2 = emb.view(B * L, d)
            # gather straight into the concatenated layout (row stride L*d per sample handled by per-sample calls)
            for b in range(B):
                ops.embed_gather(E, prompt[b], emb[b, :P_])
                ops.embed_gather(E, text[b], emb[b, P_ + nv:])
            emb[:, P_:P_ + nv].copy_(v)  # placement of the patch embeddings (data movement only)
            c.prompt, c.text, c.nv = prompt, text, nv
        else:
            raise TypeError(f"unknown task input type {kind}")
        label = mask = None
        if compute_loss:
            label = rl_label if kind == "RLTaskInput" else self._dev_ids(task.label)
            mask = task.loss_mask.to(device=self.dev, dtype=torch.float32) if torch.is_tensor(task.loss_mask) \
                else torch.as_tensor(np.asarray(task.loss_mask), dtype=torch.float32, device=self.dev)
        return emb, label, mask, c

    def _embed_bwd(self, dh: torch.Tensor, ecs: List[_Ctx], shapes):
        d = self.d_model
        gE = self.arena.view(self.arena.grad, "word_embedding.weight")   # [total_vocab_size, d]: ids beyond it read / write nothing
        b0 = 0
        for c, (B, L) in zip(ecs, shapes):
            de = dh[b0:b0 + B]
            b0 += B
            if c.kind == "NLPTaskInput":
                ops.embed_scatter_add(de.reshape(B * L, d), c.ids.view(-1), gE)
            elif c.kind == "RLTaskInput":
                dvis = self._new(*c.vis_shape) if hasattr(c, "vis") else None
                ops.rl_assemble_bwd(de.contiguous(), c.ids, c.pos, gE, self.G("rl_local_timestep_embedding.weight"), dvis)
                if dvis is not None:
                    self._vision_bwd(dvis.view(-1, d), c.vis)
            else:
                P_, nv = c.prompt.shape[1], c.nv
                for b in range(B):
                    ops.embed_scatter_add(de[b, :P_], c.prompt[b], gE)
                    ops.embed_scatter_add(de[b, P_ + nv:], c.text[b], gE)
                self._vision_bwd(de[:, P_:P_ + nv].contiguous().view(-1, d), c.vis)

    # ------------------------------------------------------------------ attention
    def _bias(self, name: str, i: int) -> torch.Tensor:
        return self.W(f"h.{i}.dec_attn.{name}" if self.untie_r else name)

    def _bias_grad(self, name: str, i: int) -> torch.Tensor:
        return self.G(f"h.{i}.dec_attn.{name}" if self.untie_r else name)

    def _attn_probs(self, qkv, R, u, vb, B, Lq, Lk, mlen, shift):
        """materialised path: returns (P [H,B,Lq,Lk] f32, T buffer, qu, qv)"""
        H, D = self.n_head, self.d_head
        nd = R.shape[0]
        qu, qv = self._new(B, Lq, H, D), self._new(B, Lq, H, D)
        ops.relattn_add_head_bias(qkv, u, vb, qu, qv, B, Lq, Lk, H, D)
        qkv5 = qkv.view(B, Lk, 3, H, D)
        AC = self._new(H, B, Lq, Lk, dtype=torch.float32)
        ops.gemm_batched(qu.permute(2, 0, 1, 3), qkv5[:, :, 1].permute(2, 0, 3, 1), AC)
        T = self._new(H, B, Lq, nd, dtype=torch.float32)
        ops.gemm_batched(qv.permute(2, 0, 1, 3), R.view(nd, H, D).permute(1, 2, 0).unsqueeze(1).expand(H, B, D, nd), T)
        ops.relattn_softmax_fwd(AC, T, None, H, B, Lq, Lk, nd, mlen, shift, 1.0 / math.sqrt(D))
        return AC, T, qu, qv

    def _attention_fwd(self, qkv, R, i, B, Lq, Lk, mlen, shift, c: Optional[_Ctx], quv=None, dstep=None, av_out=None):
        H, D = self.n_head, self.d_head
        u, vb = self._bias("r_w_bias", i), self._bias("r_r_bias", i)
        av = self._new(B, Lq, H, D) if av_out is None else av_out.view(B, Lq, H, D)
        pdrop = self._drop_args(self.dropattn, 4 * i + 2, dstep)     # dropout on the probabilities: materialised path only
        flash = (self.use_flash and mlen == 0 and Lq == Lk and shift >= 1 and ops.relattn_flash_supported(B, Lq, H, D, self.compute_dtype) and
                 pdrop is ops.NO_DROP)
        assert quv is None or flash
        if flash:
            if quv is not None:  # written by the projection's epilogue (db1_gemm_nt_headbias)
                qu, qv = quv
            else:
                qu, qv = self._keep(("qu", i), B, Lq, H, D), self._keep(("qv", i), B, Lq, H, D)
                ops.relattn_add_head_bias(qkv, u, vb, qu, qv, B, Lq, Lk, H, D)
            lse = self._keep(("lse", i), B, H, Lq, dtype=torch.float32)
            probs = mblk = None
            if c is not None and self.use_flash_bwd and self._probs_mode(B * (self.bwd_window_ga if self._win_on else 1), Lq) == "forward":
                probs = self._keep(("probs", i), B * H, ops.relattn_flash_probs_tiles(Lq), 512)    # (the causal triangle of (key block, query tile) images)
                mblk = self._keep(("mblk", i), B * H, Lq // 32, Lq, dtype=torch.float32)
            ops.relattn_flash_fwd(qu, qv, qkv.view(B, Lk, 3, H, D), R, av, lse, B, Lq, H, D, shift, 1.0 / math.sqrt(D), probs=probs, mblk=mblk)
            if c is not None:
                c.lse, c.qu, c.qv, c.probs, c.mblk = lse, qu, qv, probs, mblk
        else:
            Pm, _, _, _ = self._attn_probs(qkv, R, u, vb, B, Lq, Lk, mlen, shift)
            if pdrop is not ops.NO_DROP:
                ops.dropout(Pm, Pm, pdrop)            # (:211; [H, B, Lq, Lk] element order; the backward regenerates the mask)
            qkv5 = qkv.view(B, Lk, 3, H, D)
            ops.gemm_batched(Pm, qkv5[:, :, 2].permute(2, 0, 1, 3), av.permute(2, 0, 1, 3))
        if c is not None:
            c.flash = flash
        return av

    # ---- inference with memory (evaluate_rl.py:157-266): K/V cache + fused decode attention.  The caller-visible contract is the
    # reference's: ``mems`` are per-layer hidden states (init_mem / _update_mem, :470-504) passed back opaquely.  Beside them the
    # model keeps the projected keys / values of exactly those tensors; if the caller hands in anything else (identity check) the
    # cache is rebuilt from the hidden states, so results never depend on it.
    def _decode_R(self):
        """R_i[dist] = r_net_i(sinusoid(dist)) for dist < mem_len + 64: depends on the weights only, computed once per weight version"""
        if self._dec_R is None or self._dec_R[0] != self._wversion:
            R_in = self._sinusoid(self.mem_len + 64)
            Rs = []
            for i in range(self.n_layer):
                R = torch.empty(R_in.shape[0], self.d_model, device=self.dev, dtype=self.compute_dtype)
                ops.gemm(R_in, self.W(f"h.{i}.dec_attn.r_net.weight").t(), R)
                Rs.append(R)
            self._dec_R = (self._wversion, Rs)
        return self._dec_R[1]

    def _dec_pack(self, mems, kv):
        """cache record: the memory tensors, their version counters at this moment, and the projected keys / values that belong to them"""
        return SimpleNamespace(mems=mems, mem_versions=[m._version for m in mems], kv=kv, version=self._wversion)

    def _decode_begin(self, mems, B, L, mlen):
        """decode context for this call, or None when the fused path does not apply (then the materialised path runs)"""
        if not (self.use_decode and self.compute_dtype == torch.bfloat16 and mlen + L <= self.mem_len + 64 and
                ops.relattn_decode_supported(B, L, mlen + L, self.n_head, self.d_head, self.compute_dtype)):
            return None
        st = self._dec_state
        H, D, d = self.n_head, self.d_head, self.d_model
        # the cache belongs to exactly the tensors the previous call returned, UNMODIFIED: same objects and same torch version counters
        # (an in-place edit such as `mems[i][done_env] = 0` on an episode reset keeps the identity but bumps `_version` -> rebuild)
        valid = (st is not None and st.version == self._wversion and len(st.mems) == len(mems) and
                 all(a is b and a._version == v for a, b, v in zip(st.mems, mems, st.mem_versions)) and
                 st.kv[0].shape[0] == B and st.kv[0].shape[1] >= mlen)
        if valid:
            kv = st.kv
        else:  # rebuild from the hidden states (first call after init_mem, or a caller that edited the memory)
            kv = []
            for i in range(self.n_layer):
                p = f"h.{i}."
                m = mems[i].to(self.compute_dtype).contiguous().view(B * mlen, d)
                if self.pre_lnorm:
                    hin = self._new(B * mlen, d)
                    m1, r1 = self._new(B * mlen, dtype=torch.float32), self._new(B * mlen, dtype=torch.float32)
                    ops.layernorm_residual_fwd(m, None, 1.0, self.W(p + "dec_attn.layer_norm.weight"), self.W(p + "dec_attn.layer_norm.bias"),
                                               hin, None, m1, r1, self.layer_norm_epsilon)
                    m = hin
                qkv = self._new(B * mlen, 3 * d)
                ops.gemm(m, self.W(p + "dec_attn.qkv_net.weight").t(), qkv)
                kv.append(qkv.view(B, mlen, 3, H, D)[:, :, 1:3].contiguous())
        return SimpleNamespace(kv=kv, R=self._decode_R(), new_kv=[])

    def _decode_fused_ok(self, T, keep, dstep):
        d, dff = self.d_model, self.d_ff
        return (self.use_decode_fused and not keep and dstep is None and self.compute_dtype == torch.bfloat16 and self.activation_fn == "geglu" and
                self.W("h.0.pos_ff.layer_norm.weight").dtype == self.W("h.0.pos_ff.layer_norm.bias").dtype and
                ops.linear_decode_supported(T, d, d, False, True) and ops.linear_decode_supported(T, dff, d, True, False) and
                ops.linear_decode_supported(T, d, dff, False, True))

    def _decode_ln_prologue_ok(self, T):
        d, dff = self.d_model, self.d_ff
        # every workgroup of the linear map normalises ALL T input rows itself, so the prologue's cost grows with T while a LayerNorm launch
        # costs ~5 us whatever T is: measured at the 1.3B shapes (profiles/r05_decode_batched.txt) the qkv projection with the prologue takes
        # 8.4 / 11.8 / 34 us at T = 1 / 4 / 16 against 4.5 + 7.5 ... 9 us for launch + plain projection -> on the way in up to 4 rows only
        return (self.use_decode_ln_prologue and T <= self.decode_ln_prologue_max_tokens and ops.linear_decode_supported(T, 3 * d, d, False, False, True) and
                ops.linear_decode_supported(T, dff, d, True, False, True))

    def _attention_decode(self, qkv, i, B, L, mlen, shift, dec):
        H, D = self.n_head, self.d_head
        u, vb = self._bias("r_w_bias", i), self._bias("r_r_bias", i)
        if getattr(dec, "ring", None) is not None:   # K / V of the memory in a ring, appended in place by the attention launch (decode.RingMemory)
            if getattr(dec, "partials", False):       # one or two new tokens: the output projection merges the chunk partials itself
                part = self._new(ops.relattn_decode_ring_part_numel(B, L, mlen + L, H), dtype=torch.float32)
                ops.relattn_decode_ring_fwd(qkv, u, vb, dec.ring.kv[i], dec.ring.state, dec.R[i], None, B, L, mlen, H, D, shift, 1.0 / math.sqrt(D), part=part)
                return part
            av = self._new(B, L, H, D)
            ops.relattn_decode_ring_fwd(qkv, u, vb, dec.ring.kv[i], dec.ring.state, dec.R[i], av, B, L, mlen, H, D, shift, 1.0 / math.sqrt(D))
            return av
        qu, qv = self._new(B, L, H, D), self._new(B, L, H, D)
        ops.relattn_add_head_bias(qkv, u, vb, qu, qv, B, L, L, H, D)
        old = dec.kv[i]
        kv_all = torch.cat([old[:, old.shape[1] - mlen:], qkv.view(B, L, 3, H, D)[:, :, 1:3]], dim=1)  # [B, klen, 2, H, D]
        klen = mlen + L
        av = self._new(B, L, H, D)
        ops.relattn_decode_fwd(qu, qv, kv_all[:, :, 0], kv_all[:, :, 1], dec.R[i], av, B, L, klen, mlen, H, D, shift, 1.0 / math.sqrt(D))
        dec.new_kv.append(kv_all[:, max(0, klen - self.mem_len):])
        return av

    def _attention_bwd(self, dav, c: _Ctx, i, B, L, shift, dstep=None, dqkv_out=None, dR_out=None, uv_parts=None):
        """returns dqkv [B*L, 3d] and dR [L, d]; accumulates du / dv_bias"""
        H, D, d = self.n_head, self.d_head, self.d_model
        u, vb = self._bias("r_w_bias", i), self._bias("r_r_bias", i)
        qkv, R = c.qkv, c.R
        nd = R.shape[0]
        # deferred backward over an accumulation window: the batch is ng blocks of B / ng sequences (micro-steps), each with its own R
        Rg = getattr(c, "Rg", None)
        ng = 1 if Rg is None else int(Rg.shape[0])
        scale = 1.0 / math.sqrt(D)
        dqkv = self._new(B * L, 3 * d) if dqkv_out is None else dqkv_out
        dqkv5 = dqkv.view(B, L, 3, H, D)
        qkv5 = qkv.view(B, L, 3, H, D)
        dav4 = dav.view(B, L, H, D)
        if c.flash and self.use_flash_bwd:
            qu, qv = c.qu, c.qv
            # dT[h,b,i,dist]: entries with dist > i are never written and must read as zero.  With the plain causal window every
            # entry dist <= i is rewritten by each call, so ONE zero-initialised buffer is reused by all layers and steps; a sliding
            # window leaves unvisited entries below the diagonal, so it gets a fresh zeroed buffer per call.
            if shift >= L:
                key = (H, B, L)
                if getattr(self, "_dT_key", None) != key:
                    self._dT_buf = torch.zeros(H, B, L, L, device=self.dev, dtype=self.compute_dtype)
                    self._dT_key = key
                dT = self._dT_buf
            else:
                dT = torch.zeros(H, B, L, L, device=self.dev, dtype=self.compute_dtype)
            delta = self._new(B, H, L, dtype=torch.float32)
            ops.relattn_flash_bwd(qu, qv, qkv5, R, c.av, dav4, c.lse, delta, dqkv5, dT, B, L, H, D, shift, scale,
                                  store_probs=self._probs_mode(B, L) != "recompute", probs=c.probs, mblk=c.mblk)
            c.probs = c.mblk = None
        else:
            Pm, T, qu, qv = self._attn_probs(qkv, R, u, vb, B, L, L, 0, shift)
            dP = self._new(H, B, L, L, dtype=torch.float32)
            ops.gemm_batched(dav4.permute(2, 0, 1, 3), qkv5[:, :, 2].permute(2, 0, 3, 1), dP)
            Pd = Pm
            pdrop = self._drop_args(self.dropattn, 4 * i + 2, dstep)
            if pdrop is not ops.NO_DROP:   # the forward's dropped probabilities again (counter-based mask), and the gradient through that dropout
                Pd = torch.empty_like(Pm)
                ops.dropout(Pm, Pd, pdrop)
                ops.dropout(dP, dP, pdrop)
            ops.gemm_batched(Pd.transpose(2, 3), dav4.permute(2, 0, 1, 3), dqkv5[:, :, 2].permute(2, 0, 1, 3))          # dV
            dT = T
            ops.relattn_softmax_bwd(Pm, dP, dT, H, B, L, L, nd, 0, shift, scale)
            dS = dP
            ops.gemm_batched(dS, qkv5[:, :, 1].permute(2, 0, 1, 3), dqkv5[:, :, 0].permute(2, 0, 1, 3))                  # dq_k
            ops.gemm_batched(dS.transpose(2, 3), qu.permute(2, 0, 1, 3), dqkv5[:, :, 1].permute(2, 0, 1, 3))             # dK
        dq2d = dqkv.view(B * L, 3 * d)[:, :d]
        tri = c.flash and self.use_flash_bwd  # the fused backward's dT is zero above the causal diagonal (dist > i): skip those k-tiles
        fused_dq = tri and nd == L and ops.relattn_dqr_supported(B, L, H, D, self.compute_dtype)
        if fused_dq:
            # dq_r streamed out of dT once, added onto dq_k in the same kernel's epilogue together with the u / v gradients' column sums
            if uv_parts is not None:      # (gradient accumulation: the two column-sum reduces happen once per optimizer step, WgradStash.alloc_parts)
                assert ng == 1
                ops.relattn_dqr_fused_parts(dT, R, dqkv5[:, :, 0], uv_parts)
            elif ng > 1:
                ops.relattn_dqr_fused_groups(dT, Rg, dqkv5[:, :, 0], self._bias_grad("r_w_bias", i).view(-1), self._bias_grad("r_r_bias", i).view(-1))
            else:
                ops.relattn_dqr_fused(dT, R, dqkv5[:, :, 0], self._bias_grad("r_w_bias", i).view(-1), self._bias_grad("r_r_bias", i).view(-1))
        else:
            if uv_parts is not None or ng > 1:
                raise RuntimeError("the partial-sum stash / the deferred backward were planned for the dq_r stream kernel, which this backward did not take")
            dqv = self._new(B, L, H, D)
            ops.gemm_batched(dT, R.view(nd, H, D).permute(1, 0, 2).unsqueeze(1).expand(H, B, nd, D), dqv.permute(2, 0, 1, 3),
                             tri=(1, 0) if tri else (0, 0))                                                               # dq_r
        if ng > 1:       # dR of every micro-step: the same batched product with (head, micro-step) as the two batch dimensions
            assert dR_out is None and nd == L
            Bm = B // ng
            dR = self._new(ng * nd, d)
            ops.gemm_batched(dT.view(H, ng, Bm * L, nd).transpose(2, 3), qv.view(ng, Bm * L, H, D).permute(2, 0, 1, 3),
                             dR.view(ng, nd, H, D).permute(2, 0, 1, 3), tri=(2, L) if tri else (0, 0))
        else:
            dR = self._new(nd, d) if dR_out is None else dR_out
            ops.gemm_batched(dT.view(H, B * L, nd).transpose(1, 2).unsqueeze(1), qv.view(B * L, H, D).permute(1, 0, 2).unsqueeze(1),
                             dR.view(nd, H, D).permute(1, 0, 2).unsqueeze(1), tri=(2, L) if tri else (0, 0))
        if not fused_dq:
            # dq = dq_k + dq_r, du = colsum(dq_k), dv_bias = colsum(dq_r): one pass over the two matrices
            ops.add2d_colsums(dqv.view(B * L, d), dq2d, dq2d, self._bias_grad("r_r_bias", i).view(-1), self._bias_grad("r_w_bias", i).view(-1))
        return dqkv, dR

    # ------------------------------------------------------------------ one decoder layer (post-LN; transformer_xl.py:112-353)
    def _layer_fwd(self, i, x, R_in, B, L, mlen, shift, mem, keep: bool, dec=None, dstep=None, pend=None):
        d, di, dff = self.d_model, self.d_inner, self.d_ff
        a = 1.0 if self.deepnorm_alpha is None else self.deepnorm_alpha
        p = f"h.{i}."
        c = _Ctx() if keep else None
        T = B * L
        st = self._stash(T, keep) if (dec is None and mem is None) else None   # deferred weight gradients: inputs of the four linear maps live in the stash
        if st is not None and x.data_ptr() != st.xs(i, "qkv").data_ptr():     # (layer 0: the embedding output; later layers were written there directly)
            sx = st.xs(i, "qkv")
            sx.copy_(x)
            x = sx
        win = self._win_on and dec is None and mem is None       # deferred backward: everything the backward reads goes to the window's buffers (_keep)
        if win:
            wx = self._keep(("x", i), T, d)
            if x.data_ptr() != wx.data_ptr():                    # (layer 0: the embedding output; later layers were written there directly)
                wx.copy_(x)
                x = wx
        if dec is not None:  # K/V-cached inference: only the new tokens are projected (identical maths: qkv_net has no bias)
            qkv = self._new(T, 3 * d)
            if pend is not None:   # the previous layer left its closing LayerNorm to this projection, which also stores the rows to x
                ops.linear_decode(pend.y, self.W(p + "dec_attn.qkv_net.weight"), None, qkv, pre=(pend.res, pend.alpha, pend.gamma, pend.beta, pend.eps, x))
            elif self._decode_fused_ok(T, keep, dstep) and ops.linear_decode_supported(T, 3 * d, d, False, False, False):
                ops.linear_decode(x, self.W(p + "dec_attn.qkv_net.weight"), None, qkv)    # (<= 64 rows: a stream over W, not a 256-row tile GEMM)
            else:
                ops.gemm(x, self.W(p + "dec_attn.qkv_net.weight").t(), qkv)
            av = self._attention_decode(qkv, i, B, L, mlen, shift, dec)
        else:
            if mem is not None:
                cat = torch.cat([mem.to(self.compute_dtype), x.view(B, L, d)], dim=1).contiguous()  # data movement only (:125)
                Lk = cat.shape[1]
                xin = cat.view(B * Lk, d)
            else:
                Lk, xin = L, x
            qkv = self._keep(("qkv", i), B * Lk, 3 * d) if win else self._new(B * Lk, 3 * d)
            quv = None
            Wqkv = self.W(p + "dec_attn.qkv_net.weight")
            if (mem is None and shift >= 1 and self.use_flash and self.use_flash_bwd and self.compute_dtype == torch.bfloat16 and
                    not (self.dropattn > 0 and dstep is not None) and self.use_headbias_epilogue and ops.relattn_flash_supported(B, L, self.n_head, self.d_head, self.compute_dtype) and
                    ops.gemm_nt_headbias_supported(T, 3 * d, d, d)):
                # q + r_w_bias and q + r_r_bias leave the projection's accumulators directly (the q columns of qkv stay unwritten)
                quv = (self._keep(("qu", i), B, L, self.n_head, self.d_head), self._keep(("qv", i), B, L, self.n_head, self.d_head))
                # (the NN form against a transposed weight copy is 8 % faster alone at this shape and equal inside the step: measured in round 5,
                #  profiles/r05_nt_vs_nn.txt, and removed)
                ops.gemm_nt_headbias(xin, Wqkv, qkv, quv[0], quv[1], self._bias("r_w_bias", i), self._bias("r_r_bias", i), d)
            else:
                ops.gemm(xin, Wqkv.t(), qkv)
            if self._R_all is not None:      # r_net of all layers as one batched product at the start of the forward (_rnet_all)
                R = self._R_all[self.n_layer - 1 - i]
            else:
                R = self._keep(("R", i), R_in.shape[0], d) if win else self._new(R_in.shape[0], d)
                ops.gemm(R_in, self.W(p + "dec_attn.r_net.weight").t(), R)
            av = self._attention_fwd(qkv, R, i, B, L, Lk, mlen, shift, c, quv=quv, dstep=dstep,
                                     av_out=st.xs(i, "o") if st is not None else (self._keep(("av", i), T, d) if win else None))
        if dec is not None and self._decode_fused_ok(T, keep, dstep):
            # few new tokens: every launch is latency, so the linear maps do the layer's small follow-up work themselves (db1_linear_decode):
            # GEGLU in the epilogue, and the residual LayerNorms either on the way IN to the next linear map (<= 16 tokens: no launch, no
            # hand-off between workgroups; the layer's own closing LayerNorm is left to the next layer's qkv projection) or by the last
            # workgroup on the way out.  9 launches per layer -> 5.
            eps = self.layer_norm_epsilon
            g1, b1 = self.W(p + "dec_attn.layer_norm.weight"), self.W(p + "dec_attn.layer_norm.bias")
            g2, b2 = self.W(p + "pos_ff.layer_norm.weight"), self.W(p + "pos_ff.layer_norm.bias")
            o, h1, act, f = self._new(T, d), self._new(T, d), self._new(T, dff), self._new(T, d)
            if self._decode_ln_prologue_ok(T):
                if getattr(dec, "partials", False):
                    ops.linear_decode_attn(av, mlen + L, B, L, self.n_head, self.d_head, self.W(p + "dec_attn.o_net.weight"), o)
                else:
                    ops.linear_decode(av.view(T, d), self.W(p + "dec_attn.o_net.weight"), None, o)
                ops.linear_decode(o, self.W(p + "pos_ff.CoreNet.0.weight"), self.W(p + "pos_ff.CoreNet.0.bias"), act, geglu=True,
                                  pre=(x, a, g1, b1, eps, h1))
                ops.linear_decode(act, self.W(p + "pos_ff.CoreNet.2.weight"), self.W(p + "pos_ff.CoreNet.2.bias"), f)
                return _PendingLN(res=h1, y=f, alpha=a, gamma=g2, beta=b2, eps=eps), None
            # (17 .. 64 tokens: the LayerNorms as their own launches -- finishing them inside the linear map by the last workgroup to arrive,
            #  ln= of ops.linear_decode, measured slower inside a graph than the launch boundary it saves)
            stat = lambda: self._new(T, dtype=torch.float32)
            ops.linear_decode(av.view(T, d), self.W(p + "dec_attn.o_net.weight"), None, o)
            ops.layernorm_residual_fwd(x, o, a, g1, b1, h1, None, stat(), stat(), eps)
            ops.linear_decode(h1, self.W(p + "pos_ff.CoreNet.0.weight"), self.W(p + "pos_ff.CoreNet.0.bias"), act, geglu=True)
            out = self._new(T, d)
            ops.linear_decode(act, self.W(p + "pos_ff.CoreNet.2.weight"), self.W(p + "pos_ff.CoreNet.2.bias"), f)
            ops.layernorm_residual_fwd(h1, f, a, g2, b2, out, None, stat(), stat(), eps)
            return out, None
        o = self._keep(("s1", i), T, d)
        ops.gemm(av.view(T, d), self.W(p + "dec_attn.o_net.weight").t(), o)
        h1 = self._keep(("h1", i), T, d) if st is None else st.xs(i, "ff1")
        m1, r1 = self._keep(("m1", i), T, dtype=torch.float32), self._keep(("r1", i), T, dtype=torch.float32)
        ops.layernorm_residual_fwd(x, o, a, self.W(p + "dec_attn.layer_norm.weight"), self.W(p + "dec_attn.layer_norm.bias"),
                                   h1, o if keep else None, m1, r1, self.layer_norm_epsilon,
                                   drop=self._drop_args(self.drop_p, 4 * i, dstep))  # s1 = a x + dropout(o) overwrites o
        z, act = self._ff1_fwd(h1, p, T, act=st.xs(i, "ff2") if st is not None else (self._keep(("act", i), T, dff) if win else None), keep=keep,
                               z=self._keep(("z", i), T, di) if win else None)
        f = self._keep(("s2", i), T, d)
        ops.gemm(act, self.W(p + "pos_ff.CoreNet.2.weight").t(), f, bias=self.W(p + "pos_ff.CoreNet.2.bias"))
        if st is not None and i + 1 < self.n_layer:
            out = st.xs(i + 1, "qkv")                 # the next layer's input, where its weight gradient will look for it
        else:
            out = self._keep(("x", i + 1), T, d) if (win and i + 1 < self.n_layer) else self._new(T, d)
        m2, r2 = self._keep(("m2", i), T, dtype=torch.float32), self._keep(("r2", i), T, dtype=torch.float32)
        ops.layernorm_residual_fwd(h1, f, a, self.W(p + "pos_ff.layer_norm.weight"), self.W(p + "pos_ff.layer_norm.bias"),
                                   out, f if keep else None, m2, r2, self.layer_norm_epsilon,
                                   drop=self._drop_args(self.drop_p, 4 * i + 1, dstep))
        if keep:
            c.x, c.qkv, c.R, c.av, c.s1, c.m1, c.r1 = x, qkv, R, av, o, m1, r1
            c.h1, c.z, c.act, c.s2, c.m2, c.r2 = h1, z, act, f, m2, r2
        return out, c

    def _stash(self, T: int, keep: bool) -> Optional[WgradStash]:
        """the weight-gradient stash when this training forward / backward uses it (post-LN layers, bf16 or fp32, matching token count)"""
        st = self.wgrad_stash
        if st is None or not keep or not self.training or self.pre_lnorm or st.T != T or self.wgrad_defer_ga <= 1:
            return None
        return st

    # ---- R_i = r_net_i(position table) for ALL layers in one launch (transformer_xl.py:138: the table is the same for every layer and
    # the product does not depend on the batch).  One layer's product is 1024 x 2048 x 2048 -- 32 tiles, 35 us on a nearly empty chip,
    # 24 times per forward (and per micro-step: 13 ms per optimizer step at micro-batch 4 x GA 16); batched over the layers it is 768 tiles.
    # The 24 weights sit at a regular stride in the arena (layers are laid out one after the other, last layer first).
    def _rnet_all(self, R_in: torch.Tensor):
        n, d = self.n_layer, self.d_model
        if not self.use_rnet_batched or n < 2 or self.compute_dtype != torch.bfloat16 or R_in.shape[0] % 256 or d % 256:
            return None
        offs = [self.arena.offsets[f"h.{i}.dec_attn.r_net.weight"][0] for i in range(n)]
        stride = offs[n - 2] - offs[n - 1]               # (layer n-1 comes first)
        if stride <= 0 or any(offs[i] - offs[i + 1] != stride for i in range(n - 1)):
            return None
        W = torch.as_strided(self.arena.work, (n, 1, d, d), (stride, 0, 1, d), offs[n - 1])      # [layer n-1-j][k][n] = W_j[n][k]
        nd = R_in.shape[0]
        out = self._keep(("Rall",), n, 1, nd, d)
        ops.gemm_batched(R_in.view(1, 1, nd, d).expand(n, 1, nd, d), W, out)
        return out.view(n, nd, d)

    # ---- PositionwiseFF halves with the activation inside the GEMM where the shape allows (db1_gemm_nt_geglu / db1_gemm_nn_geglu_bwd: the
    # "bias + GEGLU" epilogue of SURVEY 8b; otherwise the same arithmetic as separate launches)
    def _ff1_fwd(self, x, p, T, act=None, keep=False, z=None):
        """z = x W1^T + b1, act = GEGLU(z)  (transformer_xl.py:264-266, activations.py:19-32)"""
        d, di, dff = self.d_model, self.d_inner, self.d_ff
        z = self._new(T, di) if z is None else z
        act = self._new(T, dff) if act is None else act
        W1, b1 = self.W(p + "pos_ff.CoreNet.0.weight"), self.W(p + "pos_ff.CoreNet.0.bias")
        if self.use_geglu_epilogue and self.activation_fn == "geglu" and ops.gemm_nt_geglu_fused(T, dff, d, self.compute_dtype):
            ops.gemm_nt_geglu(x, W1, b1, z, act)
        else:
            ops.gemm(x, W1.t(), z, bias=b1)
            ops.ffn_act_fwd(z, act, self.activation_fn)
        return z, act

    def _ff2_dgrad(self, df, z, p, T, dz=None, parts=None):
        """dz from df = d(loss)/d(CoreNet output): dact = df W2, through the activation; accumulates the first bias's gradient"""
        d, di, dff = self.d_model, self.d_inner, self.d_ff
        dz = self._new(T, di) if dz is None else dz
        W2, gb1 = self.W(p + "pos_ff.CoreNet.2.weight"), self.G(p + "pos_ff.CoreNet.0.bias")
        if parts is not None:
            ops.gemm_nn_geglu_bwd_parts(df, W2, z, dz, parts)
        elif self.use_geglu_epilogue and self.activation_fn == "geglu" and ops.gemm_nn_geglu_bwd_fused(T, dff, d, self.compute_dtype):
            ops.gemm_nn_geglu_bwd(df, W2, z, dz, gb1)
        else:
            dact = self._new(T, dff)
            ops.gemm(df, W2, dact)
            ops.ffn_act_bwd_bias(z, dact, dz, gb1, self.activation_fn)
        return dz

    # ---- pre-LN ordering of the same kernels (config default `--pre-lnorm True`; transformer_xl.py:126-137,231-233,277-282)
    def _layer_fwd_prelnorm(self, i, x, R_in, B, L, mlen, shift, mem, keep: bool, dec=None, dstep=None):
        d, di, dff = self.d_model, self.d_inner, self.d_ff
        p = f"h.{i}."
        c = _Ctx() if keep else None
        T = B * L
        if mem is not None and dec is None:
            cat = torch.cat([mem.to(self.compute_dtype), x.view(B, L, d)], dim=1).contiguous()
            Lk = cat.shape[1]
            xin = cat.view(B * Lk, d)
        else:
            Lk, xin = L, x
        hin = self._new(B * Lk, d)
        m1, r1 = self._new(B * Lk, dtype=torch.float32), self._new(B * Lk, dtype=torch.float32)
        ops.layernorm_residual_fwd(xin, None, 1.0, self.W(p + "dec_attn.layer_norm.weight"), self.W(p + "dec_attn.layer_norm.bias"),
                                   hin, None, m1, r1, self.layer_norm_epsilon)
        qkv = self._new(B * Lk, 3 * d)
        ops.gemm(hin, self.W(p + "dec_attn.qkv_net.weight").t(), qkv)
        if dec is not None:  # K/V-cached inference (the cache holds W_kv . LN(mem): LayerNorm is per token)
            av = self._attention_decode(qkv, i, B, L, mlen, shift, dec)
        else:
            R = self._new(R_in.shape[0], d)
            ops.gemm(R_in, self.W(p + "dec_attn.r_net.weight").t(), R)
            av = self._attention_fwd(qkv, R, i, B, L, Lk, mlen, shift, c, dstep=dstep)
        o = self._new(T, d)
        ops.gemm(av.view(T, d), self.W(p + "dec_attn.o_net.weight").t(), o)
        if dstep is not None and self.drop_p > 0:
            ops.dropout(o, o, self._drop_args(self.drop_p, 4 * i, dstep))      # :229
        h1 = self._new(T, d)
        ops.add(x, o, h1)                                                      # residual (:233)
        fin = self._new(T, d)
        m2, r2 = self._new(T, dtype=torch.float32), self._new(T, dtype=torch.float32)
        ops.layernorm_residual_fwd(h1, None, 1.0, self.W(p + "pos_ff.layer_norm.weight"), self.W(p + "pos_ff.layer_norm.bias"),
                                   fin, None, m2, r2, self.layer_norm_epsilon)
        z, act = self._ff1_fwd(fin, p, T, keep=keep)
        out = self._new(T, d)
        ops.gemm(act, self.W(p + "pos_ff.CoreNet.2.weight").t(), out, bias=self.W(p + "pos_ff.CoreNet.2.bias"))
        if dstep is not None and self.drop_p > 0:
            ops.dropout(out, out, self._drop_args(self.drop_p, 4 * i + 1, dstep))  # CoreNet's trailing Dropout (:262-269)
        ops.add(out, h1, out)                                                  # residual (:282)
        if keep:
            c.x, c.hin, c.qkv, c.R, c.av, c.m1, c.r1 = x, hin, qkv, R, av, m1, r1
            c.h1, c.fin, c.z, c.act, c.m2, c.r2 = h1, fin, z, act, m2, r2
        return out, c

    def _layer_bwd_prelnorm(self, i, dout, c: _Ctx, R_in, B, L, shift, dstep=None):
        d, di, dff = self.d_model, self.d_inner, self.d_ff
        p = f"h.{i}."
        T = B * L
        W, G = self.W, self.G
        df = dout
        if dstep is not None and self.drop_p > 0:                              # gradient through the feed-forward output's dropout
            df = self._new(T, d)
            ops.dropout(dout, df, self._drop_args(self.drop_p, 4 * i + 1, dstep))
        ops.gemm(df.t(), c.act, G(p + "pos_ff.CoreNet.2.weight"), beta=self._gb)
        ops.colsum_acc(df, G(p + "pos_ff.CoreNet.2.bias"))
        dz = self._ff2_dgrad(df, c.z, p, T)
        ops.gemm(dz.t(), c.fin, G(p + "pos_ff.CoreNet.0.weight"), beta=self._gb)
        dfin = self._new(T, d)
        ops.gemm(dz, W(p + "pos_ff.CoreNet.0.weight"), dfin)
        dh1 = self._new(T, d)
        ops.layernorm_residual_bwd(dfin, c.h1, W(p + "pos_ff.layer_norm.weight"), c.m2, c.r2, dh1,
                                   G(p + "pos_ff.layer_norm.weight"), G(p + "pos_ff.layer_norm.bias"))
        ops.add(dh1, dout, dh1)                                                # + the residual branch
        do = dh1
        if dstep is not None and self.drop_p > 0:                              # gradient through the attention output's dropout
            do = self._new(T, d)
            ops.dropout(dh1, do, self._drop_args(self.drop_p, 4 * i, dstep))
        ops.gemm(do.t(), c.av.view(T, d), G(p + "dec_attn.o_net.weight"), beta=self._gb)
        dav = self._new(T, d)
        ops.gemm(do, W(p + "dec_attn.o_net.weight"), dav)
        dqkv, dR = self._attention_bwd(dav, c, i, B, L, shift, dstep)
        ops.gemm(dR.t(), R_in, G(p + "dec_attn.r_net.weight"), beta=self._gb)
        ops.gemm(dqkv.t(), c.hin, G(p + "dec_attn.qkv_net.weight"), beta=self._gb)
        dhin = self._new(T, d)
        ops.gemm(dqkv, W(p + "dec_attn.qkv_net.weight"), dhin)
        dx = self._new(T, d)
        ops.layernorm_residual_bwd(dhin, c.x, W(p + "dec_attn.layer_norm.weight"), c.m1, c.r1, dx,
                                   G(p + "dec_attn.layer_norm.weight"), G(p + "dec_attn.layer_norm.bias"))
        ops.add(dx, dh1, dx)
        return dx

    def _layer_bwd(self, i, dout, c: _Ctx, R_in, B, L, shift, dstep=None, flush=True):
        d, di, dff = self.d_model, self.d_inner, self.d_ff
        a = 1.0 if self.deepnorm_alpha is None else self.deepnorm_alpha
        p = f"h.{i}."
        T = B * L
        W, G = self.W, self.G
        dropping = dstep is not None and self.drop_p > 0
        st = self._stash(T, True)   # deferred weight gradients: the four output gradients go to the stash, their products run once per optimizer step
        # ---- feed-forward.  s2 = a h1 + dropout(f): the residual branch takes ds2 as it is, the feed-forward branch takes it under
        # the forward's keep decisions (df, written by the same kernel from the same registers)
        ds2 = self._new(T, d)
        if st is not None:
            df = st.dys(i, "ff2")
        else:
            df = self._new(T, d) if dropping else ds2
        parts = st is not None and st.parts
        if parts:
            ops.layernorm_residual_bwd_parts(dout, c.s2, W(p + "pos_ff.layer_norm.weight"), c.m2, c.r2, ds2 if dropping else df, st.ln_parts[i][0][st.slot],
                                             dr_out=df if dropping else None, drop=self._drop_args(self.drop_p, 4 * i + 1, dstep))
        else:
            ops.layernorm_residual_bwd(dout, c.s2, W(p + "pos_ff.layer_norm.weight"), c.m2, c.r2, ds2 if (dropping or st is None) else df,
                                       G(p + "pos_ff.layer_norm.weight"), G(p + "pos_ff.layer_norm.bias"),
                                       dr_out=df if dropping else None, drop=self._drop_args(self.drop_p, 4 * i + 1, dstep))
        if st is not None and not dropping:
            ds2.copy_(df)                # (nothing dropped: df = ds2; the stash keeps it, the in-place dh1 below needs its own copy)
        if st is None:
            ops.gemm(df.t(), c.act, G(p + "pos_ff.CoreNet.2.weight"), beta=self._gb)
        if st is None:      # (with the stash df of every micro-step is kept: the second bias's gradient is ONE column sum per layer at the flush)
            ops.colsum_acc(df, G(p + "pos_ff.CoreNet.2.bias"))
        dz = self._ff2_dgrad(df, c.z, p, T, dz=None if st is None else st.dys(i, "ff1"), parts=st.b1_parts[i][st.slot] if parts else None)
        if st is None:
            ops.gemm(dz.t(), c.h1, G(p + "pos_ff.CoreNet.0.weight"), beta=self._gb)
        ops.gemm(dz, W(p + "pos_ff.CoreNet.0.weight"), ds2, beta=a)          # dh1 = a*ds2 + dz W1   (in place over ds2)
        dh1 = ds2
        # ---- attention
        ds1 = self._new(T, d)
        if st is not None:
            do = st.dys(i, "o")
        else:
            do = self._new(T, d) if dropping else ds1
        if parts:
            ops.layernorm_residual_bwd_parts(dh1, c.s1, W(p + "dec_attn.layer_norm.weight"), c.m1, c.r1, ds1 if dropping else do, st.ln_parts[i][1][st.slot],
                                             dr_out=do if dropping else None, drop=self._drop_args(self.drop_p, 4 * i, dstep))
        else:
            ops.layernorm_residual_bwd(dh1, c.s1, W(p + "dec_attn.layer_norm.weight"), c.m1, c.r1, ds1 if (dropping or st is None) else do,
                                       G(p + "dec_attn.layer_norm.weight"), G(p + "dec_attn.layer_norm.bias"),
                                       dr_out=do if dropping else None, drop=self._drop_args(self.drop_p, 4 * i, dstep))
        if st is not None and not dropping:
            ds1.copy_(do)
        if st is None:
            ops.gemm(do.t(), c.av.view(T, d), G(p + "dec_attn.o_net.weight"), beta=self._gb)
        dav = self._new(T, d)
        ops.gemm(do, W(p + "dec_attn.o_net.weight"), dav)
        defer_r = st is not None and st.nd == R_in.shape[0] and R_in.data_ptr() == st.rins().data_ptr()   # (the forward put this micro-step's table into the stash)
        dqkv, dR = self._attention_bwd(dav, c, i, B, L, shift, dstep, dqkv_out=None if st is None else st.dys(i, "qkv"),
                                       dR_out=st.drs(i) if defer_r else None, uv_parts=st.uv_parts[i][st.slot] if parts else None)
        if not defer_r:
            ops.gemm(dR.t(), R_in, G(p + "dec_attn.r_net.weight"), beta=self._gb)
        if st is None:
            ops.gemm(dqkv.t(), c.x, G(p + "dec_attn.qkv_net.weight"), beta=self._gb)
        ops.gemm(dqkv, W(p + "dec_attn.qkv_net.weight"), ds1, beta=a)        # dx = a*ds1 + dqkv Wqkv (in place over ds1)
        if st is not None and flush:
            self._flush_layer_wgrads(i, st)
        return ds1

    def _flush_layer_wgrads(self, i: int, st: WgradStash):
        """the four weight gradients of layer i over every micro-step stashed so far: dW = dy^T x, K = (slot + 1) * T rows"""
        self._flush_layer_rows(i, st, st.first * st.T, (st.slot + 1) * st.T)

    def _grad_pair(self, wname: str, bname: str) -> torch.Tensor:
        """the gradients of a LayerNorm's (weight | bias) as ONE [2 d] accumulator: the two are neighbours in the arena"""
        ow, sw, _ = self.arena.offsets[wname]
        ob, sb, _ = self.arena.offsets[bname]
        n = int(np.prod(sw))
        assert ob == ow + n and int(np.prod(sb)) == n, (wname, bname)
        return self.arena.grad[ow:ow + 2 * n]

    def _flush_layer_rows(self, i: int, st: WgradStash, lo: int, hi: int):
        p = f"h.{i}."
        for kind, name in (("ff2", "pos_ff.CoreNet.2.weight"), ("ff1", "pos_ff.CoreNet.0.weight"), ("o", "dec_attn.o_net.weight"),
                           ("qkv", "dec_attn.qkv_net.weight")):
            ops.gemm(st.dy[i][kind][lo:hi].t(), st.x[i][kind][lo:hi], self.G(p + name), beta=st.beta)
        ops.colsum_acc(st.dy[i]["ff2"][lo:hi], self.G(p + "pos_ff.CoreNet.2.bias"))     # the feed-forward output bias: column sums of the stashed df
        if st.parts:       # the small reductions of micro-steps [s0, s1): one column sum each over all their partial rows
            s0, s1, d = lo // st.T, hi // st.T, self.d_model
            for site, ln in enumerate(("pos_ff.layer_norm", "dec_attn.layer_norm")):
                ops.colsum_acc(st.ln_parts[i][site][s0:s1].view(-1, 2 * d), self._grad_pair(p + ln + ".weight", p + ln + ".bias"))
            uv = st.uv_parts[i][s0:s1]
            ops.colsum_acc(uv[:, 0].reshape(-1, uv.shape[-1]), self._bias_grad("r_w_bias", i).view(-1))
            ops.colsum_acc(uv[:, 1].reshape(-1, uv.shape[-1]), self._bias_grad("r_r_bias", i).view(-1))
            ops.colsum_acc(st.b1_parts[i][s0:s1].view(-1, 2 * self.d_ff), self.G(p + "pos_ff.CoreNet.0.bias"))
        if st.nd and st.r_used:      # r_net over the same micro-steps: rows [slot * nd, (slot + 1) * nd) of the position-table stash
            r0, r1 = (lo // st.T) * st.nd, (hi // st.T) * st.nd
            ops.gemm(st.dr[i][r0:r1].t(), st.rin[r0:r1], self.G(p + "dec_attn.r_net.weight"), beta=st.beta)

    def flush_deferred_wgrads(self):
        """(a hipGraph-captured boundary micro-step stashes like the others; the products then run here, outside the graph)"""
        st = self.wgrad_stash
        if st is None:
            return
        with torch.cuda.device(self.dev), ops.stream_scope():
            for i in reversed(range(self.n_layer)):
                self._flush_layer_wgrads(i, st)

    # ------------------------------------------------------------------ public API
    def init_mem(self, batch_size):
        """transformer_xl.py:470-485"""
        if self.mem_len > 0:
            return [torch.zeros(batch_size, self.mem_len, self.n_embed, dtype=self.compute_dtype, device=self.dev)
                    for _ in range(self.n_layer)]
        return None

    def forward(self, tasks_input: Sequence, compute_loss: bool = True, mems=None):
        """transformer_xl.py:506-619.  Runs on ``self.dev`` whatever the caller's current device is (streams, workspaces and the
        per-device kernel attributes all follow the current device)."""
        with torch.cuda.device(self.dev), ops.stream_scope():
            return self._forward(tasks_input, compute_loss, mems)

    def _forward(self, tasks_input: Sequence, compute_loss: bool = True, mems=None):
        assert not (compute_loss and mems is not None), "During training, Gato does not use memory mechanism."
        keep = compute_loss and torch.is_grad_enabled()
        d = self.d_model
        embs, labels, masks, ecs, shapes = [], [], [], [], []
        for t in tasks_input:
            e, lab, msk, c = self._embed_task(t, compute_loss)
            embs.append(e); ecs.append(c); shapes.append((e.shape[0], e.shape[1]))
            if compute_loss:
                labels.append(lab); masks.append(msk)
        h = embs[0] if len(embs) == 1 else torch.cat(embs, dim=0)  # concat on the batch dim (:541-545): data movement only
        B, L, _ = h.shape
        dstep = None
        if self.training and mems is None and (self.drop_p > 0 or self.embd_pdrop > 0 or self.dropattn > 0):
            if self._drop_step_dev is not None:      # graph mode: step = 0 + the device counter (bumped by the captured graph itself)
                dstep = 0
            else:
                self._drop_step += 1
                dstep = self._drop_step
            if self.embd_pdrop > 0:
                ops.dropout(h, h, self._drop_args(self.embd_pdrop, self.SITE_EMBED, dstep))                     # :545
        ring = mems if (mems is not None and not isinstance(mems, (list, tuple))) else None   # decode.RingMemory: K / V ring instead of hidden states
        if ring is not None:
            mems = None
            mlen = int(self.mem_len)
        else:
            mlen = mems[0].size(1) if mems is not None else 0
        klen = L + mlen
        shift = self._window(L, mlen)
        # The reference builds a uint8 mask (1 = hidden, transformer_xl.py:551-567) and raises ValueError when NOTHING is hidden
        # (`torch.sum(attention_mask).item()` is 0, :177,205-206): the causal part triu(1 + mlen) hides something iff qlen > 1, the
        # same_length part tril(-shift) iff shift <= qlen - 1.  (So a 1-token call without a full memory raises there, and here.)
        # The opposite extreme -- every key of a row hidden, e.g. same_length with mem_len = 0 -- does NOT raise in the reference:
        # such rows attend uniformly to all keys; the materialised kernels reproduce that (db1_relattn_softmax_fwd).
        if not (L > 1 or (self.same_length and shift <= L - 1)):
            raise ValueError("attention mask hides nothing (transformer_xl.py:177,205-206)")
        dec = self._decode_begin(mems, B, L, mlen) if (mems is not None and mlen > 0) else None
        if ring is not None:
            self.check_decode_chain()   # (a failed chain launch of an earlier call: raise before more tokens go into the same memory)
            if not (self.use_decode and self.compute_dtype == torch.bfloat16 and ring.B == B and L <= 64 and mlen + L <= ring.cap):
                raise ValueError("RingMemory needs the bf16 decode path, its own batch size and at most 64 new tokens per call")
            dec = SimpleNamespace(ring=ring, R=self._decode_R(), kv=None, new_kv=[])
            dec.partials = (self.use_decode_attn_partials and not self.pre_lnorm and self._decode_fused_ok(B * L, keep, dstep) and
                            self._decode_ln_prologue_ok(B * L) and ops.linear_decode_attn_supported(B, L, self.n_head, self.d_head, mlen + L, d))
        R_in = self._sinusoid(klen) if dec is None else None
        if dstep is not None and self.embd_pdrop > 0:   # the position table goes through the same nn.Dropout (:575); the cached table stays intact
            R_drop = torch.empty_like(R_in)
            ops.dropout(R_in, R_drop, self._drop_args(self.embd_pdrop, self.SITE_POS, dstep))
            R_in = R_drop
        x = h.view(B * L, d)
        self._win_on = False
        if self.bwd_window_ga > 1 and keep and self.training and mems is None and ring is None:
            R_in = self._window_begin(B, L, shift, R_in, dstep)      # (deferred backward: this micro-step's position table joins the window's)
        if (ring is not None and B * L == 1 and self.use_decode_chain and getattr(dec, "partials", False) and self.activation_fn == "geglu" and self.n_layer >= 2 and
                ops.decode_chain_supported(d, self.d_ff, self.n_head, self.d_head, mlen + L)):
            x = self._decode_chain_layers(x, mlen, shift, dec)
            hids, lcs = [], []
            return self._finish_forward(x, hids, lcs, [], [], [], [], R_in, B, L, shift, dstep, keep, compute_loss, mems, ring, dec, mlen)
        if self.wgrad_defer_ga > 1 and keep and self.training and not self.pre_lnorm and mems is None and not self._win_on:
            st = self.wgrad_stash
            if not 0 <= self._wg_slot < self.wgrad_defer_ga:
                raise RuntimeError(f"weight-gradient stash: micro-step {self._wg_slot} of an accumulation window of {self.wgrad_defer_ga}")
            if st is None or st.T != B * L or st.ga != self.wgrad_defer_ga or st.nd != int(R_in.shape[0]):
                first, beta = 0, 0.0
                if st is not None and self._wg_slot > 0:
                    # the token count changed INSIDE an accumulation window (a short last batch, another task mix): the operands of the
                    # micro-steps done so far live in the old stash.  Form their weight gradients now (rows [first, slot) x T_old), then
                    # go on in a stash of the new size that starts at this micro-step and accumulates onto what was just written.
                    if st.first < self._wg_slot:
                        for i in reversed(range(self.n_layer)):
                            self._flush_layer_rows(i, st, st.first * st.T, self._wg_slot * st.T)
                        first, beta = self._wg_slot, 1.0
                    else:
                        first, beta = self._wg_slot, st.beta
                self.wgrad_stash = st = None     # (free the old buffers first)
                st = self.wgrad_stash = WgradStash(self, B * L, self.wgrad_defer_ga, nd=int(R_in.shape[0]))
                st.first, st.beta = first, beta
                if B * L == B * int(R_in.shape[0]):      # (plain causal training batch: nd = L)
                    st.alloc_parts(self)
            elif self._wg_slot == 0:
                st.first = 0
                st.r_used = False
                if B * L == B * int(R_in.shape[0]) and not torch.cuda.is_current_stream_capturing():
                    st.revalidate_parts(self)       # (the plan follows the switches; a captured step keeps what its warm-up decided)
            st.slot = self._wg_slot
            rows = st.rins()                 # this micro-step's position table (after its dropout) lives in the stash: r_net's input
            rows.copy_(R_in)
            R_in = rows
            st.r_used = True
        hids, lcs = [], []
        self._R_all = self._rnet_all(R_in) if (dec is None and mems is None and not self.pre_lnorm) else None
        for i in range(self.n_layer):
            kw = {}
            if isinstance(x, _PendingLN):   # (inference, <= 16 new tokens) this layer's qkv projection normalises its input rows and stores them to x
                kw["pend"], x = x, self._new(B * L, d)
            hids.append(x)
            layer_fwd = self._layer_fwd_prelnorm if self.pre_lnorm else self._layer_fwd
            x, c = layer_fwd(i, x, R_in, B, L, mlen, shift, None if mems is None else mems[i], keep, dec, dstep, **kw)
            lcs.append(c)
        self._R_all = None       # (the layers' contexts hold their slices)
        try:
            return self._finish_forward(x, hids, lcs, ecs, shapes, labels, masks, R_in, B, L, shift, dstep, keep, compute_loss, mems, ring, dec, mlen)
        finally:
            self._win_on = False

    # ---- deferred backward (BackwardWindow): the forwards of an accumulation window write into window-sized buffers, ONE backward at its boundary
    def _window_ok(self, B: int, L: int, shift: int, nd: int) -> bool:
        ga, H, D, dt = self.bwd_window_ga, self.n_head, self.d_head, self.compute_dtype
        return (not self.pre_lnorm and dt == torch.bfloat16 and self.use_flash and self.use_flash_bwd and self.dropattn == 0 and self.fuse_head_loss and
                not self.keep_logits and self.activation_fn == "geglu" and shift >= L and nd == L and
                ops.relattn_flash_supported(B, L, H, D, dt) and ops.relattn_dqr_groups_supported(ga * B, L, H, D, dt, ga) and
                self._probs_mode(ga * B, L) == "forward")

    def _window_begin(self, B: int, L: int, shift: int, R_in: torch.Tensor, dstep):
        """called by a training forward when the engine defers the backward: joins the window (and returns the position table's copy inside it)
        or, when this micro-step cannot (another shape, an unsupported configuration), first runs the backward of what the window holds"""
        win = self._win
        if win is None or win.ga != self.bwd_window_ga:
            win = self._win = BackwardWindow(self.bwd_window_ga)
        nd = int(R_in.shape[0])
        sig = (B, L, nd)
        ok = self._window_ok(B, L, shift, nd)
        if win.n > 0 and (not ok or win.sig != sig or win.n >= win.ga):
            # a micro-step that does not fit the window in flight (a short last batch, another task mix): the window's gradients are formed
            # now, over the micro-steps it holds, and this one starts over
            self._backward_window(self.loss_grad_scale, None)
        if not ok:
            return R_in
        if win.sig != sig:
            win.reset()
            win.sig = sig
        if win.n > 0:
            c0 = win.ctxs[0]
            if (c0.dstep is None) != (dstep is None) or (dstep is not None and self._drop_step_dev is None and dstep != c0.dstep + win.n):
                raise RuntimeError("deferred backward: the dropout steps of an accumulation window must be consecutive")
        win.slot = win.n
        self._win_on = True
        rows = self._keep(("rin",), nd, self.d_model)
        rows.copy_(R_in)
        return rows

    def _decode_chain_layers(self, x, mlen, shift, dec):
        """ONE new token over the K / V ring: per layer the attention launch (chunk partials) and ONE persistent launch for everything between
        two attention launches (db1_decode_chain: o_net with the merge, LayerNorm, ff1 + GEGLU, ff2, LayerNorm, the next layer's qkv
        projection) -- 2 launches per layer instead of 5, and the weights stream without the gaps between launches"""
        d, H, D, n = self.d_model, self.n_head, self.d_head, self.n_layer
        a = 1.0 if self.deepnorm_alpha is None else self.deepnorm_alpha
        W = self.W
        qkv = self._new(1, 3 * d)
        ops.gemm(x, W("h.0.dec_attn.qkv_net.weight").t(), qkv)
        h1_out, f_out = self._new(1, d), self._new(1, d)
        for i in range(n):
            p = f"h.{i}."
            part = self._attention_decode(qkv, i, 1, 1, mlen, shift, dec)
            last = i == n - 1
            x_next = None if last else self._new(1, d)
            qkv_next = None if last else self._new(1, 3 * d)
            ops.decode_chain(part, mlen + 1, H, x, W(p + "dec_attn.o_net.weight"), W(p + "pos_ff.CoreNet.0.weight"), W(p + "pos_ff.CoreNet.0.bias"),
                             W(p + "pos_ff.CoreNet.2.weight"), W(p + "pos_ff.CoreNet.2.bias"), None if last else W(f"h.{i + 1}.dec_attn.qkv_net.weight"),
                             W(p + "dec_attn.layer_norm.weight"), W(p + "dec_attn.layer_norm.bias"), W(p + "pos_ff.layer_norm.weight"),
                             W(p + "pos_ff.layer_norm.bias"), a, self.layer_norm_epsilon, h1_out if last else None, f_out if last else None, x_next, qkv_next, i,
                             w_o_next=W(f"h.{(i + 1) % n}.dec_attn.o_net.weight"))
            if not last:
                x, qkv = x_next, qkv_next
        # the chain's hand-off polls are bounded: if one ran out (a co-tenant kernel, a CU mask: not all 256 workgroups resident) the
        # launch has set a flag and carried on with garbage.  The flag travels to pinned host memory behind the launches, and
        # check_decode_chain() -- at the next forward, in get_action after its own synchronisation, in GraphedRingStep -- raises on it.
        if os.environ.get("DB1_CHAIN_FLAG_FETCH", "1") != "0":    # (0: measurements only -- a failed hand-off then goes unnoticed)
            self._chain_watch = ops.decode_chain_flag_fetch(self.dev)
        p = f"h.{n - 1}."
        return _PendingLN(res=h1_out, y=f_out, alpha=a, gamma=W(p + "pos_ff.layer_norm.weight"),
                          beta=W(p + "pos_ff.layer_norm.bias"), eps=self.layer_norm_epsilon)

    def check_decode_chain(self, synchronize: bool = False, watch=None):
        """Raise if a persistent one-token launch (db1_decode_chain) reported that a hand-off poll ran into its limit: the logits of that
        call (and the memory it appended) are invalid.  Without ``synchronize`` this reads the flag copy of the last call the stream has
        FINISHED -- free, and exact wherever the caller has synchronised anyway (``.cpu()`` / ``.item()`` on the logits); with it, the
        stream is drained first.  The chain is switched off for this model after a failure (the per-launch path takes over); set
        ``use_decode_chain = True`` again once the device is exclusive.  ``watch``: check THAT (pinned word, scratch) pair -- a captured
        step's own (GraphedRingStep) -- and leave the model's pending watch of its eager calls alone."""
        own = watch is None
        w = self._chain_watch if own else watch
        if w is None:
            return
        if synchronize:
            torch.cuda.current_stream(self.dev).synchronize()
        if int(w[0][0]) != 0:
            self.use_decode_chain = False
            ops.decode_chain_clear_error(w)
            if own:
                self._chain_watch = None
            raise lib.Db1Error("db1_decode_chain: a hand-off poll ran into its limit (not all 256 workgroups were resident: another kernel, "
                               "stream or process shares the GPU, or a CU mask is set).  The logits and the appended memory rows of that call are "
                               "invalid; the persistent path is now off for this model (model.use_decode_chain = False), repeat the episode.")

    def _finish_forward(self, x, hids, lcs, ecs, shapes, labels, masks, R_in, B, L, shift, dstep, keep, compute_loss, mems, ring, dec, mlen):
        d = self.d_model
        head_pend = None
        if isinstance(x, _PendingLN):
            pend, x = x, self._new(B * L, d)
            if not compute_loss and ops.linear_decode_supported(B * L, self.vocab_pad, d, False, False, True):
                head_pend = pend      # the vocabulary projection normalises its input rows itself (and stores them to x)
            else:
                ops.layernorm_residual_fwd(pend.res, pend.y, pend.alpha, pend.gamma, pend.beta, x, None, self._new(B * L, dtype=torch.float32),
                                           self._new(B * L, dtype=torch.float32), pend.eps)
        Wout = self.arena.view(self.arena.work, "word_embedding.weight" if self.share_input_output_embedding else "lm_head.weight",
                               full=True).view(self.vocab_pad, d)
        T = B * L
        V = self.total_vocab_size
        loss, lm_logits = None, None
        # the fused sweep writes the head's weight gradient during the FORWARD: only forwards that are trained on take it (training mode,
        # gradients enabled); a validation pass in eval() mode goes through the logits and leaves the accumulators alone
        fused = compute_loss and keep and self.fuse_head_loss and not self.keep_logits and self.training
        if fused and self._ctx is not None and getattr(self._ctx, "dh_head", None) is not None:
            raise RuntimeError("the previous training forward (fused head + loss) was not followed by backward(): its head gradient is already in "
                               "the accumulators.  Call backward(), or zero_grad(), or run forwards that are not trained on under eval() / torch.no_grad().")
        if compute_loss:
            lab = (labels[0] if len(labels) == 1 else torch.cat(labels, dim=0)).reshape(-1).contiguous()
            msk = (masks[0] if len(masks) == 1 else torch.cat(masks, dim=0)).reshape(-1).contiguous()
            lse = self._new(T, dtype=torch.float32)
            sums = torch.zeros(2, device=self.dev, dtype=torch.float32)
        if fused:
            # loss, dh and the head's weight gradient in one sweep: the logits only ever exist 16 384 rows at a time (in the workspace)
            wname = "word_embedding.weight" if self.share_input_output_embedding else "lm_head.weight"
            gW = self.arena.view(self.arena.grad, wname, full=True).view(self.vocab_pad, d)
            dh = self._keep(("dh",), T, d)
            # (deferred backward: the head's weight gradient of the window's first micro-step writes, the later ones accumulate -- no backward in between)
            first_writer = self._grad_fresh and not (self._win_on and self._win.slot > 0)
            ops.lmhead_ce(x, Wout, lab, msk, lse, sums, V, dh=dh, dW_acc=gW, beta_dw=0.0 if first_writer else 1.0, gscale=self.loss_grad_scale,
                          chunk_rows=self.head_chunk_rows)
            loss = sums[0] / sums[1]
            ctx = _Ctx()
            ctx.ecs, ctx.shapes, ctx.lcs, ctx.R_in, ctx.dh_head = ecs, shapes, lcs, R_in, dh
            ctx.B, ctx.L, ctx.shift, ctx.dstep, ctx.fused_scale = B, L, shift, dstep, self.loss_grad_scale
            ctx.win = self._win_on
            if self._win_on:       # the layers' tensors live in the window's buffers: the per-micro-step contexts are not needed
                ctx.lcs = None
                self._win.ctxs[self._win.slot] = ctx
                self._win.n = self._win.slot + 1
            self._ctx = ctx
        else:
            logits_pad = self._new(T, self.vocab_pad)
            if head_pend is not None:
                ops.linear_decode(head_pend.y, Wout, None, logits_pad, pre=(head_pend.res, head_pend.alpha, head_pend.gamma, head_pend.beta, head_pend.eps, x))
            elif (dec is not None and not compute_loss and not keep and x.dtype == torch.bfloat16 and self.use_decode_fused and
                  ops.linear_decode_supported(T, self.vocab_pad, d, False, False, False)):
                # a few rows of an inference call with memory: the 136 MB of the tied head as one stream (the 256-row tile GEMM took 290 us for 16 rows)
                ops.linear_decode(x, Wout, None, logits_pad)
            else:
                ops.gemm(x, Wout.t(), logits_pad, useful_flops=2.0 * T * V * d)   # (the padded vocabulary columns are not counted as work)
            lm_logits = logits_pad.view(B, L, self.vocab_pad)[:, :, :V]
            if compute_loss:
                ops.masked_ce_fwd(logits_pad, lab, msk, lse, sums, V)
                loss = sums[0] / sums[1]
                if keep:
                    ctx = _Ctx()
                    ctx.ecs, ctx.shapes, ctx.lcs, ctx.R_in, ctx.hfin = ecs, shapes, lcs, R_in, x
                    ctx.logits_pad, ctx.lab, ctx.msk, ctx.lse, ctx.sums = logits_pad, lab, msk, lse, sums
                    ctx.B, ctx.L, ctx.shift, ctx.dstep, ctx.dh_head = B, L, shift, dstep, None
                    self._ctx = ctx
        res = (lm_logits, loss)
        if ring is not None:   # the ring was appended by the attention launches: only its origin moves
            ops.ring_advance(ring.state, L, ring.cap)
            return res + (ring,)
        if mems is not None:  # _update_mem (:487-504)
            end_idx = mlen + max(0, L)
            beg_idx = max(0, end_idx - self.mem_len)
            new_mems = [torch.cat([mems[i].to(self.compute_dtype), hids[i].view(B, L, d)], dim=1)[:, beg_idx:end_idx].detach()
                        for i in range(self.n_layer)]
            # the K/V cache belongs to exactly these tensors (checked by identity on the next call)
            self._dec_state = None if dec is None else self._dec_pack(new_mems, dec.new_kv)
            res = res + (new_mems,)
        return res

    def backward(self, grad_scale: float = 1.0, layer_done_hook=None, flush_wgrads: bool = True, window_boundary: bool = True):
        """Accumulate d(loss * grad_scale)/d(params) of the last forward into the gradient arena.
        ``layer_done_hook(name)`` fires as soon as a layer's gradients are final (used by the data-parallel
        engine to start that layer's bucket all-reduce while earlier layers are still in backward).
        ``flush_wgrads`` (only with a ``wgrad_stash``): form the stashed weight gradients in this backward (the last micro-step of an
        accumulation window); False on the other micro-steps.
        ``window_boundary`` (only after forwards that joined a BackwardWindow, engine option defer_backward): False = nothing to do yet, the
        activations stay in the window; True = the backward of every micro-step the window holds, as one pass."""
        with torch.cuda.device(self.dev), ops.stream_scope():
            ctx = self._ctx
            if ctx is not None and getattr(ctx, "win", False):
                self._ctx = None
                if abs(grad_scale - ctx.fused_scale) > 1e-12 * max(1.0, abs(grad_scale)):
                    raise RuntimeError(f"backward(grad_scale={grad_scale}) after a fused head sweep taken at loss_grad_scale={ctx.fused_scale}")
                if window_boundary:
                    self._backward_window(grad_scale, layer_done_hook)
                return None
            return self._backward(grad_scale, layer_done_hook, flush_wgrads)

    def free_backward_window(self):
        """release the accumulation window's activation buffers (as large as a ga x micro-batch step keeps: 149 GiB at 16 x 4 x 1024 tokens of
        DB1-1.3B) -- e.g. before a memory-hungry evaluation between training phases; the next training forward under defer_backward rebuilds
        them.  Forwards recorded for a backward that has not run yet are dropped with them."""
        self._win = None
        self._ctx = None

    def _window_layer_ctx(self, i: int, n: int, B: int, L: int) -> _Ctx:
        win, H, D = self._win, self.n_head, self.d_head
        f = lambda k: win.full((k, i), n)
        c = _Ctx()
        c.x, c.qkv, c.av, c.s1, c.m1, c.r1 = f("x"), f("qkv"), f("av").view(B, L, H, D), f("s1"), f("m1"), f("r1")
        c.h1, c.z, c.act, c.s2, c.m2, c.r2 = f("h1"), f("z"), f("act"), f("s2"), f("m2"), f("r2")
        c.lse, c.qu, c.qv, c.probs, c.mblk, c.flash = f("lse"), f("qu"), f("qv"), f("probs"), f("mblk"), True
        if ("Rall",) in win.bufs:      # r_net of all layers in one launch per micro-step: [n micro-steps, layer (last first), nd, d]
            nl = self.n_layer
            ra = win.full(("Rall",), n)
            c.Rg = ra.view(n, nl, ra.shape[-2], ra.shape[-1])[:, nl - 1 - i]
        else:
            r = f("R")
            c.Rg = r.view(n, r.shape[0] // n, r.shape[1])
        c.R = c.Rg[0]
        return c

    def _backward_window(self, grad_scale, layer_done_hook):
        """the backward of the n micro-steps a BackwardWindow holds, as ONE pass over n x B sequences (same kernels as a single micro-step of
        that size; per-micro-step: the dropout steps -- by row block inside the LayerNorm backward --, the relative-position tables R of the
        dq_r stream and dR, and the embedding backward)"""
        win = self._win
        n = 0 if win is None else win.n
        if n == 0:
            return
        ctxs = win.ctxs[:n]
        c0 = ctxs[0]
        if any(abs(c.fused_scale - grad_scale) > 1e-12 * max(1.0, abs(grad_scale)) for c in ctxs):
            raise RuntimeError("deferred backward: the micro-steps of the window were taken at different loss scales")
        self._ctx = None
        self._gb = 0.0 if self._grad_fresh else 1.0
        self._grad_fresh = False
        d = self.d_model
        Bm, L = c0.B, c0.L
        T, B = Bm * L, n * Bm
        # dropout step of the window's first micro-step: the host counter's value then, or -- under a captured forward, whose step is the device
        # counter's value at replay -- the distance back from the counter's current value (the window's last forward bumped it last)
        step0 = None if c0.dstep is None else (-(n - 1) if self._drop_step_dev is not None else c0.dstep)
        dh = win.full(("dh",), n)
        R_in = win.full(("rin",), n)
        try:
            self._drop_rps = T
            for i in reversed(range(self.n_layer)):
                dh = self._layer_bwd(i, dh, self._window_layer_ctx(i, n, B, L), R_in, B, L, c0.shift, step0, flush=False)
                if layer_done_hook is not None:
                    layer_done_hook(f"h.{i}")
        finally:
            self._drop_rps = 0
        for m, c in enumerate(ctxs):
            dhm = dh[m * T:(m + 1) * T]
            if step0 is not None and self.embd_pdrop > 0:   # gradient through the embedding dropout: that micro-step's keep decisions
                ops.dropout(dhm, dhm, self._drop_args(self.embd_pdrop, self.SITE_EMBED, step0 + m))
            self._embed_bwd(dhm.view(Bm, L, d), c.ecs, c.shapes)
        if layer_done_hook is not None:
            layer_done_hook("embeddings")
        win.ctxs = [None] * win.ga
        win.n = 0

    def _backward(self, grad_scale, layer_done_hook, flush_wgrads=True):
        ctx = self._ctx
        if ctx is None:
            raise RuntimeError("backward() without a preceding forward(compute_loss=True)")
        self._ctx = None
        self._gb = 0.0 if self._grad_fresh else 1.0   # beta of the weight-gradient GEMMs: write on a fresh arena, accumulate otherwise
        self._grad_fresh = False
        if self.wgrad_stash is not None and self.wgrad_stash.slot == 0:
            self.wgrad_stash.beta = self._gb          # the flush of this accumulation window writes (fresh arena) or accumulates
        # (a stash rebuilt mid-window, first > 0, got its beta when it was built: 1 after the partial flush of the old one)
        d, V = self.d_model, self.total_vocab_size
        B, L = ctx.B, ctx.L
        T = B * L
        if ctx.dh_head is not None:   # the head's backward already ran inside the forward's sweep (fuse_head_loss)
            if abs(grad_scale - ctx.fused_scale) > 1e-12 * max(1.0, abs(grad_scale)):
                raise RuntimeError(f"backward(grad_scale={grad_scale}) after a fused head sweep taken at loss_grad_scale={ctx.fused_scale}")
            dh = ctx.dh_head
        else:
            dlogits = self._new(T, self.vocab_pad) if self.keep_logits else ctx.logits_pad
            ops.masked_ce_bwd(ctx.logits_pad, ctx.lab, ctx.msk, ctx.lse, ctx.sums, dlogits, V, gscale=grad_scale)
            wname = "word_embedding.weight" if self.share_input_output_embedding else "lm_head.weight"
            Wout = self.arena.view(self.arena.work, wname, full=True).view(self.vocab_pad, d)
            gW = self.arena.view(self.arena.grad, wname, full=True).view(self.vocab_pad, d)
            ops.gemm(dlogits.t(), ctx.hfin, gW, beta=self._gb, useful_flops=2.0 * T * V * d)
            dh = self._new(T, d)
            ops.gemm(dlogits, Wout, dh, useful_flops=2.0 * T * V * d)
            del dlogits
        for i in reversed(range(self.n_layer)):
            if self.pre_lnorm:
                dh = self._layer_bwd_prelnorm(i, dh, ctx.lcs[i], ctx.R_in, B, L, ctx.shift, ctx.dstep)
            else:
                dh = self._layer_bwd(i, dh, ctx.lcs[i], ctx.R_in, B, L, ctx.shift, ctx.dstep, flush=flush_wgrads)
            ctx.lcs[i] = None
            if layer_done_hook is not None:
                layer_done_hook(f"h.{i}")
        if ctx.dstep is not None and self.embd_pdrop > 0:   # gradient through the embedding dropout: the same keep decisions
            ops.dropout(dh, dh, self._drop_args(self.embd_pdrop, self.SITE_EMBED, ctx.dstep))
        self._embed_bwd(dh.view(B, L, d), ctx.ecs, ctx.shapes)
        if layer_done_hook is not None:
            layer_done_hook("embeddings")

    def zero_grad(self, set_to_none: bool = False):
        self.arena.grad.zero_()
        self._grad_fresh = True
        self._ctx = None
        if self._win is not None:       # (forwards recorded for a deferred backward are dropped with the gradients)
            self._win.ctxs, self._win.n = [None] * self._win.ga, 0

    def gemm_first_grads(self) -> List[str]:
        """parameters whose gradient's FIRST writer in a backward is a GEMM (which can write with beta = 0): the five weight matrices of
        every decoder layer and the output-embedding matrix (with tied embeddings the token scatter-add comes after the head's GEMM)"""
        names = [f"h.{i}.{n}" for i in range(self.n_layer) for n in ("pos_ff.CoreNet.2.weight", "pos_ff.CoreNet.0.weight", "dec_attn.o_net.weight",
                                                                    "dec_attn.r_net.weight", "dec_attn.qkv_net.weight")]
        return names + ["word_embedding.weight" if self.share_input_output_embedding else "lm_head.weight"]

    def accumulator_segments(self) -> torch.Tensor:
        """int64 [n, 2] device table of (offset, length) runs of the gradient arena that are NOT covered by ``gemm_first_grads``: what has
        to be cleared after an optimizer step (db1_zero_segments)"""
        skip = set(self.gemm_first_grads())
        runs: List[List[int]] = []
        for name, (off, shape, alloc) in self.arena.offsets.items():
            if name in skip:
                continue
            n = _round_up(alloc, 8)
            if runs and runs[-1][0] + runs[-1][1] == off:
                runs[-1][1] += n
            else:
                runs.append([off, n])
        return torch.tensor(runs, dtype=torch.int64, device=self.dev)

    # layer -> contiguous [start, end) element range of the gradient arena (bucket boundaries for data parallelism)
    def grad_buckets(self) -> List[Tuple[str, int, int]]:
        groups: Dict[str, List[int]] = {}
        order: List[str] = []
        for name, (off, shape, alloc) in self.arena.offsets.items():
            key = ".".join(name.split(".")[:2]) if name.startswith("h.") else "embeddings"
            if key not in groups:
                groups[key] = [off, off]
                order.append(key)
            groups[key][1] = off + _round_up(alloc, 8)
        return [(k, groups[k][0], groups[k][1]) for k in order]
