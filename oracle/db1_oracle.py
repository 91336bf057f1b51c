"""CPU restatement (NumPy) of the DB1 hot path: TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``bdm_db1_amd``) never does; it fails
loudly if the HIP library is missing instead of falling back to anything here.

Every function cites the reference lines (paths relative to the reference
checkout, ``src/...``) whose arithmetic it restates.  The restatement is pinned
against golden vectors produced by importing the reference itself
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``, checked by
``tests/test_oracle_golden.py``).  Parity status: PINNED for the model forward /
backward, the patch embedder, the scalar tokenizer and the LR/WD schedule;
the optimizer step is pinned against ``torch.optim.Adam/AdamW`` (the reference's
Adam lives in the un-vendored ``deepspeed==0.6.7`` and the reference has no test
for it, so that row is "parity unpinned w.r.t. DeepSpeed" by construction).

Parameters are passed as a dict keyed by the reference's ``state_dict`` names.
Internal arithmetic is float64 unless ``dtype`` says otherwise; the sinusoid
argument ``pos * inv_freq`` is formed in float32 exactly as the reference does
(its value is what makes the table reproducible to ~1e-7).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy.special import erf as _erf

Array = np.ndarray


# --------------------------------------------------------------------------------------
# configuration (attribute names = the ones TransformerXL.__init__ reads,
# src/model/transformer_xl.py:357-439, src/tokenizer/vision_embedding.py:95-115)
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    n_embed: int = 128
    n_position: int = 256
    n_layer: int = 2
    n_head: int = 4
    n_inner: Optional[int] = None
    pre_lnorm: bool = False
    mem_len: Optional[int] = 256
    same_length: bool = True
    untie_r: bool = False
    text_vocab_size: int = 32000
    num_discrete_values: int = 1024
    num_continuous_bin: int = 1024
    overlap_with_text: bool = True
    embd_pdrop: float = 0.0
    drop: float = 0.0
    dropattn: float = 0.0
    activation_fn: str = "geglu"
    layer_norm_epsilon: float = 1e-5
    share_input_output_embedding: bool = True
    use_deepnorm: bool = False
    fp16: bool = False
    vision_patch_size: int = 16
    vision_num_input_channels: int = 3
    vision_position_vocab_size: int = 128
    vision_hidden_dropout_prob: float = 0.0

    @property
    def d_inner(self) -> int:
        return 4 * self.n_embed if self.n_inner is None else self.n_inner

    @property
    def d_head(self) -> int:
        return self.n_embed // self.n_head

    @property
    def total_vocab_size(self) -> int:
        # transformer_xl.py:378-390
        v = self.text_vocab_size + self.num_continuous_bin
        if not self.overlap_with_text:
            v += self.num_discrete_values
        return v + 1

    @property
    def rl_separator_token_id(self) -> int:
        return self.total_vocab_size - 1

    @property
    def deepnorm_alpha(self) -> float:
        return (2 * self.n_layer) ** 0.25 if self.use_deepnorm else 1.0


# --------------------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------------------
def gelu(x: Array) -> Array:
    """erf GELU (torch.nn.functional.gelu default; activations.py:26-29)."""
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def gelu_grad(x: Array) -> Array:
    return 0.5 * (1.0 + _erf(x / math.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def layernorm_fwd(s: Array, gamma: Array, beta: Array, eps: float):
    """nn.LayerNorm over the last dim (transformer_xl.py:95,238,272,290)."""
    mu = s.mean(-1, keepdims=True)
    var = ((s - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (s - mu) * rstd
    return xhat * gamma + beta, (xhat, rstd)


def layernorm_bwd(dy: Array, gamma: Array, cache):
    xhat, rstd = cache
    g = dy * gamma
    dgamma = (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    dbeta = dy.reshape(-1, xhat.shape[-1]).sum(0)
    ds = rstd * (g - g.mean(-1, keepdims=True) - xhat * (g * xhat).mean(-1, keepdims=True))
    return ds, dgamma, dbeta


def inv_freq_f32(d_model: int) -> Array:
    """PositionalEmbedding.__init__ (transformer_xl.py:40), float32 op order."""
    ar = np.arange(0.0, d_model, 2.0, dtype=np.float32) / np.float32(d_model)
    return (np.float32(1.0) / np.power(np.float32(10000.0), ar, dtype=np.float32)).astype(np.float32)


def sinusoid_table(dist: Array, inv_freq: Array) -> Array:
    """PositionalEmbedding.forward (transformer_xl.py:43-45): cat(sin, cos) of the
    float32 outer product; ``dist`` are relative distances (already clamped)."""
    arg = np.outer(dist.astype(np.float32), inv_freq.astype(np.float32)).astype(np.float32)
    arg = arg.astype(np.float64)
    return np.concatenate([np.sin(arg), np.cos(arg)], axis=-1)


def visible_window(cfg: OracleConfig, qlen: int, mlen: int) -> Tuple[int, int]:
    """Attention-mask predicate of transformer_xl.py:551-567 in closed form.

    Key j (in [mem; w] coordinates) is visible from query i  iff
    ``i - shift < j <= i + mlen``.  Returns (shift, mlen)."""
    klen = qlen + mlen
    mem_len = cfg.mem_len if cfg.mem_len is not None else 0
    if cfg.same_length:
        mask_len = klen - mem_len
        shift = qlen - mask_len if mask_len > 0 else qlen
    else:
        shift = klen  # plain causal: nothing masked below the diagonal
    return shift, mlen


def attention_mask_dense(cfg: OracleConfig, qlen: int, mlen: int) -> Array:
    """1 = masked, same convention as the reference's uint8 mask."""
    shift, _ = visible_window(cfg, qlen, mlen)
    i = np.arange(qlen)[:, None]
    j = np.arange(qlen + mlen)[None, :]
    vis = (j <= i + mlen) & (j > i - shift)
    return (~vis).astype(np.uint8)


# --------------------------------------------------------------------------------------
# relative-position attention (RelPartialLearnableMultiHeadAttn, transformer_xl.py:112-243)
# --------------------------------------------------------------------------------------
def relattn_core_fwd(q, k, v, R, u, vb, masked, scale, mlen=0, pscale=None):
    """q:(B,Lq,H,D) k,v:(B,Lk,H,D) R:(n_dist,H,D) indexed by distance, u,vb:(H,D).

    score[i,j] = ((q_i+u).k_j + (q_i+vb).R[mlen+i-j]) * scale, masked_fill(-1e30),
    softmax over j, P.v.  This is the closed form of AC + rel_shift(BD)
    (transformer_xl.py:160-173,98-110), pinned by the golden model fixtures.
    Contractions are batched matmuls (BLAS) so the same code also serves as the timed CPU baseline.
    """
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    qu = (q + u).transpose(0, 2, 1, 3)             # (B,H,Lq,D)
    qv = (q + vb).transpose(0, 2, 1, 3)
    kt = k.transpose(0, 2, 3, 1)                   # (B,H,D,Lk)
    AC = qu @ kt
    T = qv @ R.transpose(1, 2, 0)[None]            # (B,H,Lq,n_dist)
    i = np.arange(Lq)[:, None]
    j = np.arange(Lk)[None, :]
    dist = np.clip(mlen + i - j, 0, R.shape[0] - 1)  # entries with dist<0 are always masked
    BD = np.take_along_axis(T, np.broadcast_to(dist[None, None], (B, H, Lq, Lk)), axis=3)
    S = (AC + BD) * scale
    S = np.where(masked[None, None].astype(bool), -1e30, S)
    S = S - S.max(-1, keepdims=True)
    P = np.exp(S)
    P = P / P.sum(-1, keepdims=True)
    Pd = P if pscale is None else P * pscale       # self.dropatt(attn_prob) (transformer_xl.py:211): pscale = keep / (1 - p), shape (B,H,Lq,Lk)
    out = (Pd @ v.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)
    return out, (P, dist, mlen, masked, pscale)


def relattn_core_bwd(dout, q, k, v, R, u, vb, scale, cache):
    P, dist, mlen, masked, pscale = cache
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    do = dout.transpose(0, 2, 1, 3)                # (B,H,Lq,D)
    dP = do @ v.transpose(0, 2, 3, 1)
    Pd = P if pscale is None else P * pscale
    if pscale is not None:
        dP = dP * pscale                            # gradient through the dropout on the probabilities
    dv = (Pd.transpose(0, 1, 3, 2) @ do).transpose(0, 2, 1, 3)
    dS = P * (dP - (P * dP).sum(-1, keepdims=True)) * scale
    # masked_fill passes no gradient to the scores it overwrites: only matters for a row whose keys are ALL hidden (its P is
    # uniform, not zero -- e.g. same_length with mem_len = 0); elsewhere P, hence dS, is already exactly zero there
    dS = np.where(masked[None, None].astype(bool), 0.0, dS)
    dqk = (dS @ k.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)      # gradient w.r.t. (q+u)
    dk = (dS.transpose(0, 1, 3, 2) @ (q + u).transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)
    # re-index dS by distance: dT[b,n,i,r] = dS[b,n,i,j] with j = mlen + i - r (each (i,r) has at most one j;
    # masked entries of dS are exactly zero, so the clipped distances of the forward contribute nothing)
    nd = R.shape[0]
    i = np.arange(Lq)[:, None]
    r = np.arange(nd)[None, :]
    jj = mlen + i - r
    ok = (jj >= 0) & (jj < Lk)
    dT = np.take_along_axis(dS, np.broadcast_to(np.clip(jj, 0, Lk - 1)[None, None], (B, H, Lq, nd)), axis=3) * ok[None, None]
    dqr = (dT @ R.transpose(1, 0, 2)[None]).transpose(0, 2, 1, 3)   # gradient w.r.t. (q+vb)
    qv = (q + vb).transpose(0, 2, 1, 3)
    # dR[r,n,:] = sum_{b,i} dT[b,n,i,r] qv[b,n,i,:]: one BLAS product per head over the (batch, query) rows (np.einsum ran this
    # contraction in its scalar C loop: half of the whole backward's time)
    dTn = dT.transpose(1, 0, 2, 3).reshape(H, B * Lq, nd)
    qvn = qv.transpose(1, 0, 2, 3).reshape(H, B * Lq, D)
    dR = (dTn.transpose(0, 2, 1) @ qvn).transpose(1, 0, 2)
    dq = dqk + dqr
    du = dqk.sum((0, 1))
    dvb = dqr.sum((0, 1))
    return dq, dk, dv, dR, du, dvb


# --------------------------------------------------------------------------------------
# image-patch embedder (src/tokenizer/vision_embedding.py:36-86) and position ids (:117-180)
# --------------------------------------------------------------------------------------
def _im2col3x3(x: Array) -> Array:
    """x:(N,C,P,P) -> (N,P,P,C*9) with zero padding 1 (per patch: the conv never sees
    neighbouring patches, vision_embedding.py:67-79)."""
    N, C, P, _ = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    cols = np.empty((N, P, P, C, 3, 3), dtype=x.dtype)
    for ky in range(3):
        for kx in range(3):
            cols[:, :, :, :, ky, kx] = xp[:, :, ky:ky + P, kx:kx + P].transpose(0, 2, 3, 1)
    return cols.reshape(N, P, P, C * 9)


def _col2im3x3(dcols: Array, C: int) -> Array:
    N, P, _, _ = dcols.shape
    d = dcols.reshape(N, P, P, C, 3, 3)
    dxp = np.zeros((N, C, P + 2, P + 2), dtype=dcols.dtype)
    for ky in range(3):
        for kx in range(3):
            dxp[:, :, ky:ky + P, kx:kx + P] += d[:, :, :, :, ky, kx].transpose(0, 3, 1, 2)
    return dxp[:, :, 1:-1, 1:-1]


def conv3x3_fwd(x, w, b):
    cols = _im2col3x3(x)
    y = cols @ w.reshape(w.shape[0], -1).T + b  # (N,P,P,Cout)
    return y.transpose(0, 3, 1, 2), cols


def conv3x3_bwd(dy, w, cols, need_dx=True):
    N, Co, P, _ = dy.shape
    dyl = dy.transpose(0, 2, 3, 1)  # (N,P,P,Co)
    dw = (dyl.reshape(-1, Co).T @ cols.reshape(-1, cols.shape[-1])).reshape(w.shape)
    db = dyl.reshape(-1, Co).sum(0)
    dx = None
    if need_dx:
        dcols = dyl @ w.reshape(Co, -1)
        dx = _col2im3x3(dcols, w.shape[1])
    return dx, dw, db


def groupnorm_fwd(x, gamma, beta, groups=32, eps=1e-5):
    N, C, H, W = x.shape
    xg = x.reshape(N, groups, -1)
    mu = xg.mean(-1, keepdims=True)
    var = ((xg - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = ((xg - mu) * rstd).reshape(N, C, H, W)
    return xhat * gamma[None, :, None, None] + beta[None, :, None, None], (xhat, rstd, groups)


def groupnorm_bwd(dy, gamma, cache):
    xhat, rstd, groups = cache
    N, C, H, W = dy.shape
    dgamma = (dy * xhat).sum((0, 2, 3))
    dbeta = dy.sum((0, 2, 3))
    g = (dy * gamma[None, :, None, None]).reshape(N, groups, -1)
    xh = xhat.reshape(N, groups, -1)
    dx = rstd * (g - g.mean(-1, keepdims=True) - xh * (g * xh).mean(-1, keepdims=True))
    return dx.reshape(N, C, H, W), dgamma, dbeta


def patchify(pixels: Array, p: int) -> Array:
    """einops 'b c (h p1) (w p2) -> (b h w) c p1 p2' (vision_embedding.py:67-72)."""
    B, C, Hh, Ww = pixels.shape
    h, w = Hh // p, Ww // p
    x = pixels.reshape(B, C, h, p, w, p).transpose(0, 2, 4, 1, 3, 5)
    return x.reshape(B * h * w, C, p, p)


def patch_normalize(x: Array, p: int) -> Array:
    """(x-mean)/(1e-6+std_unbiased) per (patch, channel), then /sqrt(p) (vision_embedding.py:73-77)."""
    mean = x.mean((-2, -1), keepdims=True)
    std = x.std((-2, -1), keepdims=True, ddof=1)
    return (x - mean) / (1e-6 + std) / math.sqrt(p)


def patch_embed_fwd(params: Dict[str, Array], pixels: Array, p: int, prefix="vision_encoder.patch_embeddings."):
    """PatchEmbeddings.forward (vision_embedding.py:65-86). pixels:(B,C,H,W) -> (B, n_patch, d)."""
    P = lambda n: params[prefix + n]
    B = pixels.shape[0]
    x0 = patch_normalize(patchify(pixels, p), p)
    c1, cols1 = conv3x3_fwd(x0, P("conv1.weight"), P("conv1.bias"))
    g0, gc0 = groupnorm_fwd(c1, P("residual_path.0.weight"), P("residual_path.0.bias"))
    a0 = gelu(g0)
    c2, cols2 = conv3x3_fwd(a0, P("residual_path.2.weight"), P("residual_path.2.bias"))
    g1, gc1 = groupnorm_fwd(c2, P("residual_path.3.weight"), P("residual_path.3.bias"))
    a1 = gelu(g1)
    c3, cols3 = conv3x3_fwd(a1, P("residual_path.5.weight"), P("residual_path.5.bias"))
    y = c1 + c3
    wp = P("projection.weight")
    flat = y.reshape(y.shape[0], -1)
    out = flat @ wp.reshape(wp.shape[0], -1).T + P("projection.bias")
    cache = dict(cols1=cols1, gc0=gc0, g0=g0, cols2=cols2, gc1=gc1, g1=g1, cols3=cols3, flat=flat, yshape=y.shape)
    return out.reshape(B, -1, wp.shape[0]), cache


def patch_embed_bwd(params, dout: Array, cache, prefix="vision_encoder.patch_embeddings."):
    P = lambda n: params[prefix + n]
    grads: Dict[str, Array] = {}
    wp = P("projection.weight")
    do = dout.reshape(-1, wp.shape[0])
    grads[prefix + "projection.weight"] = (do.T @ cache["flat"]).reshape(wp.shape)
    grads[prefix + "projection.bias"] = do.sum(0)
    dy = (do @ wp.reshape(wp.shape[0], -1)).reshape(cache["yshape"])
    da1, dw, db = conv3x3_bwd(dy, P("residual_path.5.weight"), cache["cols3"])
    grads[prefix + "residual_path.5.weight"], grads[prefix + "residual_path.5.bias"] = dw, db
    dg1 = da1 * gelu_grad(cache["g1"])
    dc2, dgam, dbet = groupnorm_bwd(dg1, P("residual_path.3.weight"), cache["gc1"])
    grads[prefix + "residual_path.3.weight"], grads[prefix + "residual_path.3.bias"] = dgam, dbet
    da0, dw, db = conv3x3_bwd(dc2, P("residual_path.2.weight"), cache["cols2"])
    grads[prefix + "residual_path.2.weight"], grads[prefix + "residual_path.2.bias"] = dw, db
    dg0 = da0 * gelu_grad(cache["g0"])
    dc1, dgam, dbet = groupnorm_bwd(dg0, P("residual_path.0.weight"), cache["gc0"])
    grads[prefix + "residual_path.0.weight"], grads[prefix + "residual_path.0.bias"] = dgam, dbet
    dc1 = dc1 + dy
    _, dw, db = conv3x3_bwd(dc1, P("conv1.weight"), cache["cols1"], need_dx=False)
    grads[prefix + "conv1.weight"], grads[prefix + "conv1.bias"] = dw, db
    return grads


def vision_position_ids_eval(h0: int, w0: int, vocab: int) -> Tuple[Array, Array]:
    """Eval-mode row/col indices (vision_embedding.py:134-148,170-172): float32 ops,
    truncation to int32, midpoint of [low, high)."""
    seq = np.arange(h0 * w0)
    row, col = seq // w0, seq % w0
    f = np.float32

    def lo_hi(idx, n):
        hi = ((idx + 1).astype(f) / f(n) * f(vocab)).astype(np.int32)
        lo = (idx.astype(f) / f(n) * f(vocab)).astype(np.int32)
        return lo, hi

    rl, rh = lo_hi(row, h0)
    cl, ch = lo_hi(col, w0)
    r = ((rl + rh).astype(f) / f(2)).astype(np.int32)
    c = ((cl + ch).astype(f) / f(2)).astype(np.int32)
    return r, c


def vision_embed_fwd(params, cfg: OracleConfig, pixels: Array, row_ids=None, col_ids=None):
    """VisionEmbedding.forward (vision_embedding.py:117-180); ids default to eval mode."""
    p = cfg.vision_patch_size
    emb, cache = patch_embed_fwd(params, pixels, p)
    h0, w0 = pixels.shape[2] // p, pixels.shape[3] // p
    if row_ids is None:
        row_ids, col_ids = vision_position_ids_eval(h0, w0, cfg.vision_position_vocab_size)
        row_ids, col_ids = row_ids[None], col_ids[None]
    out = emb + params["vision_encoder.row_position_embeddings.weight"][row_ids] \
        + params["vision_encoder.col_position_embeddings.weight"][col_ids]
    cache["row_ids"], cache["col_ids"] = np.broadcast_to(row_ids, out.shape[:2]), np.broadcast_to(col_ids, out.shape[:2])
    return out, cache


def vision_embed_bwd(params, dout: Array, cache):
    grads = patch_embed_bwd(params, dout, cache)
    for nm, ids in (("row", cache["row_ids"]), ("col", cache["col_ids"])):
        key = f"vision_encoder.{nm}_position_embeddings.weight"
        g = np.zeros_like(params[key], dtype=dout.dtype)
        np.add.at(g, ids.reshape(-1), dout.reshape(-1, dout.shape[-1]))
        grads[key] = g
    return grads


# --------------------------------------------------------------------------------------
# dropout (nn.Dropout, transformer_xl.py:229,262-269,409,545,575) with the counter-based keep decisions of the HIP kernels
# (bdm_db1_amd/csrc/db1_common.h: Db1Drop / db1_drop_apply).  torch's own RNG stream cannot be reproduced on another device, so
# parity at p > 0 is defined on the mask function: both sides draw keep(e) from Philox4x32-10 on (e / 8, site, step) keyed by the
# seed; everything downstream of the mask is then compared as usual.
# --------------------------------------------------------------------------------------
SITE_EMBED, SITE_POS = 0xE0000000, 0xE0000001


def site_of(layer: int, which: int) -> int:
    """which: 0 = attention output (:229), 1 = feed-forward output (:269), 2 = attention probabilities (:211, element order [H, B, Lq, Lk])"""
    return layer * 4 + which


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32 with 10 rounds (Salmon et al. 2011; the Random123 / cuRAND generator) on uint32 arrays -> four uint32 arrays"""
    M = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(x, np.uint64) & M for x in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0) & M, np.uint64(k1) & M
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M, p0 & M
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return c0, c1, c2, c3


def dropout_scale(n: int, p: float, seed: int, site: int, step: int) -> Array:
    """float64 [n] (n a multiple of 8): 0 where element e is dropped, 65536 / (65536 - thr) where it is kept, thr = round(p * 65536);
    keep(e) = [u16 >= thr], the eight u16 of elements 8b .. 8b+7 being the output words of philox(b lo, b hi, site, step; seed) split
    low half first"""
    assert n % 8 == 0 and 0.0 <= p < 1.0
    thr = int(min(65535, max(0, np.rint(np.float32(p) * np.float32(65536.0))))) if p > 0 else 0
    if thr == 0:
        return np.ones(n)
    blk = np.arange(n // 8, dtype=np.uint64)
    o = philox4x32_10(blk & np.uint64(0xFFFFFFFF), blk >> np.uint64(32), np.uint64(site), np.uint64(step), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.empty((n // 8, 8), np.uint64)
    for w in range(4):
        u[:, 2 * w] = o[w] & np.uint64(0xFFFF)
        u[:, 2 * w + 1] = o[w] >> np.uint64(16)
    return np.where(u.reshape(-1) >= thr, 65536.0 / (65536 - thr), 0.0)


# --------------------------------------------------------------------------------------
# scalar tokenizer (src/tokenizer/scalar_tokenizer.py:28-63)
# --------------------------------------------------------------------------------------
def mulaw_discretize(x: Array, is_action: bool, num_bins: int = 1024, mu: float = 100.0, M: float = 256.0) -> Array:
    """float32 op order of ContinuousScalarTokenizer.discretize; ``log`` is the correctly
    rounded float32 logarithm (float64 log rounded once)."""
    f = np.float32
    x = np.asarray(x, dtype=f)
    if not is_action:
        t = (np.abs(x) * f(mu) + f(1.0)).astype(f)
        lg = np.log(t.astype(np.float64)).astype(f)
        den = f(np.log(np.float64(f(mu * M + 1.0))))
        y = ((np.sign(x) * lg).astype(f) / den).astype(f)
        x = np.clip(y, f(-1), f(1))
    z = (((x + f(1)).astype(f) / f(2)).astype(f) * f(num_bins)).astype(f)
    ids = np.trunc(z).astype(np.int32)
    return np.clip(ids, 0, num_bins - 1).astype(np.int32)


def mulaw_decode(ids: Array, is_action: bool, num_bins: int = 1024, mu: float = 100.0, M: float = 256.0) -> Array:
    """ContinuousScalarTokenizer.decode (scalar_tokenizer.py:47-63)."""
    f = np.float32
    x = np.clip(np.asarray(ids), 0, num_bins - 1).astype(f)
    x = ((x / f(num_bins)).astype(f) * f(2) - f(1)).astype(f)
    if not is_action:
        x = (np.sign(x) * (np.power(f(1 + M * mu), np.abs(x)).astype(f) - f(1)) / f(mu)).astype(f)
    return x


# --------------------------------------------------------------------------------------
# RL packing helpers (src/data/rl_dataset.py:44-71)
# --------------------------------------------------------------------------------------
def rl_action_flag_and_position_id(index_l, index_r, obs_seq_len, act_seq_len, prepend_trans_num):
    """_get_action_flag_and_position_id (rl_dataset.py:44-71).  One transition is
    [obs(obs_seq_len), SEP, act(act_seq_len)]; position_id = 1..obs_seq_len+1 on the
    observation tokens *and* the separator, 0 on action tokens; action flag = 1 on action
    tokens of non-prompt transitions.  The window [index_l, index_r] starts on a transition."""
    n = index_r - index_l + 1
    step = obs_seq_len + act_seq_len + 1
    t = np.arange(n)
    within = t % step
    pos = np.where(within <= obs_seq_len, within + 1, 0).astype(np.int64)
    flag = ((within > obs_seq_len) & (t >= prepend_trans_num * step)).astype(np.int64)
    return flag, pos


def truncate_or_pad(arr: Array, seq_len: int) -> Array:
    """_truncate_or_pad_to_match_seq_len (rl_dataset.py:865-872)."""
    arr = np.asarray(arr)
    if len(arr) >= seq_len:
        return arr[:seq_len]
    return np.pad(arr, (0, seq_len - len(arr)))


# --------------------------------------------------------------------------------------
# optimizer + schedule
# --------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01, adamw=True, grad_scale=1.0):
    """One Adam / AdamW step with torch.optim semantics (bias-corrected, eps outside sqrt
    of the corrected second moment).  ``grad_scale`` folds clipping / loss-scale in."""
    g = g * grad_scale
    if adamw:
        p = p * (1.0 - lr * wd)
    else:
        g = g + wd * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def clip_coef(total_norm: float, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ coefficient (clamped at 1)."""
    return min(1.0, max_norm / (total_norm + 1e-6))


def lr_at(step, max_lr, min_lr, warmup, decay_steps, style):
    """OptimizerParamScheduler.get_lr (optimizer_param_scheduler.py:101-134)."""
    if warmup > 0 and step <= warmup:
        return max_lr * float(step) / float(warmup)
    if style == "constant":
        return max_lr
    if step > decay_steps:
        return min_lr
    ratio = float(step - warmup) / float(decay_steps - warmup)
    if style == "linear":
        coeff = 1.0 - ratio
    elif style == "cosine":
        coeff = 0.5 * (math.cos(math.pi * ratio) + 1.0)
    else:
        raise Exception(f"{style} decay style is not supported.")
    return min_lr + coeff * (max_lr - min_lr)


def wd_at(step, start_wd, end_wd, incr_steps, style):
    """OptimizerParamScheduler.get_wd (optimizer_param_scheduler.py:73-99)."""
    if step > incr_steps:
        return end_wd
    if style == "constant":
        return end_wd
    ratio = float(step) / float(incr_steps)
    if style == "linear":
        coeff = ratio
    elif style == "cosine":
        coeff = 0.5 * (math.cos(math.pi * (1 - ratio)) + 1.0)
    else:
        raise Exception(f"{style} weight decay increment style is not supported.")
    return start_wd + coeff * (end_wd - start_wd)


# --------------------------------------------------------------------------------------
# the model (TransformerXL.forward, transformer_xl.py:506-748) with a hand-written backward
# --------------------------------------------------------------------------------------
@dataclass
class TaskBatch:
    """Duck-typed stand-in for the reference's GatoInputBase family (src/data/input_specs.py).
    kind in {"nlp","rl","ic","vqa"}; fields are numpy arrays with the reference's names."""
    kind: str
    label: Optional[Array] = None
    loss_mask: Optional[Array] = None
    position_id: Optional[Array] = None
    text_seq: Optional[Array] = None
    tensor_seq: Optional[Array] = None
    vision_seq: Optional[Array] = None
    prompt_seq: Optional[Array] = None
    img_seq: Optional[Array] = None
    ques_len: Optional[Array] = None        # VQA only; the reference reads it for a debug offset, it does not enter the maths (:721)
    vision_row_ids: Optional[Array] = None  # injected (train-mode) position picks; None = eval rule
    vision_col_ids: Optional[Array] = None


class OracleModel:
    def __init__(self, cfg: OracleConfig, params: Dict[str, Array], dtype=np.float64):
        self.cfg = cfg
        self.dtype = dtype
        self.p = {k: np.asarray(v, dtype=dtype) for k, v in params.items() if k != "pos_emb.inv_freq"}
        self.inv_freq = np.asarray(params["pos_emb.inv_freq"], np.float32) if "pos_emb.inv_freq" in params \
            else inv_freq_f32(cfg.n_embed)
        self._dropout = None

    # ---- helpers
    def _bias(self, name, i):
        return self.p[f"h.{i}.dec_attn.{name}"] if self.cfg.untie_r else self.p[name]

    def _embed_tasks(self, tasks: Sequence[TaskBatch]):
        cfg, P = self.cfg, self.p
        E = P["word_embedding.weight"]
        embs, labels, masks, caches = [], [], [], []
        for t in tasks:
            c = {"kind": t.kind}
            if t.kind == "nlp":  # _forward_nlp :662-672
                emb = E[t.text_seq]
                c["ids"] = t.text_seq
            elif t.kind == "rl":  # _forward_rl :621-660
                ids = t.tensor_seq
                B, L = ids.shape
                emb = np.zeros((B, L, cfg.n_embed), self.dtype)
                valid = ids >= 0
                emb[valid] = E[ids[valid]]
                c["ids"], c["valid"] = ids, valid
                if t.vision_seq is not None:
                    img = t.vision_seq.reshape(-1, *t.vision_seq.shape[-3:]).astype(self.dtype)
                    vis, vc = vision_embed_fwd(P, cfg, img, t.vision_row_ids, t.vision_col_ids)
                    vis = vis.reshape(B, -1, cfg.n_embed)
                    nph = int((ids == -1).sum(1)[0])
                    emb[ids == -1] = vis[:, :nph].reshape(-1, cfg.n_embed)
                    c["vis_cache"], c["vis_shape"], c["nph"] = vc, vis.shape, nph
                emb = emb + P["rl_local_timestep_embedding.weight"][t.position_id]
                c["position_id"] = t.position_id
            elif t.kind in ("ic", "vqa"):  # _forward_ic :674-703, _forward_vqa :705-748
                pe = E[t.prompt_seq]
                vis, vc = vision_embed_fwd(P, cfg, t.img_seq.astype(self.dtype), t.vision_row_ids, t.vision_col_ids)
                te = E[t.text_seq]
                emb = np.concatenate([pe, vis, te], axis=1)
                c.update(prompt=t.prompt_seq, text=t.text_seq, vis_cache=vc, nvis=vis.shape[1])
            else:
                raise ValueError(t.kind)
            embs.append(emb)
            caches.append(c)
            if t.label is not None:
                lab = np.asarray(t.label).copy()
                lab[lab == -1] = 0  # :644-645 (done for RL only in the reference; -1 never occurs elsewhere)
                labels.append(lab)
                masks.append(np.asarray(t.loss_mask, dtype=self.dtype))
        return embs, labels, masks, caches

    def _drop(self, x, p, site):
        """nn.Dropout in training mode with the shared counter-based mask function; returns (dropped x, scale array or None)"""
        if self._dropout is None or p <= 0:
            return x, None
        m = dropout_scale(x.size, p, self._dropout["seed"], site, self._dropout["step"]).reshape(x.shape).astype(self.dtype)
        return x * m, m

    def _layer_fwd(self, i, x, R_in, masked, mlen, mem=None):
        """RelPartialLearnableDecoderLayer.forward (:326-353) -> dec_attn (:112-243) + pos_ff (:276-292)."""
        cfg, P = self.cfg, self.p
        H, D = cfg.n_head, cfg.d_head
        a = cfg.deepnorm_alpha
        pre = f"h.{i}."
        c = {}
        B, L, d = x.shape
        ln1 = (P[pre + "dec_attn.layer_norm.weight"], P[pre + "dec_attn.layer_norm.bias"])
        cat = x if mem is None else np.concatenate([mem, x], axis=1)
        if cfg.pre_lnorm:
            hin, c["ln1"] = layernorm_fwd(cat, *ln1, cfg.layer_norm_epsilon)
        else:
            hin = cat
        qkv = hin @ P[pre + "dec_attn.qkv_net.weight"].T
        q, k, v = np.split(qkv, 3, axis=-1)
        q = q[:, -L:]
        Lk = k.shape[1]
        q, k, v = q.reshape(B, L, H, D), k.reshape(B, Lk, H, D), v.reshape(B, Lk, H, D)
        R = (R_in @ P[pre + "dec_attn.r_net.weight"].T).reshape(-1, H, D)
        u, vb = self._bias("r_w_bias", i), self._bias("r_r_bias", i)
        scale = 1.0 / math.sqrt(D)
        pscale = None
        if self._dropout is not None and cfg.dropattn > 0:   # mask drawn over the device's [H, B, Lq, Lk] element order
            pscale = dropout_scale(B * H * L * Lk, cfg.dropattn, self._dropout["seed"], site_of(i, 2), self._dropout["step"]) \
                .reshape(H, B, L, Lk).transpose(1, 0, 2, 3).astype(self.dtype)
        av, ac = relattn_core_fwd(q, k, v, R, u, vb, masked, scale, mlen, pscale)
        av2 = av.reshape(B, L, H * D)
        o = av2 @ P[pre + "dec_attn.o_net.weight"].T
        o, c["drop_o"] = self._drop(o, cfg.drop, site_of(i, 0))                 # :229
        if cfg.pre_lnorm:
            h1 = x + o
        else:
            h1, c["ln1"] = layernorm_fwd(x * a + o, *ln1, cfg.layer_norm_epsilon)
        c.update(x=x, hin=hin, q=q, k=k, v=v, R=R, u=u, vb=vb, scale=scale, ac=ac, av2=av2)
        # feed-forward
        ln2 = (P[pre + "pos_ff.layer_norm.weight"], P[pre + "pos_ff.layer_norm.bias"])
        if cfg.pre_lnorm:
            fin, c["ln2"] = layernorm_fwd(h1, *ln2, cfg.layer_norm_epsilon)
        else:
            fin = h1
        z = fin @ P[pre + "pos_ff.CoreNet.0.weight"].T + P[pre + "pos_ff.CoreNet.0.bias"]
        if cfg.activation_fn == "geglu":
            za, zb = np.split(z, 2, axis=-1)
            act = za * gelu(zb)
        elif cfg.activation_fn == "gelu":
            act = gelu(z)
        elif cfg.activation_fn == "relu":
            act = np.maximum(z, 0)
        else:
            raise NotImplementedError(cfg.activation_fn)
        f = act @ P[pre + "pos_ff.CoreNet.2.weight"].T + P[pre + "pos_ff.CoreNet.2.bias"]
        f, c["drop_f"] = self._drop(f, cfg.drop, site_of(i, 1))                 # CoreNet's trailing nn.Dropout, :262-269
        if cfg.pre_lnorm:
            out = f + h1
        else:
            out, c["ln2"] = layernorm_fwd(h1 * a + f, *ln2, cfg.layer_norm_epsilon)
        c.update(h1=h1, fin=fin, z=z, act=act)
        return out, c

    def _layer_bwd(self, i, dout, c, R_in, grads):
        cfg, P = self.cfg, self.p
        H, D = cfg.n_head, cfg.d_head
        a = cfg.deepnorm_alpha
        pre = f"h.{i}."
        B, L, d = dout.shape
        G = lambda n, g: grads.__setitem__(n, grads.get(n, 0) + g)
        # ---- FF
        if cfg.pre_lnorm:
            df, dh1 = dout, dout.copy()
        else:
            ds, dg, db = layernorm_bwd(dout, P[pre + "pos_ff.layer_norm.weight"], c["ln2"])
            G(pre + "pos_ff.layer_norm.weight", dg); G(pre + "pos_ff.layer_norm.bias", db)
            df, dh1 = ds, ds * a
        if c.get("drop_f") is not None:
            df = df * c["drop_f"]
        W2 = P[pre + "pos_ff.CoreNet.2.weight"]
        G(pre + "pos_ff.CoreNet.2.weight", df.reshape(-1, d).T @ c["act"].reshape(-1, W2.shape[1]))
        G(pre + "pos_ff.CoreNet.2.bias", df.reshape(-1, d).sum(0))
        dact = df @ W2
        z = c["z"]
        if cfg.activation_fn == "geglu":
            za, zb = np.split(z, 2, axis=-1)
            dz = np.concatenate([dact * gelu(zb), dact * za * gelu_grad(zb)], axis=-1)
        elif cfg.activation_fn == "gelu":
            dz = dact * gelu_grad(z)
        else:
            dz = dact * (z > 0)
        W1 = P[pre + "pos_ff.CoreNet.0.weight"]
        G(pre + "pos_ff.CoreNet.0.weight", dz.reshape(-1, W1.shape[0]).T @ c["fin"].reshape(-1, d))
        G(pre + "pos_ff.CoreNet.0.bias", dz.reshape(-1, W1.shape[0]).sum(0))
        dfin = dz @ W1
        if cfg.pre_lnorm:
            ds, dg, db = layernorm_bwd(dfin, P[pre + "pos_ff.layer_norm.weight"], c["ln2"])
            G(pre + "pos_ff.layer_norm.weight", dg); G(pre + "pos_ff.layer_norm.bias", db)
            dh1 = dh1 + ds
        else:
            dh1 = dh1 + dfin
        # ---- attention
        if cfg.pre_lnorm:
            do, dx = dh1, dh1.copy()
        else:
            ds, dg, db = layernorm_bwd(dh1, P[pre + "dec_attn.layer_norm.weight"], c["ln1"])
            G(pre + "dec_attn.layer_norm.weight", dg); G(pre + "dec_attn.layer_norm.bias", db)
            do, dx = ds, ds * a
        if c.get("drop_o") is not None:
            do = do * c["drop_o"]
        Wo = P[pre + "dec_attn.o_net.weight"]
        G(pre + "dec_attn.o_net.weight", do.reshape(-1, d).T @ c["av2"].reshape(-1, d))
        dav = (do @ Wo).reshape(B, L, H, D)
        dq, dk, dv, dR, du, dvb = relattn_core_bwd(dav, c["q"], c["k"], c["v"], c["R"], c["u"], c["vb"], c["scale"], c["ac"])
        G(pre + "dec_attn.r_w_bias" if cfg.untie_r else "r_w_bias", du)
        G(pre + "dec_attn.r_r_bias" if cfg.untie_r else "r_r_bias", dvb)
        G(pre + "dec_attn.r_net.weight", dR.reshape(-1, d).T @ R_in)
        dqkv = np.concatenate([dq.reshape(B, L, d), dk.reshape(B, L, d), dv.reshape(B, L, d)], axis=-1)
        G(pre + "dec_attn.qkv_net.weight", dqkv.reshape(-1, 3 * d).T @ c["hin"].reshape(-1, d))
        dhin = dqkv @ P[pre + "dec_attn.qkv_net.weight"]
        if cfg.pre_lnorm:
            ds, dg, db = layernorm_bwd(dhin, P[pre + "dec_attn.layer_norm.weight"], c["ln1"])
            G(pre + "dec_attn.layer_norm.weight", dg); G(pre + "dec_attn.layer_norm.bias", db)
            dx = dx + ds
        else:
            dx = dx + dhin
        return dx

    # ---- public
    def forward(self, tasks: Sequence[TaskBatch], compute_loss=True, mems: Optional[List[Array]] = None,
                keep_cache=True, dropout: Optional[dict] = None):
        """``dropout = {"seed": int, "step": int}`` = training mode with cfg.drop / cfg.embd_pdrop under the shared mask function
        (None = eval mode / p = 0, the reference's behaviour in ``model.eval()``)"""
        cfg, P = self.cfg, self.p
        assert not (compute_loss and mems is not None)
        self._dropout = dropout
        embs, labels, masks, ecaches = self._embed_tasks(tasks)
        h = np.concatenate(embs, axis=0)
        h, drop_e = self._drop(h, cfg.embd_pdrop, SITE_EMBED)                    # :545
        B, L, d = h.shape
        mlen = mems[0].shape[1] if mems is not None else 0
        klen = L + mlen
        masked = attention_mask_dense(cfg, L, mlen)
        if masked.sum() == 0:
            raise ValueError("empty attention mask (transformer_xl.py:177,205-206)")
        # distance table: pos_seq = [klen-1..0] clamped (:569-575) re-indexed by distance
        dist = np.minimum(np.arange(klen, dtype=np.float32), np.float32(cfg.n_position))
        R_in = sinusoid_table(dist, self.inv_freq).astype(self.dtype)
        R_in, _ = self._drop(R_in, cfg.embd_pdrop, SITE_POS)                     # :575 (a constant input: no gradient to carry)
        hids, lcaches = [], []
        for i in range(cfg.n_layer):
            hids.append(h)
            h, c = self._layer_fwd(i, h, R_in, masked, mlen, None if mems is None else mems[i])
            lcaches.append(c if keep_cache else None)
        Wout = P["word_embedding.weight"] if cfg.share_input_output_embedding else P["lm_head.weight"]
        logits = h @ Wout.T
        loss = None
        cache = dict(ecaches=ecaches, lcaches=lcaches, R_in=R_in, hfin=h, embs_shapes=[e.shape for e in embs], drop_e=drop_e)
        if compute_loss:
            lab = np.concatenate(labels, axis=0).astype(np.int64).reshape(-1)
            msk = np.concatenate(masks, axis=0).reshape(-1)
            lg = logits.reshape(-1, logits.shape[-1])
            mx = lg.max(-1, keepdims=True)
            lse = mx[:, 0] + np.log(np.exp(lg - mx).sum(-1))
            nll = lse - lg[np.arange(lg.shape[0]), lab]
            loss = (nll * msk).sum() / msk.sum()  # :602-609
            cache.update(lab=lab, msk=msk, lse=lse)
        new_mems = None
        if mems is not None:  # _update_mem :487-504
            mem_len = cfg.mem_len or 0
            end = mlen + max(0, L)
            beg = max(0, end - mem_len)
            new_mems = [np.concatenate([mems[i], hids[i]], axis=1)[:, beg:end] for i in range(cfg.n_layer)]
        self._cache = cache
        self._logits = logits
        return logits, loss, new_mems

    def backward(self) -> Dict[str, Array]:
        """Gradient of the loss returned by the last forward() w.r.t. every parameter."""
        cfg, P, c = self.cfg, self.p, self._cache
        grads: Dict[str, Array] = {}
        lg = self._logits.reshape(-1, self._logits.shape[-1])
        prob = np.exp(lg - c["lse"][:, None])
        prob[np.arange(lg.shape[0]), c["lab"]] -= 1.0
        dlg = prob * (c["msk"] / c["msk"].sum())[:, None]
        hfin = c["hfin"]
        B, L, d = hfin.shape
        wname = "word_embedding.weight" if cfg.share_input_output_embedding else "lm_head.weight"
        grads[wname] = dlg.T @ hfin.reshape(-1, d)
        dh = (dlg @ P[wname]).reshape(B, L, d)
        for i in reversed(range(cfg.n_layer)):
            dh = self._layer_bwd(i, dh, c["lcaches"][i], c["R_in"], grads)
        if c.get("drop_e") is not None:
            dh = dh * c["drop_e"]
        # embeddings
        E = "word_embedding.weight"
        gE = grads.get(E, np.zeros_like(P[E]))
        b0 = 0
        for ec, shp in zip(c["ecaches"], c["embs_shapes"]):
            de = dh[b0:b0 + shp[0]]
            b0 += shp[0]
            if ec["kind"] == "nlp":
                np.add.at(gE, ec["ids"].reshape(-1), de.reshape(-1, d))
            elif ec["kind"] == "rl":
                ids, valid = ec["ids"], ec["valid"]
                np.add.at(gE, ids[valid], de[valid])
                key = "rl_local_timestep_embedding.weight"
                g = grads.get(key, np.zeros_like(P[key]))
                np.add.at(g, ec["position_id"].reshape(-1), de.reshape(-1, d))
                grads[key] = g
                if "vis_cache" in ec:
                    dvis = np.zeros(ec["vis_shape"], self.dtype)
                    dvis[:, :ec["nph"]] = de[ids == -1].reshape(shp[0], ec["nph"], d)
                    dvis = dvis.reshape(ec["vis_cache"]["row_ids"].shape + (d,))
                    for k2, v2 in vision_embed_bwd(P, dvis, ec["vis_cache"]).items():
                        grads[k2] = grads.get(k2, 0) + v2
            else:
                npmt, nvis = ec["prompt"].shape[1], ec["nvis"]
                np.add.at(gE, ec["prompt"].reshape(-1), de[:, :npmt].reshape(-1, d))
                np.add.at(gE, ec["text"].reshape(-1), de[:, npmt + nvis:].reshape(-1, d))
                for k2, v2 in vision_embed_bwd(P, de[:, npmt:npmt + nvis], ec["vis_cache"]).items():
                    grads[k2] = grads.get(k2, 0) + v2
        grads[E] = gE
        return grads


def count_params(cfg: OracleConfig) -> int:
    d, H, D, V = cfg.n_embed, cfg.n_head, cfg.d_head, cfg.total_vocab_size
    per_layer = 3 * d * d + d * d + d * d + 2 * d + cfg.d_inner * d + cfg.d_inner \
        + (cfg.d_inner // 2 if cfg.activation_fn == "geglu" else cfg.d_inner) * d + d + 2 * d
    if cfg.untie_r:
        per_layer += 2 * H * D
    vis = 64 * 3 * 9 + 64 + 2 * (64 * 64 * 9 + 64) + 4 * 64 + d * 64 * cfg.vision_patch_size ** 2 + d \
        + 2 * cfg.vision_position_vocab_size * d
    n = cfg.n_layer * per_layer + V * d + vis + 513 * d + (0 if cfg.untie_r else 2 * H * D)
    if not cfg.share_input_output_embedding:
        n += V * d
    return n
