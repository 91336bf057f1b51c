"""CPU restatement of the DB1 text path in torch eager ops (the reference IS torch eager: this is how its own CPU path computes, the
reference itself cannot travel to the GPU box): TEST INFRASTRUCTURE, NOT PRODUCT.  Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; ``bdm_db1_amd`` never does.

Forward as the reference executes it -- full (not causal-skipped) score matrices, ``_rel_shift`` by pad + view, ``masked_fill(-1e30)``,
softmax, P.v; post-LN residual blocks; GEGLU feed-forward; tied head; masked cross-entropy normalised by the mask sum -- backward by
autograd.  Covers the released configuration's flags only (post-LN, same_length, tied u / v, tied embeddings, geglu, text batches, no
memory, no dropout); pinned against the NumPy oracle (itself pinned to the reference's golden vectors) by tests/test_oracle_golden.py.
Reference lines (src/model/transformer_xl.py): PositionalEmbedding :34-50, _rel_shift :98-110, attention :122-243, PositionwiseFF :246-292,
mask :551-567, forward :506-619; GEGLU src/model/activations.py:19-32.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


class TorchCpuModel:
    def __init__(self, cfg, params: Dict[str, np.ndarray], dtype=torch.float32):
        assert not cfg.pre_lnorm and cfg.same_length and not cfg.untie_r and cfg.share_input_output_embedding and cfg.activation_fn == "geglu"
        self.cfg = cfg
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in params.items() if k != "pos_emb.inv_freq"}
        d = cfg.n_embed
        self.inv_freq = 1 / (10000 ** (torch.arange(0.0, d, 2.0) / d))      # :40
        self.dtype = dtype

    def _rel_shift(self, x):                                                # :98-110, x [qlen, klen, B, H]
        zero_pad = torch.zeros((x.size(0), 1, *x.size()[2:]), dtype=x.dtype)
        x_padded = torch.cat([zero_pad, x], dim=1)
        x_padded = x_padded.view(x.size(1) + 1, x.size(0), *x.size()[2:])
        return x_padded[1:].view_as(x)

    def _layer(self, i, w, r, mask):
        cfg, P = self.cfg, self.p
        H, D = cfg.n_head, cfg.n_embed // cfg.n_head
        pre = f"h.{i}."
        qlen, bsz = w.size(0), w.size(1)
        w_heads = F.linear(w, P[pre + "dec_attn.qkv_net.weight"])           # :136
        r_head_k = F.linear(r, P[pre + "dec_attn.r_net.weight"])            # :138
        q, k, v = torch.chunk(w_heads, 3, dim=-1)
        klen = k.size(0)
        q, k, v = q.view(qlen, bsz, H, D), k.view(klen, bsz, H, D), v.view(klen, bsz, H, D)
        r_head_k = r_head_k.view(klen, H, D)
        AC = torch.einsum("ibnd,jbnd->ijbn", q + P["r_w_bias"], k)          # :161-164
        BD = self._rel_shift(torch.einsum("ibnd,jnd->ijbn", q + P["r_r_bias"], r_head_k))   # :166-170
        score = (AC + BD) * (1.0 / math.sqrt(D))                            # :173
        score = score.float().masked_fill(mask[:, :, None, None], -1e30)    # :176-207
        prob = F.softmax(score, dim=1).to(w.dtype)                          # :209
        attn_vec = torch.einsum("ijbn,jbnd->ibnd", prob, v).contiguous().view(qlen, bsz, H * D)   # :220-225
        attn_out = F.linear(attn_vec, P[pre + "dec_attn.o_net.weight"])     # :228
        h1 = F.layer_norm(w + attn_out, (cfg.n_embed,), P[pre + "dec_attn.layer_norm.weight"], P[pre + "dec_attn.layer_norm.bias"],
                          cfg.layer_norm_epsilon)                           # :238
        z = F.linear(h1, P[pre + "pos_ff.CoreNet.0.weight"], P[pre + "pos_ff.CoreNet.0.bias"])    # :264
        a, b = z.chunk(2, dim=-1)
        core = F.linear(a * F.gelu(b), P[pre + "pos_ff.CoreNet.2.weight"], P[pre + "pos_ff.CoreNet.2.bias"])   # activations.py:19-32, :268
        return F.layer_norm(h1 + core, (cfg.n_embed,), P[pre + "pos_ff.layer_norm.weight"], P[pre + "pos_ff.layer_norm.bias"],
                            cfg.layer_norm_epsilon)                         # :290

    def forward(self, text_seq, label, loss_mask):
        """text_seq, label [B, L] int64, loss_mask [B, L] -> (logits [B, L, V], loss)"""
        cfg = self.cfg
        ids = torch.as_tensor(np.asarray(text_seq), dtype=torch.long)
        E = self.p["word_embedding.weight"]
        h = F.embedding(ids, E).transpose(0, 1).contiguous()                # [L, B, d] (the reference runs time-major inside the layers)
        qlen = h.size(0)
        mem_len = cfg.mem_len if cfg.mem_len is not None else 0
        ones = torch.ones(qlen, qlen, dtype=torch.uint8)
        mask_len = qlen - mem_len
        shift = qlen - mask_len if mask_len > 0 else qlen
        mask = (torch.triu(ones, 1) + torch.tril(ones, -shift)).bool()      # :551-563 (mlen = 0)
        pos_seq = torch.arange(qlen - 1, -1, -1.0).clamp(max=cfg.n_position)    # :569-573
        sinusoid = torch.ger(pos_seq, self.inv_freq)
        r = torch.cat([sinusoid.sin(), sinusoid.cos()], dim=-1).to(self.dtype)  # :43-45
        for i in range(cfg.n_layer):
            h = self._layer(i, h, r, mask)
        logits = F.linear(h.transpose(0, 1), E)                             # :593-598
        lab = torch.as_tensor(np.asarray(label), dtype=torch.long).reshape(-1)
        msk = torch.as_tensor(np.asarray(loss_mask), dtype=torch.float32).reshape(-1)
        losses = F.cross_entropy(logits.reshape(-1, logits.size(-1)).float(), lab, reduction="none")   # :602-609
        loss = (losses * msk).sum() / msk.sum()
        self._loss = loss
        return logits, loss

    def backward(self) -> Dict[str, np.ndarray]:
        for t in self.p.values():
            t.grad = None
        self._loss.backward()
        return {k: (t.grad.numpy() if t.grad is not None else np.zeros(tuple(t.shape), np.float32)) for k, t in self.p.items()}
