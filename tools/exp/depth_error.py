"""Experiment: how the bf16 path's error against the fp32 HIP path grows with depth (DB1-1.3B geometry, B x 1024 tokens), and which kernel family
carries it.  python tools/exp/depth_error.py [B] [variants...]   variants: default noflash generic noheadbias"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
from bdm_db1_amd import TransformerXL, synth, lib
from bdm_db1_amd.data import NLPTaskInput
import test_full_depth_gpu as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
variants = sys.argv[2:] or ["default"]
depths = [1, 2, 4, 8, 16, 24]


def run(cfg, params, batch, dtype, variant):
    text, label, mask = batch
    model = TransformerXL(cfg, compute_dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items() if k in dict(model.named_parameters()) or True}, strict=False)
    model.eval()
    if variant == "noflash":
        model.use_flash = False
    if variant == "noheadbias":
        model.use_headbias_epilogue = False
    if variant == "generic" and dtype == torch.bfloat16:
        lib.load().db1_test_gemm_force_generic(1)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
    x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=T(mask), label=T(label), text_seq=T(text), text_len=None)
    with torch.enable_grad():
        logits, loss = model([x])
    lg = logits.float().cpu().numpy()
    model.backward()
    g = {n: model.G(n).float().cpu().numpy().copy() for n in ("h.0.dec_attn.qkv_net.weight", "word_embedding.weight")}
    lib.load().db1_test_gemm_force_generic(0)
    del model
    torch.cuda.empty_cache()
    return lg, float(loss), g


l2 = F._l2rel
full = synth.db1_config("1.3B", n_layer=24)
params = F._params(full)
for nl in depths:
    cfg = synth.db1_config("1.3B", n_layer=nl)
    batch = F._batch(cfg, B)
    lg32, loss32, g32 = run(cfg, params, batch, torch.float32, "default")
    for v in variants:
        lg16, loss16, g16 = run(cfg, params, batch, torch.bfloat16, v)
        print(json.dumps({"layers": nl, "variant": v, "logits_l2": round(l2(lg16, lg32), 5), "logits_max": round(F._maxrel(lg16, lg32), 5),
                          "mean_signed_rel": round(float(np.mean((lg16 - lg32) * np.sign(lg32)) / np.mean(np.abs(lg32))), 5),
                          "loss16": loss16, "loss32": loss32, "g_qkv0_l2": round(l2(g16["h.0.dec_attn.qkv_net.weight"], g32["h.0.dec_attn.qkv_net.weight"]), 5),
                          "g_emb_l2": round(l2(g16["word_embedding.weight"], g32["word_embedding.weight"]), 5)}), flush=True)
