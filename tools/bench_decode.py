#!/usr/bin/env python3
"""Inference with Transformer-XL memory at the DB1-1.3B geometry (the evaluate_rl loop: evaluate_rl.py:157-266):
batch 1, memory of mem_len = 1024 positions, one call with a transition's observation tokens, then 1-token calls.
    python tools/bench_decode.py [q_first=22] [steps=20]
Prints ms per call for q = q_first and q = 1 (HIP-event timed, memory full)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bdm_db1_amd import TransformerXL, synth, lib  # noqa: E402
lib.apply_env_knobs()
from bdm_db1_amd.data import NLPTaskInput  # noqa: E402

q_first = int(sys.argv[1]) if len(sys.argv) > 1 else 22
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = TransformerXL(synth.db1_config("1.3B"), device=dev)
model.eval()


def call(ids, mems):
    x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
    with torch.no_grad():
        logits, _, mems = model([x], compute_loss=False, mems=mems)
    return logits, mems


def timed(q, n, mems):
    ids = torch.randint(0, 32000, (1, q), device=dev)
    for _ in range(3):
        _, mems = call(ids, mems)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        logits, mems = call(ids, mems)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    return e0.elapsed_time(e1) / n, wall, mems


for q in (q_first, 1, -q_first, -1):  # same calls as ONE hipGraph replay each (bdm_db1_amd/decode.py); negative: the K / V ring form
    from bdm_db1_amd import GraphedMemoryStep, GraphedRingStep
    ring = q < 0
    q = abs(q)
    step = (GraphedRingStep if ring else GraphedMemoryStep)(model, batch_size=1, n_new=q)
    ids = torch.randint(0, 32000, (1, q), device=dev)
    for _ in range(3):
        step(ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(ids)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"graphed{' ring' if ring else '     '} q={q:3d} mem={model.mem_len}: {ms:8.3f} ms/call (wall)  {q / ms * 1e3:9.1f} tokens/s")
    del step

mems = model.init_mem(1)
for q in (q_first, 1):
    ms, wall, mems = timed(q, steps, mems)
    print(f"decode q={q:3d} mem={mems[0].shape[1]}: {ms:8.3f} ms/call (GPU events)  {wall:8.3f} ms/call (wall)  {q / ms * 1e3:9.1f} tokens/s")
