#!/usr/bin/env python3
"""One-token calls for M environments at once over a K / V ring (DB1-1.3B geometry, full memory): ms per call graphed and eager.
    python tools/bench_decode_batched.py M [calls=30]        (under rocprofv3 --kernel-trace --stats: the per-kernel picture of the eager calls)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bdm_db1_amd import GraphedRingStep, RingMemory, TransformerXL, lib, synth  # noqa: E402
lib.apply_env_knobs()
from bdm_db1_amd.data import NLPTaskInput  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = TransformerXL(synth.db1_config("1.3B"), device=dev)
model.eval()
ids = torch.randint(0, 32000, (M, 1), device=dev)
if os.environ.get("DB1_DECODE_GRAPH", "1") != "0":
    step = GraphedRingStep(model, batch_size=M, n_new=1)
    step.ids.copy_(ids)
    for _ in range(5):
        step(step.ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        step(step.ids)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / calls * 1e3
    print(f"graphed ring M={M}: {ms:.4f} ms/call  {M / ms * 1e3:.1f} tokens/s")
    del step
mem = RingMemory(model, M)
x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
with torch.no_grad():
    for _ in range(3):
        model([x], compute_loss=False, mems=mem)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        model([x], compute_loss=False, mems=mem)
    e1.record()
    torch.cuda.synchronize()
print(f"eager ring M={M}: {e0.elapsed_time(e1) / calls:.4f} ms/call (GPU events)")
