#!/usr/bin/env python3
"""Experiment: do HBM-bound kernels hide behind another sequence's GEMMs?  Two independent DB1-1.3B engines with HALF the micro-batch each,
driven from two host threads on two HIP streams of one GPU, against one engine with the whole micro-batch.  (If the aggregate rate is
clearly higher, the step should process its micro-batch as two interleaved halves.)
    python tools/exp/exp_two_streams.py [batch=64] [steps=6]"""
import os
import sys
import threading
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))   # tools/ (bench_kernels) and the repository root
import torch  # noqa: E402

from bdm_db1_amd import TransformerXL, initialize, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
cfg = synth.db1_config("1.3B", drop=0.1, embd_pdrop=0.1)
L = cfg.n_position


def make(bsz, seed):
    torch.manual_seed(seed)
    m = TransformerXL(cfg, device=dev)
    e, _, _, _ = initialize(SimpleNamespace(lr=1e-4, weight_decay=0.01, clip_grad=1.0, optimizer="adam", keep_logits=False), m)
    e.train()
    return e, [synth.text_batch(bsz, L, seed, dev)]


def run(engine, batch, stream, steps, barrier=None):
    with torch.cuda.stream(stream):
        if barrier is not None:
            barrier.wait()
        for _ in range(steps):
            _, loss = engine(batch)
            engine.backward(loss)
            engine.step()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


# ---- one engine, whole micro-batch
e1, b1 = make(B, 1)
run(e1, b1, torch.cuda.current_stream(), 2)
dt = timed(lambda: run(e1, b1, torch.cuda.current_stream(), STEPS))
print(f"one engine   B={B}: {B * L * STEPS / dt:9.0f} tok/s  {dt / STEPS * 1e3:7.1f} ms/step", flush=True)
del e1, b1
torch.cuda.empty_cache()

# ---- two engines with half the micro-batch, two streams, two host threads
pairs = [make(B // 2, 10 + i) for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for (e, b), s in zip(pairs, streams):
    run(e, b, s, 2)
torch.cuda.synchronize()
for stagger in (0, 1):
    def both():
        bar = threading.Barrier(2)
        ths = [threading.Thread(target=run, args=(e, b, s, STEPS, bar)) for (e, b), s in zip(pairs, streams)]
        if stagger:   # start the second sequence half a layer late so that its GEMMs meet the first one's element-wise kernels
            ths[0].start(); time.sleep(0.004); bar2 = None
        for t in ths[stagger:]:
            t.start()
        for t in ths:
            t.join()
    if stagger:
        # (a barrier of 2 cannot be used with a delayed start)
        def both():  # noqa: F811
            ths = [threading.Thread(target=run, args=(e, b, s, STEPS, None)) for (e, b), s in zip(pairs, streams)]
            ths[0].start(); time.sleep(0.004); ths[1].start()
            for t in ths:
                t.join()
    dt = timed(both)
    print(f"two engines  B={B // 2}+{B // 2} (stagger {stagger}): {B * L * STEPS / dt:9.0f} tok/s  {dt / STEPS * 1e3:7.1f} ms per pair of steps", flush=True)
# ---- for reference: one engine at half the micro-batch alone
dt = timed(lambda: run(pairs[0][0], pairs[0][1], streams[0], STEPS))
print(f"one engine   B={B // 2}: {B // 2 * L * STEPS / dt:9.0f} tok/s  {dt / STEPS * 1e3:7.1f} ms/step", flush=True)
