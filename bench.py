#!/usr/bin/env python3
"""DB1-1.3B pre-training throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: spawns its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = forward + backward + gradient all-reduce (N > 1) + global-norm clip + fused Adam on one
micro-batch of B synthetic 1024-token sequences per GPU (weak scaling: per-GPU work is fixed).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GEGLU_NOTE = ("since round 4 the two feed-forward GEMMs of a layer carry the GEGLU activation and its backward in their epilogues (db1_gemm_nt_geglu / "
              "db1_gemm_nn_geglu_bwd): ~20 ms per step of formerly separate HBM-bound passes now sit INSIDE this family's time while its FLOPs are unchanged, "
              "so the fraction reads ~0.02 lower than with DB1_GEGLU_EPI=0 although the step is 2 % faster (same-box A/B: profiles/r04a_bench_{default,geglu_unfused}.json)")
FAMILY_NOTE = ("bf16 MFMA tile GEMM family gemm_bf16_{w4,w4n,pp,pp32,tile256,tile}_kernel + splitk_reduce_kernel: every tile-GEMM launch of the timed steps "
               "(decoder layers incl. the batched dR contraction and split-K reduces, and the three head products of the chunked head + loss sweep with its "
               "ce_fwd_bwd / ce_sum kernels), rank 0; FLOPs = executed work: the vocabulary pad of the head and k-tiles skipped as structural zeros are not counted")
FLOP_PER_TOKEN = 6_899_036_160       # fwd+bwd, DB1-1.3B, L=1024, causal-counted attention (SURVEY.md 8d / BASELINE.md 4)
FLOP_PER_PATCH = 317_227_008         # image-patch embedder fwd+bwd
MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak, MI355X_MICROARCH.md


HBM_PEAK_GBPS = 8000.0               # HBM3E peak, MI355X_MICROARCH.md
# the reference's OWN torch-CPU path, timed by the survey in its container (SURVEY.md section 6 / BASELINE.md section 2): 0.76 s per
# 1.3B-geometry layer, forward + backward, one 1024-token sequence, 8 host cores, fp32 -> ~56 tokens/s.  Quoted beside the port's
# number for context only: the reference cannot travel to the GPU box, so it cannot be re-timed there.
REFERENCE_CPU_TOKENS_PER_S_8_CORES = 56.0


def _oracle_params(O, rng, d, H, k, vocab):
    f = np.float32
    params = {"r_w_bias": (rng.standard_normal((H, d // H)) * 0.02).astype(f), "r_r_bias": (rng.standard_normal((H, d // H)) * 0.02).astype(f),
              "word_embedding.weight": (rng.standard_normal((vocab, d)) * 0.02).astype(f)}
    for i in range(k):
        p = f"h.{i}."
        params[p + "dec_attn.qkv_net.weight"] = (rng.standard_normal((3 * d, d)) * 0.02).astype(f)
        params[p + "dec_attn.o_net.weight"] = (rng.standard_normal((d, d)) * 0.02).astype(f)
        params[p + "dec_attn.r_net.weight"] = (rng.standard_normal((d, d)) * 0.02).astype(f)
        params[p + "pos_ff.CoreNet.0.weight"] = (rng.standard_normal((4 * d, d)) * 0.02).astype(f)
        params[p + "pos_ff.CoreNet.0.bias"] = np.zeros(4 * d, f)
        params[p + "pos_ff.CoreNet.2.weight"] = (rng.standard_normal((d, 2 * d)) * 0.02).astype(f)
        params[p + "pos_ff.CoreNet.2.bias"] = np.zeros(d, f)
        for ln in ("dec_attn.layer_norm", "pos_ff.layer_norm"):
            params[p + ln + ".weight"], params[p + ln + ".bias"] = np.ones(d, f), np.zeros(d, f)
    return params


def _time_oracle(O, cfg, params, ids, repeats):
    model = O.OracleModel(cfg, params, dtype=np.float32)
    task = O.TaskBatch(kind="nlp", text_seq=ids[:, :-1], label=ids[:, 1:], loss_mask=np.ones((ids.shape[0], ids.shape[1] - 1), np.float32))
    ts = []
    for r in range(repeats + 1):   # the first run pays first-touch page faults and BLAS thread start-up: not timed
        t0 = time.perf_counter()
        model.forward([task])
        model.backward()
        if r:
            ts.append(time.perf_counter() - t0)
    return ts


def cpu_baseline(repeats: int = 5):
    """The CPU oracle (oracle/db1_oracle.py, NumPy + OpenBLAS, fp32) timed on this box's host cores, SURVEY 8d's recipe (median of 5 after
    one untimed run): (1) BASELINE config 1 -- DB1-tiny (2 layers, d = 128, 4 heads) on 8 x 256 tokens, forward + loss + backward, whole;
    (2) a bounded sample of the benchmarked workload -- DB1-1.3B geometry, ONE 1024-token sequence (config 2 at B = 1), forward + backward
    through k = 1 and k = 2 of the 24 decoder layers plus the tied head and loss; tokens/s is extrapolated to 24 layers from the per-layer
    difference.  Reported baseline only (rank 0, N = 1)."""
    from oracle import db1_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    fmt = lambda v: "/".join(f"{x:.2f}" for x in v)
    rng = np.random.default_rng(0)
    # ---- config 1: DB1-tiny, 8 x 256 tokens (seed 0), the whole model
    cfg_t = O.OracleConfig(n_embed=128, n_layer=2, n_head=4, n_position=256, mem_len=256)
    ts_t = _time_oracle(O, cfg_t, _oracle_params(O, rng, 128, 4, 2, cfg_t.total_vocab_size), rng.integers(0, 32000, (8, 257)), repeats)
    tiny = {"value": round(8 * 256 / float(np.median(ts_t)), 1), "unit": "tokens/s", "seconds": [round(x, 3) for x in ts_t],
            "sample": "BASELINE config 1: DB1-tiny (2 layers, d 128, 4 heads, vocabulary 33 025), 8 x 256 tokens, fwd + loss + bwd, fp32, median of "
                      f"{repeats} after one untimed run"}
    # ---- config 2 at B = 1: DB1-1.3B geometry
    d, H, L = 2048, 16, 1024
    med, runs = {}, {}
    for k in (1, 2):
        cfg = O.OracleConfig(n_embed=d, n_layer=k, n_head=H, n_position=L, mem_len=L)
        runs[k] = _time_oracle(O, cfg, _oracle_params(O, rng, d, H, k, cfg.total_vocab_size), rng.integers(0, 32000, (1, L + 1)), repeats)
        med[k] = float(np.median(runs[k]))
    per_layer = max(med[2] - med[1], 1e-9)
    fixed = max(med[1] - per_layer, 0.0)
    full = fixed + 24 * per_layer
    # ---- the same sample through the torch-eager restatement (oracle/db1_torch_cpu.py: the ops the reference itself runs on a CPU -- einsum
    # scores over the FULL L x L matrix, pad + view rel_shift, autograd backward), all host cores: the stronger of the two ports is the value
    torch_port = None
    try:
        from oracle.db1_torch_cpu import TorchCpuModel
        all_threads = torch.get_num_threads()
        best_t = None
        # eager torch on many cores loses to itself on few (128 threads: 17 tokens/s on the r04 box, the survey's 8-core run of the reference:
        # 56): the port is timed at 8 and at 32 threads (median of 3 after one untimed run) and the faster one is reported
        for tthreads in sorted({min(8, all_threads), min(32, all_threads)}):
            torch.set_num_threads(tthreads)
            tmed, truns = {}, {}
            for k in (1, 2):
                cfg = O.OracleConfig(n_embed=d, n_layer=k, n_head=H, n_position=L, mem_len=L)
                tm = TorchCpuModel(cfg, _oracle_params(O, rng, d, H, k, cfg.total_vocab_size))
                ids = rng.integers(0, 32000, (1, L + 1))
                ts = []
                for r in range(3 + 1):
                    t0 = time.perf_counter()
                    tm.forward(ids[:, :-1], ids[:, 1:], np.ones((1, L), np.float32))
                    tm.backward()
                    if r:
                        ts.append(time.perf_counter() - t0)
                truns[k], tmed[k] = ts, float(np.median(ts))
            tper = max(tmed[2] - tmed[1], 1e-9)
            tfull = max(tmed[1] - tper, 0.0) + 24 * tper
            cand = {"value": round(L / tfull, 2), "unit": "tokens/s", "cores": int(tthreads),
                    "sample": f"the same sample through oracle/db1_torch_cpu.py (torch eager ops + autograd, fp32, {tthreads} threads, median of 3): "
                              f"{fmt(truns[1])} s; {fmt(truns[2])} s, extrapolated to 24 layers ({tfull:.1f} s/sequence)"}
            if best_t is None or cand["value"] > best_t["value"]:
                other = None if best_t is None else {"cores": best_t["cores"], "value": best_t["value"]}
                best_t = dict(cand, other_thread_count=other) if other else cand
            else:
                best_t = dict(best_t, other_thread_count={"cores": cand["cores"], "value": cand["value"]})
        # ---- and the WHOLE 24-layer model through the faster thread count, measured ("time it once instead of extrapolating"):
        # one untimed run (first-touch page faults of ~10 GB of saved activations), then one timed forward + backward.  The 24 layers share
        # the arrays of layer 0 as their initial values (every layer still has its own torch parameter and gradient).
        try:
            torch.set_num_threads(best_t["cores"])
            cfg24 = O.OracleConfig(n_embed=d, n_layer=24, n_head=H, n_position=L, mem_len=L)
            p1 = _oracle_params(O, rng, d, H, 1, cfg24.total_vocab_size)
            p24 = dict(p1)
            for i in range(1, 24):
                for k_, v_ in p1.items():
                    if k_.startswith("h.0."):
                        p24[f"h.{i}." + k_[4:]] = v_
            tm = TorchCpuModel(cfg24, p24)
            ids = rng.integers(0, 32000, (1, L + 1))
            secs = []
            for r in range(2):
                t0 = time.perf_counter()
                tm.forward(ids[:, :-1], ids[:, 1:], np.ones((1, L), np.float32))
                tm.backward()
                secs.append(time.perf_counter() - t0)
            del tm, p24
            best_t = dict(best_t, extrapolated_value=best_t["value"], value=round(L / secs[1], 2), measured_24_layers=True,
                          seconds_24_layers=[round(x, 2) for x in secs],
                          sample=best_t["sample"] + f"; VALUE = all 24 layers + head measured once after one untimed run: {secs[1]:.1f} s per 1024-token sequence "
                                                    f"(untimed first run {secs[0]:.1f} s)")
        except Exception as e24:
            best_t = dict(best_t, measured_24_layers=False, error_24_layers=repr(e24))
        torch.set_num_threads(all_threads)
        torch_port = best_t
    except Exception as e:   # (never take the line down)
        torch_port = {"value": None, "error": repr(e)}
    numpy_port = {"value": round(L / full, 2), "unit": "tokens/s", "cores": int(threads)}
    if torch_port.get("value") and torch_port["value"] > numpy_port["value"]:
        best = {"value": torch_port["value"], "cores": torch_port["cores"], "port": "torch-eager restatement (oracle/db1_torch_cpu.py)"}
    else:
        best = {"value": numpy_port["value"], "cores": numpy_port["cores"], "port": "NumPy oracle (oracle/db1_oracle.py)"}
    return {"value": best["value"], "unit": "tokens/s", "cores": best["cores"], "kind": "port", "port": best["port"],
            "measured_24_layers": bool(torch_port.get("measured_24_layers")) and best["port"].startswith("torch"),
            "numpy_oracle": numpy_port, "torch_eager_port": torch_port,
            "sample": f"DB1-1.3B geometry, 1 sequence x 1024 tokens, fwd+bwd, fp32, two CPU ports of the path timed (value = the faster); NumPy/OpenBLAS oracle: 1 and 2 decoder layers + tied head, "
                      f"median of {repeats} timed runs each after one untimed run ({fmt(runs[1])} s; {fmt(runs[2])} s), extrapolated to 24 layers (the torch port's value is MEASURED on all 24 layers) "
                      f"({full:.1f} s/sequence); {os.cpu_count()} logical CPUs on the box, {threads} BLAS threads (its elementwise passes over [16, 1024, 1024] "
                      f"fp32 tensors are single-threaded NumPy); torch-eager port: see torch_eager_port.sample",
            "tiny_config1": tiny,
            "reference_torch_cpu": {"value": REFERENCE_CPU_TOKENS_PER_S_8_CORES, "unit": "tokens/s", "cores": 8,
                                    "note": "the reference's own torch-CPU forward+backward at the same geometry, timed by the survey in its "
                                            "container (SURVEY.md section 6); context only, not re-timed on this box"}}


def decode_leg(model, dev, calls: int = 30):
    """SURVEY 8f-1 (evaluate_rl.py:157-266): inference with a full Transformer-XL memory, batch 1, ONE new token per call -- the loop the
    released evaluation runs -- on the model the training steps just used: ms per call as one hipGraph replay and eager, and the call
    against its roofline: it has to stream every bf16 decoder / head weight and the cached K / V of the memory once (algorithmic bytes)
    at the HBM rate."""
    from bdm_db1_amd import GraphedMemoryStep, GraphedRingStep
    from bdm_db1_amd.data import NLPTaskInput
    was_training = model.training
    model.eval()
    d, H, V, nl, mem = model.d_model, model.n_head, model.total_vocab_size, model.n_layer, int(model.mem_len)
    di = model.d_inner if hasattr(model, "d_inner") else 4 * d
    w_layer = (3 * d * d + d * d + di * d + d * (di // 2 if str(getattr(model, "activation_fn", "geglu")) == "geglu" else di)) * 2
    bytes_call = nl * (w_layer + 2 * mem * d * 2) + V * d * 2        # weights + K / V of the memory per layer + the tied head
    try:
        ids = torch.randint(0, 32000, (1, 1), device=dev)

        def timed(step, in_place=False):
            tok = ids
            if in_place:      # the token is written into the step's static input buffer (what a device-side sampler does): no copy per call
                step.ids.copy_(ids)
                tok = step.ids
            for _ in range(5):
                step(tok)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(calls):
                step(tok)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / calls
        ms_graph = timed(GraphedRingStep(model, batch_size=1, n_new=1), in_place=True)   # K / V ring appended in place: the default graphed form
        ms_graph_list = timed(GraphedMemoryStep(model, batch_size=1, n_new=1))     # the list-memory contract (hidden states copied per call)
        mems = model.init_mem(1)
        x = NLPTaskInput(position_id=None, attention_mask=None, loss_mask=None, label=None, text_seq=ids, text_len=None)
        with torch.no_grad():
            for _ in range(3):
                _, _, mems = model([x], compute_loss=False, mems=mems)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                _, _, mems = model([x], compute_loss=False, mems=mems)
            torch.cuda.synchronize()
        ms_eager = (time.perf_counter() - t0) / calls * 1e3
        gbps = bytes_call / (ms_graph * 1e-3) / 1e9
        # M environments per call (evaluate_rl.py:452-482 gives a rank up to ~110 independent environments; get_action_batched): the weights
        # are streamed once per call whatever M is, every environment brings its own K / V ring (2 x mem x d bf16 per layer)
        batched = {}
        for M in (4, 16):
            try:
                idsM = torch.randint(0, 32000, (M, 1), device=dev)
                stepM = GraphedRingStep(model, batch_size=M, n_new=1)
                stepM.ids.copy_(idsM)
                for _ in range(5):
                    stepM(stepM.ids)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(calls):
                    stepM(stepM.ids)
                e1.record()
                torch.cuda.synchronize()
                msM = e0.elapsed_time(e1) / calls
                bytesM = nl * (w_layer + M * 2 * mem * d * 2) + V * d * 2
                batched[str(M)] = {"ms_per_call": round(msM, 4), "tokens_per_s": round(M * 1e3 / msM, 1), "speedup_vs_batch_1": round(M * ms_graph / msM, 2),
                                   "hbm_frac": round(bytesM / (msM * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_call": int(bytesM)}
                del stepM
            except Exception as eM:
                batched[str(M)] = {"ms_per_call": None, "error": repr(eM)}
        out = {"workload": f"inference with Transformer-XL memory, batch 1, mem_len {mem}, 1 new token per call (evaluate_rl.py:157-266), bf16, K / V of the memory cached",
               "ms_per_call": round(ms_graph, 4), "ms_per_call_list_memory_graph": round(ms_graph_list, 4), "ms_per_call_eager": round(ms_eager, 4), "calls": calls, "tokens_per_s": round(1e3 / ms_graph, 1),
               "batched_environments": dict(batched, note="M environments per call over one RingMemory(model, M): one weight stream per token for all of them "
                                                           "(bdm_db1_amd.evaluation.get_action_batched); tokens_per_s = M / ms_per_call"),
               "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                            "algorithmic_bytes_per_call": int(bytes_call),
                            "kernel": ("the whole call as one hipGraph replay over a K / V ring: per layer relattn_decode_ring (chunk partials) + db1_decode_chain "
                                       "(o_net with the merge, LN, ff1 + GEGLU, ff2, LN, next qkv as one persistent launch)" if getattr(model, "use_decode_chain", False) and nl >= 2 else
                                       "the whole call as one hipGraph replay over a K / V ring (skinny GEMMs over the bf16 weights + relattn_decode_ring)")}}
    except Exception as e:   # the decode leg must never take the bench line down
        out = {"ms_per_call": None, "error": repr(e)}
    model.train(was_training)
    return out


# ---- box calibration: the boxes of the pool differ by +-2-4 % in what they sustain under the 1.4 kW cap -- as much as
# a round's gain -- so every bench line says what ITS box does on two code-independent probes, and carries a tokens/s normalised to the
# pool's median box.  POOL_MFMA_RANDOM_TF: median of the in-register MFMA rates this project has recorded on the pool's boxes
# (profiles/r06_box_calibration.txt lists them).  The step does not scale linearly with that rate: over the round-6 records (four boxes,
# 2145 ... 2220 TFLOP/s against 153.8 ... 156.3 k tokens/s) d ln(tokens/s) / d ln(mfma_random_tf) ~ 0.4 -- the probe runs the matrix pipe alone
# at the cap, the step shares the cap with its data path and a quarter of it is HBM-bound -- so the normalisation uses that exponent.  It is an
# attribution aid, not a measurement: `value` is what was measured.
POOL_MFMA_RANDOM_TF = 2205.0
BOX_RATE_EXPONENT = 0.4


def _smi_sample_start():
    """rocm-smi clock / power read, started now and collected later (it takes ~0.5 s: it runs beside the warm-up steps)"""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    try:
        return subprocess.Popen([exe, "--showclocks", "--showpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    except Exception:
        return None


def _smi_sample_collect(proc, dev_index: int):
    if proc is None:
        return None, None
    try:
        txt, _ = proc.communicate(timeout=20)
        rec = json.loads(txt[txt.index("{"):])
        card = rec.get(f"card{dev_index}") or next(iter(rec.values()))
        sclk = power = None
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl and "clock" in kl and sclk is None:
                digits = "".join(ch for ch in str(v) if ch.isdigit() or ch == ".")
                sclk = float(digits) if digits else None
            if "power" in kl and "(w)" in kl and power is None:
                try:
                    power = float(v)
                except Exception:
                    pass
        return sclk, power
    except Exception:
        return None, None


def box_calibration(dev, smi_proc=None):
    """{mfma_random_tf, copy_gbps[, sclk_mhz, power_w]}: ~50 ms of in-register bf16 MFMAs on random-mantissa operands (csrc/calib.hip; best of
    two after a short ramp), a 4 GB device copy (2 GB read + 2 GB written, best of three), and rocm-smi's clock / power if a sample was taken"""
    from bdm_db1_amd import lib, ops
    out = {}
    try:
        sink = torch.empty(65536, device=dev, dtype=torch.float32)
        iters = 400000

        def run(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.call("db1_test_mfma_calibration", ops.P(sink), n, 1, ops.stream())
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)
        run(iters // 8)
        ms = min(run(iters) for _ in range(2))
        out["mfma_random_tf"] = round(256 * 4 * iters * 16 * 16384.0 / (ms * 1e-3) / 1e12, 1)
        src = torch.empty(1 << 29, device=dev, dtype=torch.float32)
        dst = torch.empty_like(src)
        src.zero_()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dst.copy_(src)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out["copy_gbps"] = round(2 * src.numel() * 4 / (best * 1e-3) / 1e9, 1)
        del src, dst, sink
    except Exception as e:
        out["error"] = repr(e)
    if smi_proc is not None:
        sclk, power = _smi_sample_collect(smi_proc, dev.index or 0)
        out["sclk_mhz"], out["power_w"] = sclk, power
    return out


def _count_patches(batch):
    n = 0
    for t in batch:
        if getattr(t, "img_seq", None) is not None:
            n += t.img_seq.shape[0] * (t.img_seq.shape[2] // 16) * (t.img_seq.shape[3] // 16)
        if getattr(t, "vision_seq", None) is not None:
            v = t.vision_seq
            n += v.shape[0] * v.shape[1] * (v.shape[3] // 16) * (v.shape[4] // 16)
    return int(n)


LEG_WORKLOADS = {
    "rl": "BASELINE config 4: RL-trajectory sequences (Atari-like 3x64x80 observation frames -> 20 image patches + separator + discrete action per "
          "transition, 47 transitions per sequence; rl_dataset.py:590-752)",
    "mixture": "BASELINE config 5 stand-in: 50 % RL-trajectory rows with 3x64x80 observation frames, 25 % text rows, 25 % caption rows with one "
               "3x224x224 image each (blendable_dataset.py:45-72; the 870-task weights are unpublished)",
    "caption": "BASELINE config 3: caption sequences (8 prompt ids + one 3x224x224 image = 196 patches + 820 text ids; coco_token_dataset.py:104-152)",
}


def workload_leg(engine, model, dev, cfg, B, L, seed, workload: str, world: int = 1, steps: int = 5, warmup: int = 2):
    """Another of BASELINE.json's workloads on the SAME model, engine and ranks, right after the text steps: forward + backward (+ the
    bucketed gradient all-reduce with more than one rank) + clip + Adam, ``steps`` timed steps after ``warmup`` untimed ones, bracketed by
    barrier + device synchronisation and taken as the MAX over ranks like the main leg; tokens/s is the whole job's.  Every rank calls this
    (the collectives of the step and the all_gather of the timings need all of them); only rank 0's return value is printed."""
    from bdm_db1_amd import synth
    # with more than one rank a leg that fails on ONE rank must not leave the others inside a collective until the RCCL timeout.
    # The part that can fail on its own -- building the batch (host memory, a bad shape) -- is agreed on first (MIN over an ok flag: every
    # rank skips the leg together); a failure inside the steps, whose backward is full of collectives, is re-raised: the launcher tears the
    # job down at once instead of hanging it.
    err = None
    try:
        batch = [synth.rl_batch(B, L, seed, dev, cfg)] if workload == "rl" else ([synth.caption_batch(B, L, seed, dev, cfg)] if workload == "caption"
                                                                                    else synth.mixture_batch(B, L, seed, dev, cfg))
        n_patches = _count_patches(batch)
        rows = sum(int(t.label.shape[0]) for t in batch)      # (= B; a tiny debug batch rounds every task of the mixture up to one row)
    except Exception as e:
        err = repr(e)
    if world > 1:
        ok = torch.tensor([0.0 if err else 1.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0:
            return {"tokens_per_s": None, "error": err or "another rank could not build this leg's batch"}
    elif err:
        return {"tokens_per_s": None, "error": err}
    try:

        def step():
            _, loss = engine(batch)
            engine.backward(loss)
            engine.step()
            return loss

        def fence():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        fence()
        engine.time_comm = world > 1
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        fence()
        dt_local = time.perf_counter() - t0
        out = {}
        dt = dt_local
        if world > 1:
            ts = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(ts, torch.tensor([dt_local], device=dev, dtype=torch.float64))
            per_rank = [float(t.item()) for t in ts]
            dt = max(per_rank)
            exposed = torch.tensor([engine.exposed_comm_ms()], device=dev, dtype=torch.float64)
            dist.all_reduce(exposed, op=dist.ReduceOp.MAX)
            pt = torch.tensor([float(n_patches)], device=dev, dtype=torch.float64)
            dist.all_reduce(pt)                       # (every rank builds its own rows: the job's patch count is the sum)
            n_patches_all = int(pt.item())
            out["data_parallel"] = {"ms_per_step_min": round(min(per_rank) / steps * 1e3, 3), "ms_per_step_max": round(max(per_rank) / steps * 1e3, 3),
                                    "exposed_comm_ms_per_step_max": round(float(exposed.item()) / steps, 3)}
        else:
            n_patches_all = n_patches
        engine.time_comm = False
        dt /= steps
        flops = world * rows * L * FLOP_PER_TOKEN + n_patches_all * FLOP_PER_PATCH
        out.update({"workload": f"DB1-1.3B {workload} pre-training step -- {LEG_WORKLOADS[workload]} -- seq_len {L}, {rows} sequences/GPU, same model / engine / dropout / "
                                f"ranks as the text steps",
                    "n_gpus": world, "sequences_per_gpu": rows, "tokens_per_s": round(world * rows * L / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup,
                    "image_patches_per_step": n_patches_all,
                    "pct_mfma_peak_step": round(100.0 * flops / dt / 1e12 / (MFMA_BF16_PEAK_TFLOPS * world), 2), "final_loss": round(float(loss), 4),
                    "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)})
        return out
    except Exception as e:   # a leg must never take the single-rank bench line down
        if world > 1:
            raise
        return {"tokens_per_s": None, "error": repr(e)}


def ga16_leg(model, dev, cfg, L, seed, micro_batch: int = 4, ga: int = 16, steps: int = 3, warmup: int = 1, mode: str = "backward"):
    """The reference's OWN batch geometry (scripts/evaluate/evaluate_rl_1.2B.sh:28-42; train.py:216-232): micro-batch 4 x gradient
    accumulation 16 = 64 sequences per optimizer step on this GPU, on the same model.  ``mode`` "backward" (default): every micro-step's
    FORWARD (+ the fused head / loss sweep: its loss and dh) is a hipGraph replay into the accumulation window's activation buffers and the
    backward of all 16 micro-steps runs ONCE, on the boundary, as one pass over 64 sequences (engine option defer_backward,
    model.BackwardWindow); "wgrad" (round 4 / 5): forward + backward per micro-step as a replay, only the weight gradients once per
    optimizer step from the stashed operands (WgradStash).  Then clip + Adam.  One timed step = one OPTIMIZER step (16 micro-steps);
    ``steps`` of them after ``warmup``."""
    from types import SimpleNamespace
    from bdm_db1_amd import GraphedTrainStep, initialize, synth
    gstep = None
    try:
        model._ctx = None
        model.wgrad_stash = None
        torch.cuda.empty_cache()
        eargs = SimpleNamespace(lr=1e-4, weight_decay=0.01, clip_grad=1.0, optimizer="adam", keep_logits=False, fuse_head_loss=True,
                                gradient_accumulation_steps=ga, defer_wgrad=mode == "wgrad", defer_backward=mode == "backward")
        engine, _, _, _ = initialize(eargs, model)
        engine.train()
        batch = [synth.text_batch(micro_batch, L, seed, dev)]
        gstep = GraphedTrainStep(engine, batch)
        if mode == "backward" and not gstep.window:
            raise RuntimeError("the deferred backward did not take this configuration")

        def opt_step():
            for _ in range(ga):
                loss = gstep(batch)
                engine.step()
            return loss
        for _ in range(warmup):
            opt_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = opt_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        toks = micro_batch * ga * L
        out = {"workload": f"DB1-1.3B text pre-training at the reference's batch geometry: micro-batch {micro_batch} x gradient accumulation {ga} "
                           f"({micro_batch * ga} sequences of {L} tokens per optimizer step; evaluate_rl_1.2B.sh:28-42), " +
                           ("micro-step forwards as hipGraph replays into the accumulation window, ONE backward per optimizer step over the whole window"
                            if mode == "backward" else "micro-steps as hipGraph replays, weight gradients once per optimizer step from the stashed operands") +
                           ", same model / dropout as the text steps", "mode": mode,
               "tokens_per_s": round(toks / dt, 1), "ms_per_optimizer_step": round(dt * 1e3, 3), "optimizer_steps": steps, "warmup": warmup,
               "micro_batch": micro_batch, "grad_accumulation": ga,
               "pct_mfma_peak_step": round(100.0 * toks * FLOP_PER_TOKEN / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, 2), "final_loss": round(float(loss), 4),
               "stash_gib": round(model.wgrad_stash.nbytes() / 2**30, 1) if model.wgrad_stash is not None else None,
               "window_gib": round(model._win.nbytes() / 2**30, 1) if (mode == "backward" and model._win is not None) else None,
               "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)}
    except Exception as e:   # the leg must never take the bench line down
        out = {"tokens_per_s": None, "error": repr(e)}
    if gstep is not None:
        gstep.close()
    model.wgrad_stash = None
    model.wgrad_defer_ga = 0
    model._win = None
    model.bwd_window_ga = 0
    model._ctx = None
    torch.cuda.empty_cache()
    return out


def rocprof_crosscheck(family_kernels_note: str):
    """the newest committed per-kernel table of a rocprofv3 kernel trace cut to the timed steps of THIS command (tools/prof_step.sh ->
    tools/prof_table.py -> profiles/r*_step_table.json): the same GEMM-family fraction, computed from the profiler's kernel durations"""
    import glob
    import re
    # rNN[x]_step_table.json only: tables of other workloads carry the workload in their name (r06k_rl_step_table.json)
    cands = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_step_table.json"))
                   if re.fullmatch(r"r\d+[a-z]?_step_table\.json", os.path.basename(f)))
    if not cands:
        return None
    try:
        rec = json.load(open(cands[-1]))
        fam = rec["families"]["gemm"]
        return {"frac": fam["frac"], "achieved": fam["achieved"], "ms_per_step": fam["ms_per_step"], "source": os.path.relpath(cands[-1], ROOT),
                "note": "rocprofv3 --kernel-trace of an earlier run of this command on another box of the pool, cut to the timed steps by marker kernels; "
                        "the same kernels as `frac`: " + family_kernels_note}
    except Exception:
        return None


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N copies of this script, one rank per GPU, with the
    rendezvous environment torch.distributed.run would have set (127.0.0.1, a free port).  Rank 0's stdout is ours, so
    its ONE JSON line is the output.  Every rank's stderr goes to a temporary file; when a rank fails (or the job exceeds
    DB1_LAUNCH_TIMEOUT_S, default 1800 s) the others are stopped by PID and the tail of each failing rank's stderr is shown."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmo = float(os.environ.get("DB1_LAUNCH_TIMEOUT_S", 1800))
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # dmabuf IPC: this image's host driver supports no other kind, and without the variable RCCL's intra-node transport (and any
        # sharing of device tensors across processes) fails with `hipIpcGetMemHandle: invalid argument` (the environment notes of this
        # build; already exported on the GPU boxes -- kept so that a caller's cleaned environment cannot lose it)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        log = tempfile.NamedTemporaryFile(mode="w+", prefix=f"db1_bench_rank{r}_", suffix=".err", delete=False)
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=log))
    rc, failed = 0, []
    alive = list(procs)
    t0 = time.time()
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                failed.append(procs.index(p))
                if rc == 0:
                    rc = code
                    for q in alive:       # a dead rank leaves the others waiting in a collective forever
                        q.terminate()
        if alive and time.time() - t0 > tmo:
            rc = rc or 124
            failed += [procs.index(q) for q in alive]
            print(f"bench.py: {len(alive)} rank(s) still running after {tmo:.0f} s (DB1_LAUNCH_TIMEOUT_S): stopping them", file=sys.stderr)
            for q in alive:
                q.terminate()
            for q in alive:
                try:
                    q.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    q.kill()
            alive = []
        time.sleep(0.2)
    for r, log in enumerate(logs):
        log.flush()
        log.seek(0)
        txt = log.read()
        log.close()
        if r in failed and txt.strip():
            print(f"---- rank {r} stderr (tail) ----\n{txt[-3000:]}", file=sys.stderr)
        elif r == 0 and txt.strip() and rc == 0 and os.environ.get("DB1_LAUNCH_VERBOSE"):
            print(txt[-2000:], file=sys.stderr)
        try:
            os.unlink(log.name)
        except OSError:
            pass
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DB1_BENCH_BATCH", 64)), help="sequences per GPU per step")
    ap.add_argument("--ga", type=int, default=1, help="gradient-accumulation micro-steps per optimizer step (the reference trains micro-batch 4 x GA 16, "
                                                      "scripts/evaluate/evaluate_rl_1.2B.sh:28-42): one timed step = GA x (fwd + bwd) + clip + Adam")
    ap.add_argument("--graph", action="store_true", help="forward + backward of a micro-step as one hipGraph replay (bdm_db1_amd.GraphedTrainStep): what small "
                                                         "micro-batches need (at 4 sequences the eager step is host-bound); no per-kernel timing in this mode")
    ap.add_argument("--defer-backward", action="store_true", help="with --ga > 1: ONE backward per optimizer step over the whole accumulation window (engine option "
                    "defer_backward, model.BackwardWindow); the forwards of the micro-steps write into window-sized activation buffers")
    ap.add_argument("--ga16-mode", default=os.environ.get("DB1_GA16_MODE", "backward"), choices=["backward", "wgrad"], help="the ga16 leg: one backward per window / per-micro-step backward with deferred weight gradients")
    ap.add_argument("--no-defer-wgrad", action="store_true", help="with --ga > 1: form the weight gradients per micro-step (K = micro-batch tokens, fp32 accumulate) instead of "
                                                                  "once per optimizer step from the stashed operands of all micro-steps (bdm_db1_amd WgradStash)")
    ap.add_argument("--layers", type=int, default=24, help="debug only: anything but 24 is not the benchmark config")
    ap.add_argument("--workload", default="text", choices=["text", "caption", "rl", "mixture"])
    ap.add_argument("--dropout", type=float, default=0.1, help="drop = embd_pdrop of the training step (the reference's defaults, src/config.py:123,161: 0.1)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the inference-with-memory leg after the timed steps")
    ap.add_argument("--no-mixture", action="store_true", help="skip the RL-trajectory and mixed-modal legs (BASELINE configs 4 / 5: a few steps of each on the same model and ranks) after the timed steps")
    ap.add_argument("--leg-steps", type=int, default=5, help="timed steps of each extra workload leg")
    ap.add_argument("--no-box", action="store_true", help="skip the box calibration probes (in-register MFMA rate, device copy, rocm-smi sample) before / after the timed steps")
    ap.add_argument("--no-ga16", action="store_true", help="skip the leg at the reference's batch geometry (micro-batch 4 x GA 16, graphed micro-steps, 3 optimizer steps; N = 1 only)")
    ap.add_argument("--no-flash", action="store_true")
    ap.add_argument("--flash-probs", choices=["forward", "scratch", "recompute"], default="forward",
                    help="A/B: what the flash backward recomputes (nothing: the forward keeps p~ per layer / the query side only / both sides)")
    ap.add_argument("--materialise-logits", action="store_true", help="A/B: head GEMM + CE on a full (tokens x vocabulary) logits buffer instead of the chunked sweep")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if "WORLD_SIZE" in os.environ and os.environ.get("DB1_BENCH_FAIL_RANK") == str(rank):   # test hook: tests/test_dp_gpu.py (failure reporting of self_launch)
        raise SystemExit(f"DB1_BENCH_FAIL_RANK={rank}: this rank was asked to fail")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # DB1_DIST_BACKEND=gloo lets the multi-rank path be exercised on a single-GPU box (ranks share the device; RCCL refuses that)
    backend = os.environ.get("DB1_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        from bdm_db1_amd.engine import init_distributed
        init_distributed(dist_backend=backend)   # "nccl" is RCCL on ROCm: one process per GPU over xGMI, high-priority comm stream

    from bdm_db1_amd import TransformerXL, initialize, lib, mpu, ops, synth
    lib.apply_env_knobs()      # (A/B runs only: DB1_* dispatcher knobs of include/db1_hip_test.h; nothing is set in a default run)
    from types import SimpleNamespace
    if world > 1:
        mpu.initialize_model_parallel()
    cfg = synth.db1_config("1.3B", n_layer=args.layers, drop=args.dropout, embd_pdrop=args.dropout)
    torch.manual_seed(1234)
    if world > 1:   # the ranks build their 22 GB of parameters / optimizer state one after the other (host-side allocation and first-touch
        time.sleep(float(os.environ.get("DB1_LAUNCH_STAGGER_S", 0.5)) * rank)   # work of eight processes at once helps nobody)
    model = TransformerXL(cfg, device=dev)
    model.use_flash = not args.no_flash
    model.flash_probs_mode = args.flash_probs
    eargs = SimpleNamespace(lr=1e-4, weight_decay=0.01, clip_grad=1.0, optimizer="adam", keep_logits=False, fuse_head_loss=not args.materialise_logits,
                            gradient_accumulation_steps=args.ga, defer_wgrad=args.ga > 1 and not args.no_defer_wgrad,
                            defer_backward=args.ga > 1 and args.defer_backward)
    engine, _, _, _ = initialize(eargs, model, mpu=mpu if world > 1 else None)
    engine.train()
    B, L = args.batch, cfg.n_position
    seed = 1234 + rank
    if args.workload == "text":
        batch = [synth.text_batch(B, L, seed, dev)]
    elif args.workload == "caption":
        batch = [synth.caption_batch(B, L, seed, dev, cfg)]
    elif args.workload == "rl":
        batch = [synth.rl_batch(B, L, seed, dev, cfg)]
    else:
        batch = synth.mixture_batch(B, L, seed, dev, cfg)
    n_patches = _count_patches(batch)

    gstep = None
    if args.graph:
        from bdm_db1_amd import GraphedTrainStep
        gstep = GraphedTrainStep(engine, batch)
        args.no_kernel_timing = True

    def step():
        for _ in range(args.ga):   # train.py:216-232: GA micro-steps of engine(x) -> backward -> step; the optimizer runs on the boundary
            if gstep is not None:
                loss = gstep(batch)
            else:
                logits, loss = engine(batch)
                engine.backward(loss)
            engine.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    box = box_calibration(dev) if (rank == 0 and not args.no_box) else None     # (before the warm-up: an idle, cool box; the same probe again after the timed steps)
    smi = _smi_sample_start() if (box is not None and args.warmup > 0) else None   # clock / power as rocm-smi sees them WHILE the warm-up steps run
    for _ in range(args.warmup):
        loss = step()
    fence()
    if box is not None:
        box["sclk_mhz_during_warmup"], box["power_w_during_warmup"] = _smi_sample_collect(smi, dev.index or 0)
    # HIP-event pairs inside the timed region only around the launches of the ROOFLINE family (every tile GEMM + the head sweep): an event pair
    # per launch costs the step ~0.7 % when all ~700 launches of a step carry one (measured, same box: 424.2 vs 420.5 ms), so the other
    # families' rooflines (`kernels`) are taken on two extra, untimed steps after the timed region
    timer = None if args.no_kernel_timing else ops.KernelTimer(only=("gemm", "lmhead_ce"))
    ops.set_gemm_timer(timer)
    engine.time_comm = world > 1          # event pairs around the wait for the bucket all-reduces (GradSync.finish): the exposed communication
    ops.marker(1)                         # empty marker kernels: tools/prof_table.py cuts a rocprofv3 kernel trace to the timed steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    ops.marker(2)
    fence()
    dt_local = time.perf_counter() - t0
    ops.set_gemm_timer(None)
    dt, dp_info = dt_local, None
    if world > 1:
        ts = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(ts, torch.tensor([dt_local], device=dev, dtype=torch.float64))
        per_rank = [float(t.item()) for t in ts]
        dt = max(per_rank)                                    # the job is as slow as its slowest rank
        exposed = torch.tensor([engine.exposed_comm_ms()], device=dev, dtype=torch.float64)
        dist.all_reduce(exposed, op=dist.ReduceOp.MAX)
        dp_info = {"ms_per_step_per_rank": [round(t / args.steps * 1e3, 3) for t in per_rank],
                   "ms_per_step_min": round(min(per_rank) / args.steps * 1e3, 3), "ms_per_step_max": round(max(per_rank) / args.steps * 1e3, 3),
                   "exposed_comm_ms_per_step_max": round(float(exposed.item()) / args.steps, 3),
                   "note": "exposed communication = time the compute stream waits in GradSync.finish for bucket all-reduces that were launched from the "
                           "backward (HIP events around the waits, max over ranks); gradients cross xGMI in bf16, 25 per-layer buckets + embeddings"}
    engine.time_comm = False
    timer_rest, rest_steps = None, 2
    if timer is not None:
        timer_rest = ops.KernelTimer()
        ops.set_gemm_timer(timer_rest)
        for _ in range(rest_steps):
            step()
        fence()
        ops.set_gemm_timer(None)
    loss_v = float(loss)
    if box is not None:
        after = box_calibration(dev)
        box["mfma_random_tf_after"], box["copy_gbps_after"] = after.get("mfma_random_tf"), after.get("copy_gbps")

    peak_gb = torch.cuda.max_memory_allocated(dev) / 2**30
    tokens = world * B * L * args.steps * args.ga
    tok_s = tokens / dt
    flops_step_all = world * args.ga * (B * L * FLOP_PER_TOKEN + n_patches * FLOP_PER_PATCH)
    out = {
        "metric": "pretrain tokens/sec (whole node) DB1-1.3B seq1024", "value": round(tok_s, 1), "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"DB1-1.3B {args.workload} causal LM pre-training step (fwd+bwd+clip+Adam, training mode: dropout "
                               f"{args.dropout:g} on embeddings / attention / feed-forward outputs as the reference's defaults), seq_len 1024, "
                               f"{B} sequences/GPU/micro-step x {args.ga} micro-step(s) per optimizer step, random-init weights", "dropout": args.dropout, "n_layer": args.layers, "n_embed": 2048, "n_head": 16,
                   "seq_len": L, "batch_per_gpu": B, "grad_accumulation": args.ga, "micro_step_as_hipgraph": bool(args.graph), "weight_gradients": ("once per optimizer step from the stashed operands of all micro-steps" if getattr(engine, "defer_wgrad", False) else ("one backward per optimizer step over the whole accumulation window" if getattr(engine, "defer_backward", False) else "per micro-step")), "global_batch": B * world * args.ga, "parallelism": f"dp{world}",
                   "attention_backward": {"forward": "nothing recomputed (the forward keeps its probabilities)", "scratch": "query side recomputes, P / dS through scratch",
                                          "recompute": "both sides recompute"}[model._probs_mode(B, L)],
                   "params": int(sum(int(np.prod(s)) for _, s, _ in model.arena.offsets.values()))},
        "pct_mfma_peak_step": round(100.0 * flops_step_all / (dt / args.steps) / 1e12 / (MFMA_BF16_PEAK_TFLOPS * world), 2),
        "final_loss": round(loss_v, 4),
        "peak_hbm_gib": round(peak_gb, 1),
    }
    if dp_info is not None:
        out["data_parallel"] = dp_info
    if box is not None:
        out["box"] = box
        if box.get("mfma_random_tf"):
            out["value_normalised"] = round(tok_s * (POOL_MFMA_RANDOM_TF / box["mfma_random_tf"]) ** BOX_RATE_EXPONENT, 1)
            out["value_normalised_note"] = (f"value x ({POOL_MFMA_RANDOM_TF:g} / box.mfma_random_tf) ^ {BOX_RATE_EXPONENT:g}: tokens/s this run would show on the pool's median "
                                            "box (median in-register MFMA rate of the boxes recorded in profiles/r06_box_calibration.txt; the exponent is the slope "
                                            "ln tokens/s against ln rate fitted over those records) -- an attribution aid for deltas between records; `value` is the measurement")
    summ = timer.summary() if timer is not None else {}
    if "gemm" in summ:
        # the dominant kernel family = EVERY bf16 tile-GEMM launch of the step: the decoder layers' products and the three head products of
        # the chunked head + loss sweep (db1_lmhead_ce_fwd_bwd sequences them with its loss kernel behind ONE C call, so that call is timed
        # as a whole: the ~2 ms per step of ce_fwd_bwd_kernel / ce_sum_kernel ride along, on both sides of the cross-check)
        ms, flops, launches = summ["gemm"]
        ms_dec, flops_dec, launches_dec = ms, flops, launches
        if "lmhead_ce" in summ:
            hms, hflops, hn = summ["lmhead_ce"]
            ms, flops, launches = ms + hms, flops + hflops, launches + hn
        ach = flops / (ms * 1e-3) / 1e12
        traffic, traffic_src, traffic_box = None, None, None
        try:  # HBM-side bytes per tile-GEMM launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read live)
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc.json")))
            if cands and args.workload == "text":
                rec = json.load(open(cands[-1]))
                if rec.get("batch_per_gpu", 16) == B:  # bytes per launch scale with the batch: only the matching recording applies
                    traffic = rec["tile_gemm_avg_bytes_per_launch"]
                    traffic_src = os.path.relpath(cands[-1], ROOT) + " (PMC passes of an earlier run of this command, not this run)"
                    traffic_box = rec.get("box")        # that run's box calibration (None for records older than round 6)
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch (HBM-side, PMC)",
                           "traffic_source": traffic_src, "traffic_box": traffic_box,
                           "kernel": FAMILY_NOTE,
                           "launches": launches, "avg_launch_us": round(ms * 1e3 / launches, 2),
                           "flop_per_launch_avg": round(flops / launches),
                           "flop_per_step": round(flops / args.steps), "ms_per_step": round(ms / args.steps, 3),
                           "kernel_time_share_of_step": round(ms / (dt * 1e3), 4),
                           "decoder_layers_only": {"achieved": round(flops_dec / (ms_dec * 1e-3) / 1e12, 1), "frac": round(flops_dec / (ms_dec * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                                   "launches": launches_dec, "ms_per_step": round(ms_dec / args.steps, 3)},
                           "geglu_epilogues": bool(getattr(model, "use_geglu_epilogue", False)), "geglu_note": GEGLU_NOTE,
                           "rocprof": rocprof_crosscheck(FAMILY_NOTE) if args.workload == "text" and B == 64 and args.ga == 1 else None}
        # the other kernels of the step against THEIR rooflines: algorithmic FLOPs or bytes / HIP-event time on the launch stream
        ks = {}
        rest = timer_rest.summary() if timer_rest is not None else {}
        step_ms_rest = dt / args.steps * 1e3 * rest_steps     # (share_of_step of these families: against the timed region's ms per step)
        for fam, (fms, work, n) in sorted(rest.items()):
            if fam in ("gemm",) or fms <= 0:
                continue
            # (lmhead_ce stays listed on its own as well: it is part of the roofline family above)
            mfma = fam.startswith("flash") or fam == "lmhead_ce"
            rate = work / (fms * 1e-3) / (1e12 if mfma else 1e9)
            peak = MFMA_BF16_PEAK_TFLOPS if mfma else HBM_PEAK_GBPS
            ks[fam] = {"bound": "mfma" if mfma else "hbm", "achieved": round(rate, 1), "unit": "TFLOP/s" if mfma else "GB/s",
                       "frac": round(rate / peak, 4), "avg_us": round(fms * 1e3 / n, 1), "launches": n,
                       "share_of_step": round(fms / step_ms_rest, 4)}
        # the attention kernels are streams since DESIGN 4a: beside the MFMA fraction of their algorithmic FLOPs, the HBM-side bytes per launch of
        # the committed PMC passes (same workload, an earlier run) over the live HIP-event time
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc.json")))
            rec = json.load(open(cands[-1])) if cands and args.workload == "text" else None
            if rec and rec.get("batch_per_gpu", 16) == B:
                pk = rec["kernels"]
                groups = {"flash_fwd": ["relattn_flash_fwd_kernel"], "flash_bwd": ["relattn_flash_bwd_q", "relattn_flash_bwd_kv"]}
                mode_tag = {"forward": ("<true>", "q2_kernel", "kv2_kernel<true>"), "scratch": ("<false>", "q_kernel<true>", "kv2_kernel<false>"),
                            "recompute": ("<false>", "q_kernel<false>", "kv_kernel")}[model._probs_mode(B, L)]
                for fam, names in groups.items():
                    if fam not in ks:
                        continue
                    want = [mode_tag[0]] if fam == "flash_fwd" else list(mode_tag[1:])
                    tot = sum(v["hbm_side_bytes_per_launch"] for k, v in pk.items() if any(k.startswith(nm) for nm in names) and any(w in k for w in want))
                    if tot > 0:
                        gbs = tot / (ks[fam]["avg_us"] * 1e-6) / 1e9
                        ks[fam].update({"hbm_side_bytes_per_launch": int(tot), "hbm_gbps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBPS, 4),
                                        "hbm_source": os.path.relpath(cands[-1], ROOT)})
        except Exception:
            pass
        out["roofline"]["rocprof_frac"] = out["roofline"]["rocprof"]["frac"] if out["roofline"]["rocprof"] else None
        # work per step of every timed family (FLOPs or bytes): what tools/prof_table.py divides the profiler's kernel durations into
        out["work_per_step"] = {fam: w / rest_steps for fam, (_, w, _) in rest.items()}
        out["work_per_step"].update({fam: w / args.steps for fam, (_, w, _) in summ.items()})
        out["kernels_note"] = (f"`kernels`: every other kernel family against its own roofline, HIP events on {rest_steps} extra steps run after the timed region "
                               "(an event pair around each of a step's ~700 launches costs the step ~0.7 %; inside the timed region only the roofline family carries them)")
        out["kernels"] = ks
    if gstep is not None:
        gstep.close()
    # BASELINE configs 4 and 5 (RL trajectories, the mixture) on the same model / engine / RANKS right after the text steps, so that an N-GPU
    # run of this command yields their whole-job tokens/s at N too (every rank takes part: the legs contain the step's collectives)
    if not args.no_mixture and args.workload == "text" and args.ga == 1 and gstep is None:
        for wl, k in (("rl", 200), ("mixture", 100)):
            leg = workload_leg(engine, model, dev, cfg, B, L, seed + k, wl, world=world, steps=args.leg_steps, warmup=2)
            if rank == 0:
                out[wl] = leg
    if rank == 0 and world == 1 and not args.no_ga16 and args.workload == "text" and args.ga == 1 and gstep is None and args.layers == 24:
        out["ga16"] = ga16_leg(model, dev, cfg, L, seed + 300, mode=args.ga16_mode)
    if rank == 0 and world == 1 and not args.no_decode and args.layers == 24:
        out["decode"] = decode_leg(model, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the baseline leg must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
    if args.layers != 24:
        out["INVALID"] = "debug run: n_layer != 24 is not the benchmark configuration"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
