"""Experiment: shader clock and socket power while one kernel runs back to back (rocm-smi sampled from a side thread).
Usage: python tools/exp/exp_clocks.py     (needs a GPU; prints one line per workload)"""
import os, subprocess, sys, threading, time, re
import os, sys
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))   # tools/ (bench_kernels) and the repository root
import torch
from bdm_db1_amd import ops

DEV = "cuda"


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True, timeout=20).stdout
        except Exception as e:  # noqa: BLE001
            txt = str(e)
        sclk = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", txt)
        pw = re.search(r"Power \(W\):\s*([\d.]+)", txt)
        out.append((int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        if len(out) == 1 and os.environ.get("EXP_RAW"):
            print(txt)


def run(name, fn, secs=6.0):
    fn(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); n = 0
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    ms = e0.elapsed_time(e1) / n
    out = out[1:] if len(out) > 2 else out
    clk = [c for c, _ in out if c > 0]; pw = [p for _, p in out if p > 0]
    print(f"{name:28s} {ms * 1e3:9.1f} us/launch   sclk {min(clk) if clk else -1}-{max(clk) if clk else -1} MHz   power {min(pw) if pw else -1:.0f}-{max(pw) if pw else -1:.0f} W   ({len(out)} samples)")
    return ms


M, N, K = 65536, 8192, 2048
x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
t = run("idle (sleep)", lambda: time.sleep(0.01), 3.0)
t = run("ff1 NT, random data", lambda: ops.gemm(x, w.t(), y))
print(f"    -> {2 * M * N * K / t / 1e9:.0f} TFLOP/s")
xz, wz = torch.zeros_like(x), torch.zeros_like(w)
t = run("ff1 NT, zero data", lambda: ops.gemm(xz, wz.t(), y))
print(f"    -> {2 * M * N * K / t / 1e9:.0f} TFLOP/s")
t = run("ff1 NT, torch (hipBLASLt)", lambda: torch.matmul(x, w.t(), out=y))
print(f"    -> {2 * M * N * K / t / 1e9:.0f} TFLOP/s")
a = torch.randn(M, N, device=DEV).to(torch.bfloat16); b = torch.empty(M, N // 2, device=DEV, dtype=torch.bfloat16)
t = run("GEGLU forward (HBM-bound)", lambda: ops.ffn_act_fwd(a, b, "geglu"))
print(f"    -> {(M * N * 2 + M * N) / t / 1e9:.2f} TB/s")

# the relative-position flash attention kernels (64 sequences x 1024 tokens x 16 heads x 128)
Bq, L, H, D = 64, 1024, 16, 128
qkv5 = torch.randn(Bq, L, 3, H, D, device=DEV).to(torch.bfloat16)
qu = torch.randn(Bq, L, H, D, device=DEV).to(torch.bfloat16); qv = torch.randn(Bq, L, H, D, device=DEV).to(torch.bfloat16)
R = torch.randn(L, H, D, device=DEV).to(torch.bfloat16)
o = torch.empty(Bq, L, H, D, device=DEV, dtype=torch.bfloat16); lse = torch.empty(Bq, H, L, device=DEV)
sc = 1.0 / D ** 0.5
t = run("flash forward", lambda: ops.relattn_flash_fwd(qu, qv, qkv5, R, o, lse, Bq, L, H, D, L, sc))
print(f"    -> {3 * 2 * Bq * H * (L * (L + 1) / 2) * D / t / 1e9:.0f} TFLOP/s (3 causal contractions)")
do = torch.randn_like(o); delta = torch.empty(Bq, H, L, device=DEV); dqkv5 = torch.empty_like(qkv5); dT = torch.empty(Bq, H, L, L, device=DEV, dtype=torch.bfloat16)
t = run("flash backward (q + kv)", lambda: ops.relattn_flash_bwd(qu, qv, qkv5, R, o, do, lse, delta, dqkv5, dT, Bq, L, H, D, L, sc))
print(f"    -> {9 * 2 * Bq * H * (L * (L + 1) / 2) * D / t / 1e9:.0f} TFLOP/s (9 causal contractions incl. the recomputed ones)")
