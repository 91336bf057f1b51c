"""HBM-side bytes per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over a short bench run.
Run on the GPU box:   python tools/pmc_traffic.py gpurun_out/traffic            (writes <dir>/hbm_traffic_pmc.json)
bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024  (gfx950: FETCH_SIZE tallies 64 B per 128-B request; calibrated on adam_kernel
and ce_fwd, see the "method" field).  One counter per pass, kernel-trace only (no other trace domain)."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "traffic"))
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out_dir, ctr.lower())
    subprocess.run(["rm", "-rf", d])
    cmd = ["timeout", "600", "rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-decode", "--no-mixture", "--no-ga16"]   # (hipGraph replays under counter collection abort the queue)
    with open(os.path.join(out_dir, ctr.lower() + ".log"), "w") as lf:
        subprocess.run(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL, stdout=lf, stderr=subprocess.STDOUT)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for row in csv.DictReader(open(f[0], newline="")):
            if row["Counter_Name"] != ctr:
                continue
            name = re.sub(r"^void ", "", row["Kernel_Name"]).split("(")[0]
            name = re.sub(r"<.*", "", name) if name.startswith("at::") else name
            acc[name[:90]].append(float(row["Counter_Value"]))
    vals[ctr] = acc
kernels = {}
for name in sorted(set(vals["FETCH_SIZE"]) | set(vals["WRITE_SIZE"])):
    fv, wv = vals["FETCH_SIZE"].get(name, []), vals["WRITE_SIZE"].get(name, [])
    n = max(len(fv), len(wv))
    if not n:
        continue
    fb = 2 * 1024 * sum(fv) / max(len(fv), 1)
    wb = 1024 * sum(wv) / max(len(wv), 1)
    kernels[name] = {"launches": n, "hbm_side_bytes_per_launch": round(fb + wb), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb)}
gemm = [(k, v) for k, v in kernels.items() if k.startswith("gemm_bf16_")]
tot_l = sum(v["launches"] for _, v in gemm)
res = {
    "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
              "--no-kernel-timing --no-decode` (default batch, recorded in batch_per_gpu); bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request; "
              "calibrated on adam_kernel: 16 B/param read, 14 B/param written, and on ce_fwd: T x vocab_pad x 2 B read)",
    "kernels": kernels,
    "tile_gemm_avg_bytes_per_launch": round(sum(v["hbm_side_bytes_per_launch"] * v["launches"] for _, v in gemm) / tot_l) if tot_l else None,
    "tile_gemm_launches": tot_l,
    "batch_per_gpu": int(os.environ.get("DB1_BENCH_BATCH", 64)),
}
# the box these passes ran on (bench.py `box`: in-register MFMA rate, device copy): bench.py prints it beside the traffic figure it takes from here
try:
    r = subprocess.run([sys.executable, "-c", "import sys, json, torch; sys.path.insert(0, %r); import bench; print(json.dumps(bench.box_calibration(torch.device('cuda', 0))))" % ROOT],
                       cwd="/tmp", env=env, stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=300)
    res["box"] = json.loads(r.stdout.strip().splitlines()[-1])
except Exception as e:
    res["box"] = {"error": repr(e)}
json.dump(res, open(os.path.join(out_dir, "hbm_traffic_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"})[:600])
