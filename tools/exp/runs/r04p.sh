cd $GRAFT_REPO_ROOT
echo "== gemm 4: w4n=3"; DB1_W4N=3 timeout 300 python tools/bench_kernels.py gemm 4 2>&1 | grep -v "^/opt" | head -12
echo "== gemm 8: w4n=1"; timeout 300 python tools/bench_kernels.py gemm 8 2>&1 | grep -v "^/opt" | head -12
echo "== gemm 8: w4n=3"; DB1_W4N=3 timeout 300 python tools/bench_kernels.py gemm 8 2>&1 | grep -v "^/opt" | head -12
timeout 400 python bench.py --batch 4 --ga 16 --graph --steps 4 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/r04p_b4g.json 2> gpurun_out/r04p_b4g.err
timeout 400 python bench.py --batch 8 --ga 8 --graph --steps 4 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/r04p_b8g.json 2> gpurun_out/r04p_b8g.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04p_b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["pct_mfma_peak_step"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json",".err")).read()[-1200:])
PY
