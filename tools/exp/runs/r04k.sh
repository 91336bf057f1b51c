cd $GRAFT_REPO_ROOT
python -m pytest tests/test_defer_wgrad_gpu.py -x -q > gpurun_out/r04k_tests.log 2>&1; tail -15 gpurun_out/r04k_tests.log
for a in "" "--no-defer-wgrad"; do
timeout 400 python bench.py --batch 4 --ga 16 --graph --steps 4 --warmup 2 --no-cpu-baseline --no-decode $a > gpurun_out/r04k_b4g$a.json 2> gpurun_out/r04k_b4g$a.err
timeout 400 python bench.py --batch 4 --ga 16 --steps 3 --warmup 1 --no-cpu-baseline --no-decode $a > gpurun_out/r04k_b4e$a.json 2> gpurun_out/r04k_b4e$a.err
done
timeout 400 python bench.py --batch 8 --ga 8 --graph --steps 4 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/r04k_b8g.json 2> gpurun_out/r04k_b8g.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04k_b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["pct_mfma_peak_step"], d["peak_hbm_gib"], d["config"]["weight_gradients"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json",".err")).read()[-1200:])
PY
