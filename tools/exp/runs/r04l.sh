cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r04l_pytest.log 2>&1; tail -6 gpurun_out/r04l_pytest.log
timeout 400 python bench.py --batch 4 --ga 16 --steps 3 --warmup 1 --no-cpu-baseline --no-decode > gpurun_out/r04l_b4e.json 2> gpurun_out/r04l_b4e.err
timeout 400 python bench.py --no-cpu-baseline --no-decode --steps 6 --warmup 2 > gpurun_out/r04l_b64.json 2> gpurun_out/r04l_b64.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04l_b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["pct_mfma_peak_step"], d["peak_hbm_gib"], d["config"]["weight_gradients"], d["roofline"]["frac"], d.get("mixture",{}).get("tokens_per_s"))
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json",".err")).read()[-1200:])
PY
