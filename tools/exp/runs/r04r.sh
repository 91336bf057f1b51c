cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r04r_pytest.log 2>&1; tail -4 gpurun_out/r04r_pytest.log
timeout 400 python bench.py --batch 4 --ga 16 --graph --steps 4 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/r04r_b4g.json 2> gpurun_out/r04r_b4g.err
timeout 400 python bench.py --batch 8 --ga 8 --graph --steps 4 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/r04r_b8g.json 2> gpurun_out/r04r_b8g.err
timeout 400 python bench.py --batch 4 --ga 16 --steps 3 --warmup 1 --no-cpu-baseline --no-decode > gpurun_out/r04r_b4e.json 2> gpurun_out/r04r_b4e.err
timeout 400 python bench.py --batch 16 --ga 4 --graph --steps 4 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/r04r_b16g.json 2> gpurun_out/r04r_b16g.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04r_b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["pct_mfma_peak_step"], d["peak_hbm_gib"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json",".err")).read()[-1200:])
PY
