cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dropout.py tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_geometry_gpu.py -x -q > gpurun_out/r04g_tests.log 2>&1; tail -5 gpurun_out/r04g_tests.log
timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 6 --warmup 2 > gpurun_out/r04g_bench.json 2> gpurun_out/r04g_bench.err
timeout 300 python bench.py --batch 4 --ga 16 --graph --steps 4 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/r04g_bench_b4.json 2> gpurun_out/r04g_bench_b4.err
timeout 300 python bench.py --batch 8 --ga 8 --graph --steps 4 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/r04g_bench_b8.json 2> gpurun_out/r04g_bench_b8.err
python - <<'PY'
import json
for n in ("bench","bench_b4","bench_b8"):
    try:
        d=json.loads(open(f"gpurun_out/r04g_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["pct_mfma_peak_step"])
    except Exception as e:
        print(n, "failed", e, open(f"gpurun_out/r04g_{n}.err").read()[-800:])
PY
