cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_w4n_gpu.py tests/test_flash_gpu.py -x -q 2>&1 | tail -5
python tools/bench_kernels.py small_out 64 2>&1 | grep -i "dR" 
DB1_W4N=0 python tools/bench_kernels.py small_out 64 2>&1 | grep -i "dR"
timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 6 --warmup 2 > gpurun_out/r04s_bench.json 2> gpurun_out/r04s_bench.err
DB1_W4N=0 timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 6 --warmup 2 > gpurun_out/r04s_bench_now4n.json 2> gpurun_out/r04s_bench_now4n.err
python - <<'PY'
import json
for n in ("bench","bench_now4n"):
    try:
        d=json.loads(open(f"gpurun_out/r04s_{n}.json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d["pct_mfma_peak_step"])
    except Exception as e:
        print(n, "failed", e, open(f"gpurun_out/r04s_{n}.err").read()[-800:])
PY
