cd $GRAFT_REPO_ROOT
python -m pytest tests/test_geglu_epilogue_gpu.py -x -q > gpurun_out/r04b_geglu.log 2>&1; tail -15 gpurun_out/r04b_geglu.log
timeout 900 python tools/exp/depth_error.py 4 default > gpurun_out/r04b_depth.txt 2>&1; cat gpurun_out/r04b_depth.txt | tail -8
timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 6 --warmup 2 > gpurun_out/r04b_bench_fused.json 2> gpurun_out/r04b_bench_fused.err
DB1_GEGLU_EPI=0 timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 6 --warmup 2 > gpurun_out/r04b_bench_unfused.json 2> gpurun_out/r04b_bench_unfused.err
python - <<'PY'
import json
for n in ("fused","unfused"):
    try:
        d=json.loads(open(f"gpurun_out/r04b_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v["avg_us"],v["launches"]) for k,v in d["kernels"].items() if k.startswith("ffn")})
    except Exception as e:
        print(n, "failed", e, open(f"gpurun_out/r04b_bench_{n}.err").read()[-1500:])
PY
