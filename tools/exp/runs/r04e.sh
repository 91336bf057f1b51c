cd $GRAFT_REPO_ROOT
python -m pytest tests/test_geglu_epilogue_gpu.py -x -q > gpurun_out/r04e_geglu.log 2>&1; tail -5 gpurun_out/r04e_geglu.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 6 --warmup 2 > gpurun_out/r04e_bench_fused$i.json 2> gpurun_out/r04e_bench_fused.err
DB1_GEGLU_EPI=0 timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 6 --warmup 2 > gpurun_out/r04e_bench_unfused$i.json 2> gpurun_out/r04e_bench_unfused.err
done
python - <<'PY'
import json
for n in ("fused1","unfused1","fused2","unfused2"):
    try:
        d=json.loads(open(f"gpurun_out/r04e_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v["avg_us"],v["launches"]) for k,v in d["kernels"].items() if k.startswith("ffn")})
    except Exception as e:
        print(n, "failed", e)
PY
bash tools/prof_step.sh > /dev/null 2>&1; head -12 gpurun_out/step_table.txt
