cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dropout.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 6 --warmup 2 > gpurun_out/r04i_bench.json 2> gpurun_out/r04i_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04i_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k:(v["avg_us"],v["launches"],v["frac"]) for k,v in d["kernels"].items() if "layernorm" in k})
PY
