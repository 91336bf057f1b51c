cd $GRAFT_REPO_ROOT
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode --no-mixture > gpurun_out/r04z_bench.json 2>gpurun_out/r04z_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04z_bench.json"))
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items():
    if isinstance(v, dict) and any(t in k for t in ("layernorm","ce","sumsq","adam")): print(k, v)
PY
