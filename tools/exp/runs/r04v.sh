cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/evidence_r04d
s=$(date +%s)
timeout 900 python bench.py > gpurun_out/evidence_r04d/r04d_bench_default.json 2> gpurun_out/evidence_r04d/bench_default.err </dev/null
e=$(date +%s)
echo "default bench wall seconds: $((e-s))"
python - <<PY
import json
d=json.load(open("gpurun_out/evidence_r04d/r04d_bench_default.json"))
print(d["value"], d["ms_per_step"], d.get("pct_mfma_peak_step"), d["roofline"]["frac"], d["roofline"].get("rocprof_frac"))
c=d["cpu_baseline"]; print(c["value"], c["port"], c["numpy_oracle"], {k:v for k,v in c["torch_eager_port"].items() if k!="sample"})
print(d["mixture"]["tokens_per_s"], d["decode"]["ms_per_call"], d["decode"]["roofline"]["frac"])
PY
