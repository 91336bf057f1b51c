# the micro-batch sweep at a fixed 64 sequences per GPU per optimizer step (the reference's own geometry is micro-batch 4 x GA 16).
#   TAG=r03 bash tools/sweep_microbatch.sh      -> gpurun_out/sweep_$TAG/${TAG}_bench_b{4,8,16,32,64}.json
R=${GRAFT_REPO_ROOT:-.}
TAG=${TAG:-r03}
E=$R/gpurun_out/sweep_$TAG
mkdir -p $E
cd $R
for b in ${SWEEP_BATCHES:-4 8 16 32 64}; do
  ga=$((64 / b))
  timeout 600 python bench.py --batch $b --ga $ga --steps 4 --warmup 2 --no-cpu-baseline --no-decode ${SWEEP_ARGS} > $E/${TAG}_bench_b${b}.json 2> $E/bench_b${b}.err </dev/null
  python - <<P
import json
try:
    d = json.load(open("$E/${TAG}_bench_b${b}.json"))
    print("B=$b GA=$ga", d["value"], "tok/s", d["ms_per_step"], "ms/step", d["pct_mfma_peak_step"], "% peak", "peak HBM", d["peak_hbm_gib"], "GiB")
except Exception as e:
    print("B=$b failed", e)
P
done
