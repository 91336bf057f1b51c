cd $GRAFT_REPO_ROOT
for v in 1; do
DB1_PER_BUCKET_NORM=$v DB1_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --batch 16 --steps 3 --warmup 1 2> gpurun_out/r04t_gloo_$v.err </dev/null | grep "^{" > gpurun_out/r04t_gloo_$v.json
python - <<PY
import json
d=json.load(open("gpurun_out/r04t_gloo_$v.json"))
print("per-bucket norm $v:", d["value"], d["ms_per_step"], d.get("data_parallel"), "final loss", d.get("final_loss"))
PY
tail -2 gpurun_out/r04t_gloo_$v.err
done
