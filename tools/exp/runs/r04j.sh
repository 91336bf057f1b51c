cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 8 --warmup 3 > gpurun_out/r04j_t$i.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-decode --no-mixture --steps 8 --warmup 3 --no-kernel-timing > gpurun_out/r04j_n$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("t1","n1","t2","n2"):
    d=json.loads(open(f"gpurun_out/r04j_{n}.json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"])
PY
