# round 6, first GPU call: the deferred backward (one backward per accumulation window) -- parity tests, then the 4 x GA 16 A/B on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_defer_backward_gpu.py tests/test_defer_wgrad_gpu.py tests/test_dropout.py tests/test_flash_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -25 $O/tests.log
for mode in wgrad backward wgrad backward; do
  if [ $mode = backward ]; then F="--defer-backward"; else F=""; fi
  timeout 600 python bench.py --batch 4 --ga 16 --graph $F --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-mixture --no-ga16 2> $O/bench_$mode.err | grep "^{" >> $O/bench_$mode.json
  tail -3 $O/bench_$mode.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06a/bench_*.json")):
    for l in open(f):
        d=json.loads(l); print(f, d.get("value"), d.get("ms_per_step"), d.get("pct_mfma_peak_step"), d.get("peak_hbm_gib"))
PY
