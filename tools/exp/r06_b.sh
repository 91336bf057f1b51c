# round 6, second GPU call: the whole GPU suite after the clean-up, then the default bench line (box calibration, the ga16 leg in window mode)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -15 $O/tests.log
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"
tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06b/bench_default.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","pct_mfma_peak_step","peak_hbm_gib","box","value_normalised")})
print("ga16", d.get("ga16"))
print("rl", (d.get("rl") or {}).get("tokens_per_s"), "mixture", (d.get("mixture") or {}).get("tokens_per_s"))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","traffic")})
print("kernels", {k:(v["avg_us"],v["frac"]) for k,v in d.get("kernels",{}).items()})
print("decode", json.dumps(d.get("decode"))[:600])
PY
