R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06o; mkdir -p $O; cd $R
python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
python bench.py 2>$O/bench_default.err | tee $O/bench_default.json | python -c "
import sys,json; r=json.loads(sys.stdin.read())
print('text', r['value'], r['ms_per_step'], r.get('pct_mfma_peak_step'), 'roofline', r['roofline']['frac'], 'peak GiB', r.get('peak_hbm_gib'))
for k in ('rl','mixture','ga16'):
    o=r.get(k) or {}
    print(k, o.get('tokens_per_s') or o.get('value'), o.get('pct_mfma_peak_step'))
print('decode', (r.get('decode') or {}).get('ms_per_call_graphed') or r.get('decode'))
print('cpu', r.get('cpu_baseline'))
print('box', r.get('box'))
"
