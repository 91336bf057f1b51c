R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_geometry_vision_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee $O/tests_a.log
for i in 0 1; do python bench.py --workload rl --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-mixture --no-ga16 2>$O/bench_rl_$i.err | tee $O/bench_rl_$i.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('rl', r['value'], r['ms_per_step'], r.get('pct_mfma_peak_step'), r.get('box',{}).get('mfma_random_tf'))"; done
