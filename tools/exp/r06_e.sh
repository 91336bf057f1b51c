R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv or patch or implicit" > $O/tests_conv.log 2>&1; tail -12 $O/tests_conv.log
timeout 600 python tools/exp/exp_conv_patch.py > $O/conv_patch_ab.txt 2>&1; cat $O/conv_patch_ab.txt
timeout 900 python -m pytest tests/test_geometry_vision_gpu.py tests/test_model_gpu.py -q -m gpu > $O/tests_b.log 2>&1; tail -5 $O/tests_b.log
for k in 0 1 2; do
  DB1_CONV_PATCH=$k timeout 600 python bench.py --workload rl --no-cpu-baseline --no-decode --no-mixture --no-ga16 --no-box --steps 6 --warmup 2 2> $O/bench_rl_$k.err | grep "^{" > $O/bench_rl_$k.json
  python -c "
import json;d=json.loads(open('$O/bench_rl_$k.json').readline());print('rl conv_patch=$k', d['value'], d['ms_per_step'], d['pct_mfma_peak_step'])"
done
