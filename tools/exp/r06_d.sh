# round 6, fourth GPU call: the patch-resident convolution (parity + A/B), the 2-rank window test, the M = 16 / 4 decode kernel table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv or patch or implicit" > $O/tests_conv.log 2>&1; tail -5 $O/tests_conv.log
timeout 900 python -m pytest tests/test_defer_backward_gpu.py tests/test_geometry_vision_gpu.py tests/test_decode_gpu.py -q -m gpu > $O/tests_b.log 2>&1; tail -8 $O/tests_b.log
timeout 600 python tools/exp/exp_conv_patch.py > $O/conv_patch_ab.txt 2>&1; cat $O/conv_patch_ab.txt
for w in rl mixture; do
 for k in 0 1; do
  DB1_CONV_PATCH=$k timeout 600 python bench.py --workload $w --no-cpu-baseline --no-decode --no-mixture --no-ga16 --no-box --steps 6 --warmup 2 2> $O/bench_${w}_$k.err | grep "^{" > $O/bench_${w}_$k.json
  python -c "
import json;d=json.loads(open('$O/bench_${w}_$k.json').readline());print('$w conv_patch=$k', d['value'], d['ms_per_step'], d['pct_mfma_peak_step'])"
 done
done
MS="16 4" bash tools/prof_decode_batched.sh > $O/prof_decb.log 2>&1
cp gpurun_out/decb_stats_16.csv gpurun_out/decb_stats_4.csv $O/ 2>/dev/null
head -22 $O/decb_stats_16.csv
