# round 6: packed fp32 VALU in the hand-scheduled flash forward (FW3_PACK): parity, then a same-box A/B (the library is rebuilt on the box for the other arm)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_geometry_gpu.py -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
for rnd in 1 2; do
  timeout 300 python tools/bench_kernels.py flash 64 > $O/flash_pack1_$rnd.txt 2>&1; grep -i "fwd\|forward" $O/flash_pack1_$rnd.txt | head -4
  FW3_PACK=0 python tools/gen_flash_fwd3.py > /dev/null && python -m bdm_db1_amd.build > /dev/null 2>&1
  timeout 300 python tools/bench_kernels.py flash 64 > $O/flash_pack0_$rnd.txt 2>&1; grep -i "fwd\|forward" $O/flash_pack0_$rnd.txt | head -4
  python tools/gen_flash_fwd3.py > /dev/null && python -m bdm_db1_amd.build > /dev/null 2>&1
done
timeout 600 python bench.py --no-cpu-baseline --no-decode --no-mixture --no-ga16 --steps 8 --warmup 3 2> $O/bench.err | grep "^{" > $O/bench_pack1.json
python -c "
import json;d=json.loads(open('$O/bench_pack1.json').readline());print('pack1', d['value'], d['ms_per_step'], d['pct_mfma_peak_step'], d['kernels']['flash_fwd'], d['box'])"
