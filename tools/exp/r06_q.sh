# the window form of the per-head dR (two batch levels, 1024 tiles of 256 x 128) on the 4-wave kernel instead of the 3-stage tile kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R
python tools/exp/exp_dr_split.py > $O/dr_window_w4n.txt 2>&1; cat $O/dr_window_w4n.txt
python -m pytest tests/test_kernels_gpu.py tests/test_w4n_gpu.py tests/test_defer_backward_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.log
for i in 0 1; do python bench.py --batch 4 --ga 16 --graph --defer-backward --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-mixture --no-ga16 --no-box 2>$O/bench_win_$i.err | tee $O/bench_win_$i.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('4 x GA16 window', r['value'], r['ms_per_step'], r.get('pct_mfma_peak_step'))"; done
