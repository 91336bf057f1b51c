# the per-head dR contraction: 4 k slices / plain walk (tri_split 0) against 8 slices / heavy tile rows first (1) and the walk alone (2)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R
python tools/exp/exp_dr_split.py > $O/dr_split_ab.txt 2>&1; cat $O/dr_split_ab.txt
python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "structural or split_k" 2>&1 | tail -5 | tee $O/tests_a.log
for k in 0 1 0 1; do DB1_TRI_SPLIT=$k python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-mixture --no-ga16 2>$O/bench_$k.err | tee -a $O/bench_ab.jsonl | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('tri_split=$k', r['value'], r['ms_per_step'], r.get('pct_mfma_peak_step'))"; done
