# the 64-sequence step replayed as a hipGraph (GraphedTrainStep) against the eager loop, same box, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06r; mkdir -p $O; cd $R
for i in 0 1; do
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-mixture --no-ga16 --no-box --no-kernel-timing 2>$O/eager_$i.err | tee $O/eager_$i.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('eager  ', r['value'], r['ms_per_step'], r.get('pct_mfma_peak_step'), r.get('peak_hbm_gib'))"
python bench.py --graph --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-mixture --no-ga16 --no-box 2>$O/graph_$i.err | tee $O/graph_$i.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('graphed', r['value'], r['ms_per_step'], r.get('pct_mfma_peak_step'), r.get('peak_hbm_gib'))"
done
tail -3 $O/graph_0.err
