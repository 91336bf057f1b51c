# round 6: smoke(), the 20-step trajectory with the oracle at every step (record incl. the window-mode GA legs), a 2-rank gloo window run of bench.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
DB1_TRAJ_ORACLE_STEPS=20 DB1_TRAJ_RECORD=$O/r06_trajectory.json timeout 2400 python -m pytest tests/test_trajectory_gpu.py -q -m gpu > $O/traj.log 2>&1; tail -4 $O/traj.log
DB1_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --batch 4 --ga 4 --defer-backward --layers 4 --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-box 2> $O/bench_gloo_window.err </dev/null | grep "^{" > $O/bench_gloo_window.json
python -c "
import json;d=json.loads(open('$O/bench_gloo_window.json').readline());print('gloo 2 ranks window', d['value'], d['n_gpus'], d['config']['weight_gradients'], (d.get('rl') or {}).get('tokens_per_s'), (d.get('mixture') or {}).get('tokens_per_s'))"
tail -3 $O/bench_gloo_window.err
