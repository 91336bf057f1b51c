# kernel-trace stats of the default bench; writes gpurun_out/step_stats.csv (top kernels by total time)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step -o s -- python $R/bench.py --steps 4 --warmup 2 ${BENCH_ARGS} > $R/gpurun_out/prof_step.log 2>&1 </dev/null
cd $R
f=$(ls gpurun_out/prof_step/*kernel_stats.csv | head -1)
python tools/stats_summary.py "$f" > gpurun_out/step_stats.csv
