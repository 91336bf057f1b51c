# rocprofv3 kernel trace of the default bench, cut to its TIMED steps by the marker kernels bench.py launches:
#   gpurun_out/step_table.json  per-kernel table {calls/step, avg us, ms/step, share} + every timed family against its roofline from the profiler's
#                               own durations (tools/prof_table.py; copy to profiles/rNN_step_table.json: bench.py's roofline.rocprof_frac reads it)
#   gpurun_out/step_stats.csv   the whole-process kernel_stats summary (top kernels by total time), as before
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_step
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step -o s -- python $R/bench.py --steps ${PROF_STEPS:-6} --warmup 2 --no-cpu-baseline --no-decode --no-mixture --no-ga16 ${BENCH_ARGS} > $R/gpurun_out/prof_step.log 2>&1 </dev/null
cd $R
f=$(ls gpurun_out/prof_step/*kernel_stats.csv | head -1)
python tools/stats_summary.py "$f" > gpurun_out/step_stats.csv
t=$(ls gpurun_out/prof_step/*kernel_trace.csv | head -1)
python tools/prof_table.py "$t" gpurun_out/prof_step.log gpurun_out/step_table.json > gpurun_out/step_table.txt 2>&1
tail -20 gpurun_out/step_table.txt
