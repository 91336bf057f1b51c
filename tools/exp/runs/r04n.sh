cd $GRAFT_REPO_ROOT
PROF_STEPS=3 BENCH_ARGS="--batch 4 --ga 16 --graph" bash tools/prof_step.sh > /dev/null 2>&1
cp gpurun_out/step_table.txt gpurun_out/r04n_b4g_table.txt; cp gpurun_out/step_table.json gpurun_out/r04n_b4g_table.json
head -50 gpurun_out/r04n_b4g_table.txt; tail -5 gpurun_out/prof_step.log | cut -c1-400
