cd $GRAFT_REPO_ROOT
PROF_STEPS=4 BENCH_ARGS="--workload rl" bash tools/prof_step.sh > /dev/null 2>&1
cp gpurun_out/step_table.txt gpurun_out/r04u_rl_table.txt
head -64 gpurun_out/r04u_rl_table.txt
