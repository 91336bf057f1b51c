R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06k; mkdir -p $O; cd $R
BENCH_ARGS="--workload rl" PROF_STEPS=4 bash tools/prof_step.sh > $O/prof_rl.log 2>&1
cp gpurun_out/step_table.txt $O/rl_step_table.txt; cp gpurun_out/step_table.json $O/rl_step_table.json
head -48 $O/rl_step_table.txt | cut -c1-150
