# round 6, third GPU call: the whole GPU suite again (the second call stopped at a test-collection error), then a kernel table of the RL step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -15 $O/tests.log
BENCH_ARGS="--workload rl" PROF_STEPS=4 bash tools/prof_step.sh > $O/prof_rl.log 2>&1
cp gpurun_out/step_table.txt $O/rl_step_table.txt; cp gpurun_out/step_table.json $O/rl_step_table.json
head -70 $O/rl_step_table.txt
