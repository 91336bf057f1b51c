R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s; mkdir -p $O; cd $R
PROF_STEPS=6 bash tools/prof_step.sh > $O/prof.log 2>&1
cp gpurun_out/step_table.txt $O/step_table.txt; cp gpurun_out/step_table.json $O/step_table.json; cp gpurun_out/step_stats.csv $O/kernel_stats_short.csv
head -16 $O/step_table.txt | cut -c1-150; tail -10 $O/step_table.txt
