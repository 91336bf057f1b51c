cd $GRAFT_REPO_ROOT
PROF_STEPS=3 BENCH_ARGS="--batch 4 --ga 16" bash tools/prof_step.sh > /dev/null 2>&1
cp gpurun_out/step_table.txt gpurun_out/r04h_b4_table.txt; cp gpurun_out/step_table.json gpurun_out/r04h_b4_table.json
head -60 gpurun_out/r04h_b4_table.txt
