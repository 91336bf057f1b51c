# round 6: kernel table of the 4 x GA 16 optimizer step in window mode (forward graphs + ONE backward), kernel tests touched since
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "vision or add or conv or patch" > $O/tests.log 2>&1; tail -4 $O/tests.log
BENCH_ARGS="--batch 4 --ga 16 --graph --defer-backward --no-box" PROF_STEPS=3 bash tools/prof_step.sh > $O/prof.log 2>&1
cp gpurun_out/step_table.txt $O/b4_ga16_window_step_table.txt; cp gpurun_out/step_table.json $O/b4_ga16_window_step_table.json
head -60 $O/b4_ga16_window_step_table.txt
