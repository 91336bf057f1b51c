cd $GRAFT_REPO_ROOT
python -m pytest tests/test_geglu_epilogue_gpu.py tests/test_full_depth_gpu.py -q > gpurun_out/r04c_tests.log 2>&1; tail -15 gpurun_out/r04c_tests.log
cat gpurun_out/full_depth_teacher_forced.json | tr -d '\n ' | head -c 3000; echo
