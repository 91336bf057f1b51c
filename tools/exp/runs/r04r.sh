cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python tools/exp/exp_clocks.py > gpurun_out/r04_clock_power.txt 2>&1
cat gpurun_out/r04_clock_power.txt | tail -15
