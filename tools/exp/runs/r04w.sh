cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_decode_gpu.py -x -q 2>&1 | tail -5
timeout 300 python tools/bench_decode.py 2>&1 | grep "ring q=  1"
DB1_DECODE_CHAIN=0 timeout 300 python tools/bench_decode.py 2>&1 | grep "ring q=  1"
