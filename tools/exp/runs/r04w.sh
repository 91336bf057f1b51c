cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_decode_gpu.py -x -q -k chain 2>&1 | tail -1
for i in 1 2; do timeout 300 python tools/bench_decode.py 2>&1 | grep "ring q=  1"; done
