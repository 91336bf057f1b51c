cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python tools/bench_decode.py 2>&1 | grep "ring q=  1"; done
