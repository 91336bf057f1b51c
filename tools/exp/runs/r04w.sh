cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 200 python tools/exp/dbg_chain_inmodel.py 10 2>&1 | grep -v "worker wave" | tail -4
timeout 300 python tools/bench_decode.py 2>&1 | grep "ring q=  1"
