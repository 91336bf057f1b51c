cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 120 python tools/exp/dbg_chain.py 2>&1 | grep -v "worker wave" | head -3
timeout 300 python -m pytest tests/test_decode_gpu.py -x -q 2>&1 | tail -3
timeout 200 python tools/exp/dbg_chain_inmodel.py 10 2>&1 | grep -v "worker wave\|latest B0" | tail -4
timeout 300 python tools/bench_decode.py 2>&1 | grep "ring q=  1"
