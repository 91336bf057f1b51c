cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_decode_gpu.py -x -q 2>&1 | tail -5
