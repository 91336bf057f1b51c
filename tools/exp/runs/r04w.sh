cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gradsync_gpu.py -x -q 2>&1 | tail -8
DB1_DEBUG_STREAM=1 timeout 300 python -m pytest tests/test_gradsync_gpu.py -x -q 2>&1 | tail -3
