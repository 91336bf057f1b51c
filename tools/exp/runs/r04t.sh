cd $GRAFT_REPO_ROOT
python -m pytest tests/test_decode_gpu.py tests/test_kernels_gpu.py -x -q -k "decode or skinny or linear" 2>&1 | tail -4
python tools/bench_decode.py 2>&1 | grep -v "^/opt" | tail -8
echo "== rows8 off"; DB1_LINEAR_DECODE_ROWS8=0 python tools/bench_decode.py 2>&1 | grep -v "^/opt" | tail -8
