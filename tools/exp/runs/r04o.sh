cd $GRAFT_REPO_ROOT
python -m pytest tests/test_w4n_gpu.py -x -q 2>&1 | tail -12
echo "== gemm 4: w4n before split-K (default)"; timeout 300 python tools/bench_kernels.py gemm 4 2>&1 | grep -v "^/opt" | head -12
echo "== gemm 4: w4n off"; DB1_W4N=0 timeout 300 python tools/bench_kernels.py gemm 4 2>&1 | grep -v "^/opt" | head -12
echo "== gemm 4: w4n after split-K"; DB1_W4N=2 timeout 300 python tools/bench_kernels.py gemm 4 2>&1 | grep -v "^/opt" | head -12
