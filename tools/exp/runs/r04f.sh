cd $GRAFT_REPO_ROOT
echo "== gemm 4 default"; timeout 300 python tools/bench_kernels.py gemm 4 2>&1 | grep -v "^/opt" 
echo "== gemm 4 halfwave 16"; DB1_GEMM_HALFWAVE=16 timeout 300 python tools/bench_kernels.py gemm 4 2>&1 | grep -v "^/opt"
echo "== gemm 8 default"; timeout 300 python tools/bench_kernels.py gemm 8 2>&1 | grep -v "^/opt" 
