cd $GRAFT_REPO_ROOT
for b in 64 16; do
echo "== B=$b w4n=3"; python tools/bench_kernels.py small_out $b 2>&1 | grep -v "^/opt" | head -4
echo "== B=$b w4n=0"; DB1_W4N=0 python tools/bench_kernels.py small_out $b 2>&1 | grep -v "^/opt" | head -4
done
