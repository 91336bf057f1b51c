# PMC passes over the GEMM micro-benchmark (one counter group per pass, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DB1_GEMM_TILE=${DB1_GEMM_TILE:-512}
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_g$i -o a -- python $R/tools/bench_kernels.py gemm > $R/gpurun_out/pmc_g$i.log 2>&1 </dev/null
  (cd $R; python tools/pmc_summary.py $(ls gpurun_out/pmc_g$i/*counter_collection.csv | head -1) gemm_bf16 > gpurun_out/pmc_g$i.txt 2>&1)
done
