# PMC passes over the flash forward kernels (compiled loop and hand-scheduled loop), B = 64; one counter group per pass, no trace domains
#   bash tools/exp/pmc_fwd2.sh <tag> "<counters pass 1>" "<counters pass 2>" ...     (FW2_HALF=1: one wave per SIMD)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcf_${tag}_$i
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcf_${tag}_$i -o a -- python $R/tools/exp/check_fwd2.py time > $R/gpurun_out/pmcf_${tag}_$i.log 2>&1 </dev/null
  (cd $R; f=$(ls gpurun_out/pmcf_${tag}_$i/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f "relattn_flash_fwd" > gpurun_out/pmcf_${tag}_$i.txt 2>&1; rm -rf gpurun_out/pmcf_${tag}_$i)
  cat $R/gpurun_out/pmcf_${tag}_$i.txt
done
