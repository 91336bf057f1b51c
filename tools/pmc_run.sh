# usage: bash tools/pmc_run.sh <kernel-regex> "<bench args>" "<counters pass 1>" "<counters pass 2>" ...
# One counter group per pass, no trace domains; summaries land in gpurun_out/pmc_<i>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pat=$1; shift
args=$1; shift
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_$i
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$i -o a -- python $R/tools/bench_kernels.py $args > $R/gpurun_out/pmc_$i.log 2>&1 </dev/null
  (cd $R; f=$(ls gpurun_out/pmc_$i/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f "$pat" > gpurun_out/pmc_$i.txt 2>&1)
done
