# per-kernel picture of the batched one-token call: rocprofv3 kernel stats of tools/bench_decode_batched.py for M = 1, 4, 16 (eager calls only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for M in ${MS:-1 4 16}; do
  rm -rf $R/gpurun_out/prof_decb_$M
  DB1_DECODE_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_decb_$M -o s -- python $R/tools/bench_decode_batched.py $M 30 > $R/gpurun_out/prof_decb_$M.log 2>&1 </dev/null
  (cd $R; python tools/stats_summary.py $(ls gpurun_out/prof_decb_$M/*kernel_stats.csv | head -1) > gpurun_out/decb_stats_$M.csv; head -25 gpurun_out/decb_stats_$M.csv; tail -2 gpurun_out/prof_decb_$M.log)
done
