cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_dec
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -o s -- python $R/tools/bench_decode.py 22 10 > $R/gpurun_out/prof_dec.log 2>&1 </dev/null
cd $R; python tools/stats_summary.py $(ls gpurun_out/prof_dec/*kernel_stats.csv | head -1) > gpurun_out/dec_stats.csv
