# Round evidence on the GPU box: default bench line, kernel-trace stats, PMC traffic.  Outputs under gpurun_out/evidence/.
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/evidence
mkdir -p $E
cd $R
timeout 900 python bench.py > $E/bench_default.json 2> $E/bench_default.err </dev/null
bash tools/prof_step.sh; cp gpurun_out/step_stats.csv $E/kernel_stats_short.csv; cp $(ls gpurun_out/prof_step/*kernel_stats.csv | head -1) $E/kernel_stats.csv
timeout 1500 python tools/pmc_traffic.py $E/traffic > $E/traffic.log 2>&1 </dev/null
cp $E/traffic/hbm_traffic_pmc.json $E/ 2>/dev/null
