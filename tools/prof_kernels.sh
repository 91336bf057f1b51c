# kernel-trace stats of a tools/bench_kernels.py run:  bash tools/prof_kernels.sh flash 64   -> prints the top kernels by total time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_k
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o s -- python $R/tools/bench_kernels.py "$@" > /tmp/prof_k.log 2>&1 </dev/null
grep -v "^W2\|^\[roc\|^E2" /tmp/prof_k.log | tail -12
cd $R
f=$(ls /tmp/prof_k/*kernel_stats.csv | head -1)
python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):5d} {float(r["AverageNs"]) / 1e3:10.1f} us')
P
