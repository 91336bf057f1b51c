R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o s -- python $R/tools/exp/prof_ring.py ${1:-1} > /tmp/pr.log 2>&1 </dev/null
tail -3 /tmp/pr.log | cut -c1-200
python $R/tools/stats_summary.py $(find /tmp/pr -name '*kernel_stats.csv' | head -1) | head -30
python - $(find /tmp/pr -name '*kernel_trace.csv' | head -1) <<'P'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "skinny" in n or "decode" in n or "ln_fwd" in n:
        d[(n[:60], r["Grid_Size_X"], r.get("Grid_Size_Y"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v.sort(); print(k, len(v), "median %.1f min %.1f" % (v[len(v)//2], v[0]))
P
python $R/tools/exp/trace_tail.py $(find /tmp/pr -name '*kernel_trace.csv' | head -1) ${TAILN:-135} > $R/gpurun_out/ring_trace_tail.txt
