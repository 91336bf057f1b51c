# kernel trace of eager 1-token calls over the K / V ring (the kernels of the graphed call, one by one) + smoke()
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_dec
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -o d -- python $R/tools/exp/prof_decode_ring.py 30 > $R/gpurun_out/prof_dec.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_dec/**/d_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = ["# rocprofv3 --kernel-trace --stats of tools/exp/prof_decode_ring.py 30: thirty eager 1-token calls over the K / V ring, DB1-1.3B, memory full (24 layers per call)",
       "# (the first rows are the call's own kernels; fills / copies / casts below them belong to building the model)",
       "kernel,calls,avg_us,total_ms,percent"]
for r in rows[:16]:
    out.append(f"{r['Name'][:90]},{r['Calls']},{float(r['AverageNs'])/1e3:.2f},{float(r['TotalDurationNs'])/1e6:.3f},{r['Percentage']}")
open("gpurun_out/r04_decode_kernel_stats.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
PY
