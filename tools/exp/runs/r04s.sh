# HBM-side bytes of the decode kernels (eager 1-token calls: graph replays under counter collection abort the queue), separate PMC passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_dec_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_dec_$c -o a -- python $R/tools/exp/prof_decode_ring.py 12 > $R/gpurun_out/pmc_dec_$c.log 2>&1 </dev/null
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for c, mul in (("FETCH_SIZE", 2 * 1024), ("WRITE_SIZE", 1024)):
    f = glob.glob(f"gpurun_out/pmc_dec_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0], newline="")):
        if row["Counter_Name"] == c and ("decode" in row["Kernel_Name"] or "skinny" in row["Kernel_Name"]):
            acc[row["Kernel_Name"].replace("void ", "").split("(")[0][:60]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {"launches": len(v)})[c.lower() + "_bytes_per_launch"] = round(mul * sum(v) / len(v))
res = {"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/exp/prof_decode_ring.py 12 (eager 1-token calls, DB1-1.3B, memory full); bytes = 2 * FETCH_SIZE * 1024, WRITE_SIZE * 1024 (gfx950 correction of tools/pmc_traffic.py)",
       "algorithmic_bytes": {"decode_chain_kernel": 2 * (2048 * 2048 + 8192 * 2048 + 2048 * 4096 + 6144 * 2048), "relattn_decode_ring_kernel": 2 * 1025 * 2 * 2048 + 2 * 1025 * 2048},
       "kernels": out}
json.dump(res, open("gpurun_out/r04_decode_hbm_traffic_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:1800])
PY
