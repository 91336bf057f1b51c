"""Aggregate a rocprofv3 counter_collection CSV: per (short kernel name, counter) mean value over dispatches."""
import csv, re, sys, collections
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "gemm|relattn|adam")
acc = collections.defaultdict(list)
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"]
        if not re.search(pat, name) or "at::native" in name:
            continue
        short = re.sub(r"^void ", "", name).split("(")[0][:70]
        acc[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:70s} {c:32s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
