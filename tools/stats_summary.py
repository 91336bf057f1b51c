"""Shorten a rocprofv3 kernel_stats.csv to: kernel, calls, total ms, average us, share."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print("kernel,calls,total_ms,avg_us,pct")
for r in rows[:45]:
    name = re.sub(r"^void ", "", r["Name"]).split("(")[0][:90]
    print(f'{name},{r["Calls"]},{float(r["TotalDurationNs"]) / 1e6:.2f},{float(r["AverageNs"]) / 1e3:.1f},{r["Percentage"]}')
