#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel_stats.csv as a compact table (top N kernels)."""
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/**/*_kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = list(csv.DictReader(open(files[0])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in rows[:top]:
    name = r["Name"].split("(")[0][-60:]
    print(f"{name:60s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={float(r['Percentage']):6.2f}")
