#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: calls, total ms, average us per kernel (optionally only names matching argv[2])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    if pat in r["Name"]:
        n = r["Name"].replace("(anonymous namespace)::", "")[:80]
        print(f"{n:80s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us")
