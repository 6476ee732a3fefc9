#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory.  python tools/kstats.py DIR [N] [filter]"""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 15
flt = sys.argv[3] if len(sys.argv) > 3 else ""
for f in sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[:1]:
    rows = [r for r in csv.DictReader(open(f)) if flt in r["Name"]]
    for r in rows[:n]:
        print("%-78s calls %5s avg %8.1f us" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3))
