#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 *kernel_stats.csv: name, calls, average microseconds.  usage: top_kernels.py file.csv [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print("%-78s %6s %9.1f us" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3))
