#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 *kernel_stats.csv: name, calls, average microseconds, and the total GPU time of all
kernels (divide by the number of steps for the busy time per step).  usage: top_kernels.py file.csv [n] [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print("%-78s %6s %9.1f us" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3))
total = sum(float(r["TotalDurationNs"]) for r in rows) / 1e3
print("all kernels: %.0f us" % total + (" = %.1f us per step over %s steps" % (total / int(sys.argv[3]), sys.argv[3]) if len(sys.argv) > 3 else ""))
