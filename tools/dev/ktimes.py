"""median duration per (kernel, grid) from a rocprofv3 kernel trace (dev)"""
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or sys.argv[2] in r["Kernel_Name"]:
        d[(r["Kernel_Name"].replace("void ", "")[:60], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v.sort()
    print("%-62s grid %8s  n %4d  median %8.1f us  min %8.1f" % (k[0], k[1], len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
