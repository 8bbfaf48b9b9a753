"""One steady-state training step as a timeline, from a rocprofv3 kernel trace (dev): start offset, duration, queue, kernel.
usage: python tools/dev/timeline.py <kernel_trace.csv> [step index from the end, default 3; negative: from the start]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the coarse_z kernel of the sampler
starts = [i for i, r in enumerate(rows) if "coarse_z_kernel" in r["Kernel_Name"]]
i0, i1 = (starts[-back - 1], starts[-back]) if back > 0 else (starts[-back], starts[-back + 1])
t0 = int(rows[i0]["Start_Timestamp"])
# kernels of the previous step may still run on the side stream: include everything that overlaps [t0, t1)
t1 = int(rows[i1]["Start_Timestamp"])
print("step wall (sampler start to next sampler start): %.1f us" % ((t1 - t0) / 1e3))
qs = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= t0 or s >= t1:
        continue
    q = qs.setdefault(r["Queue_Id"], len(qs))
    name = r["Kernel_Name"].replace("void ", "").replace("vdn::", "")
    print("%8.1f %7.1f  q%d %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, "        " * q, name[:70]))
