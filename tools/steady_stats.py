"""Per-kernel statistics of the steady-state tail of a `rocprofv3 --kernel-trace --output-format csv` run: only dispatches that
start in the last FRACTION of the traced time are kept (the bench's work lists shrink over its first ~600 steps; the timed regions
that decide `value` lie behind that), so that the table reproduces the bench line without conversion.
usage: steady_stats.py <kernel_trace.csv> [fraction=0.3] [steps_in_window: divide counts by it]   -> CSV on stdout"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
cut = t1 - frac * (t1 - t0)
keep = [r for r in rows if r[1] >= cut]
by = defaultdict(list)
for name, s, e in keep:
    by[name].append((e - s) / 1e3)
# steps in the window = launches of a kernel that runs exactly once per step
steps = len(by.get("vdn::composite_bwd_kernel(VdnCompositeBwdArgs)", [])) or 1
busy = sum(sum(v) for v in by.values())
print("# window: last %.0f %% of the trace = %.1f ms, %d training steps, kernel time %.1f us/step (sum over kernels)" % (
    100 * frac, (t1 - cut) / 1e6, steps, busy / steps))
print("Name,Calls,CallsPerStep,AverageUs,MinUs,MaxUs,UsPerStep,Percentage")
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%.2f,%.1f,%.1f,%.1f,%.1f,%.2f' % (name, len(v), len(v) / steps, sum(v) / len(v), min(v), max(v), sum(v) / steps, 100.0 * sum(v) / busy))
