"""Per-kernel statistics of the steady-state tail of a `rocprofv3 --kernel-trace --output-format csv` run: only dispatches that
start in the last FRACTION of the traced time are kept (the bench's work lists shrink over its first ~600 steps; the timed regions
that decide `value` lie behind that), so that the table reproduces the bench line without conversion.
usage: steady_stats.py <kernel_trace.csv> [fraction=0.3] [bench.json of the traced run]   -> CSV on stdout
With the traced run's bench line the header also carries the mean work-list rows per step (work_list_rows_mean), so that a
kernel's roofline fraction can be recomputed from the table alone: FLOP per row x rows / AverageUs / peak."""
import json
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
steps = len(by.get("vdn::coarse_z_kernel(VdnCoarseArgs)", [])) or 1
busy = sum(sum(v) for v in by.values())
print("# window: last %.0f %% of the trace = %.1f ms, %d training steps, kernel time %.1f us/step (sum over kernels)" % (
    100 * frac, (t1 - cut) / 1e6, steps, busy / steps))
if len(sys.argv) > 3:
    try:
        line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
        j = json.loads(line)
        wl = j["work_list_rows_mean"]
        print("# mean work-list rows per step (steady state, %d steps): foreground %.0f of %d (sdf_fwd2 / rendernet / sdf_bwd kernels), "
              "background %.0f of %d (nerf kernels); bench line of this run: %.4f ms/step" % (
                  wl["over_steps"], wl["foreground"], j["config"]["foreground_points_total"], wl["background"],
                  j["config"]["background_points_total"], j["ms_per_step"]))
        fg = wl["foreground"]
        k = [v for n, v in by.items() if "sdf_fwd2_kernel<1, true" in n]
        k3 = [v for n, v in by.items() if "sdf_fwd2_kernel<3, true" in n]        # VDN_TRAIN_COLOR_FUSED=1: + the colour head, 542 720 FLOP/row more
        if k3 and not k:
            us3 = sum(k3[0]) / steps
            print("# fused SDF + colour forward of the step: 2 509 824 FLOP/row x %.0f rows / %.1f us sdf_fwd2_kernel<3,true> = %.1f TFLOP/s = %.3f of the "
                  "2.5 PFLOP/s bf16 MFMA peak" % (fg, us3, 2509824.0 * fg / us3 / 1e6, 2509824.0 * fg / us3 / 1e6 / 2500.0))
        tail = [v for n, v in by.items() if "sdf_fwd1_split_kernel<true>" in n]
        if k:
            us = sum(k[0]) / steps
            us_tail = sum(tail[0]) / steps if tail else 0.0
            print("# fused SDF forward of the step: 1 967 104 FLOP/row x %.0f rows / (%.1f us sdf_fwd2_kernel<1,true>%s) = %.1f TFLOP/s = %.3f of the "
                  "2.5 PFLOP/s bf16 MFMA peak" % (fg, us, " + %.1f us sdf_fwd1_split_kernel<true>, which evaluates the rows behind the first "
                                                  "32 768" % us_tail if tail else "", 1967104.0 * fg / (us + us_tail) / 1e6,
                                                  1967104.0 * fg / (us + us_tail) / 1e6 / 2500.0))
    except Exception as e:          # the table itself does not depend on it
        print("# (no bench line: %s)" % e)
print("Name,Calls,CallsPerStep,AverageUs,MinUs,MaxUs,UsPerStep,Percentage")
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%.2f,%.1f,%.1f,%.1f,%.1f,%.2f' % (name, len(v), len(v) / steps, sum(v) / len(v), min(v), max(v), sum(v) / steps, 100.0 * sum(v) / busy))
