#!/bin/bash
# HBM traffic of the dominant kernel (fused SDF forward, 65 536 points) from the L2 memory-side counters,
# in two separate --pmc passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 TCC
# slots, WRITE_SIZE 2). Usage (on the GPU box, repo root): bash tools/collect_traffic.sh [bf16|fp32] [sdf1|sdf1t]
PREC=${1:-bf16}
WHICH=${2:-sdf1}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/traffic_${PREC}_${WHICH}_$C -- python tools/kernel_loop.py $WHICH 65536 $PREC 6 > /dev/null 2>&1
done
python - "$PREC" "$WHICH" <<'PY'
import csv, glob, json, sys
prec, which = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/traffic_%s_%s_%s/*/*counter_collection.csv" % (prec, which, c))[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "sdf_fwd_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    res[c] = sum(vals[1:]) / max(len(vals) - 1, 1)          # skip the first (cold) launch
# counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B => x2 (guide, section HBM)
out = {"kernel": "sdf_fwd_kernel<%s,1> 65536 points" % prec, "fetch_kib_raw": res["FETCH_SIZE"], "write_kib_raw": res["WRITE_SIZE"],
       "hbm_bytes_per_launch": (2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0,
       "note": "FETCH_SIZE doubled per the gfx950 correction; " + ("training-mode launch (saves H, V, PE planes; fp32 also S)" if which == "sdf1t" else "inference-mode launch (S workspace only)")}
print(json.dumps(out))
open("gpurun_out/traffic_%s%s.json" % (prec, "_train" if which == "sdf1t" else ""), "w").write(json.dumps(out))
PY
