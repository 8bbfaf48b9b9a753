#!/usr/bin/env python3
"""profiles/<round>_traffic_*.json (the file bench.py's roofline.traffic reads) from a summary written by
tools/summarise_counters.py:  tools/traffic_json.py <summary.json> <kernel-name substring> "<note>" > profiles/...json"""
import json, sys
summ, key, note = json.load(open(sys.argv[1])), sys.argv[2], sys.argv[3]
ks = [k for k in summ if key in k and "hbm_bytes_per_dispatch" in summ[k]]
assert len(ks) >= 1, (key, list(summ))
k = max(ks, key=lambda n: summ[n]["hbm_bytes_per_dispatch"]["total"])
h = summ[k]["hbm_bytes_per_dispatch"]
out = {"kernel": k, "fetch_bytes_x2": h["fetch_x2"], "write_bytes": h["write"], "hbm_bytes_per_launch": h["total"],
       "note": note + " FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM); separate --pmc passes (tools/collect_counters.sh)."}
if "kernel_trace_us" in summ[k]:
    out["kernel_trace_us"] = summ[k]["kernel_trace_us"]
print(json.dumps(out, indent=1))
