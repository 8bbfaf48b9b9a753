#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc csv files written by tools/collect_counters.sh: per kernel, the mean of every counter
per dispatch and the derived shares the north_star asks for (MFMA-busy, VALU-active, wait shares, HBM bytes).
   usage: tools/summarise_counters.py gpurun_out/<tag> [substring of the kernel name ...] > profiles/rNN_counters_<what>.json
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; FETCH_SIZE is reported in KiB and is doubled (gfx950 tallies
128-B requests at 64 B); WRITE_SIZE in KiB."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
filters = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name") or row.get("Kernel Name")
        if filters and not any(s in k for s in filters):
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if filters and not any(s in k for s in filters):
            continue
        dur[k].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3)
out = {}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {"counters_mean_per_dispatch": m, "dispatches": {c: len(v) for c, v in cs.items()}}
    if k in dur:
        d["kernel_trace_us"] = {"mean": sum(dur[k]) / len(dur[k]), "min": min(dur[k]), "n": len(dur[k])}
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        der = {}
        for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA",
                  "SQ_ACTIVE_INST_MISC", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                der[c + "/SQ_WAVE_CYCLES"] = m[c] / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
            pass
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            # wave cycles are quad-cycles per wave; with one wave per SIMD, 4 * SQ_WAVE_CYCLES = SIMD-cycles covered by waves
            der["mfma_busy_share_of_wave_time (one wave per SIMD)"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc)
        d["derived"] = der
    if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
        d["hbm_bytes_per_dispatch"] = {"fetch_x2": 2 * 1024 * m.get("FETCH_SIZE", 0.0), "write": 1024 * m.get("WRITE_SIZE", 0.0),
                                       "total": 2 * 1024 * m.get("FETCH_SIZE", 0.0) + 1024 * m.get("WRITE_SIZE", 0.0)}
    out[k] = d
json.dump(out, sys.stdout, indent=1, sort_keys=True)
