"""Compact per-kernel table from a tools/summarise_counters.py JSON (development aid)."""
import json, sys
d = json.load(open(sys.argv[1]))
rows = []
for k, v in d.items():
    m, der = v["counters_mean_per_dispatch"], v.get("derived", {})
    us = (v.get("kernel_trace_us") or {}).get("mean", 0)
    hb = (v.get("hbm_bytes_per_dispatch") or {}).get("total", 0)
    rows.append((us * (v.get("kernel_trace_us") or {}).get("n", 0), k[:52], us, der.get("SQ_ACTIVE_INST_VALU/SQ_WAVE_CYCLES", 0), der.get("SQ_ACTIVE_INST_LDS/SQ_WAVE_CYCLES", 0),
                 der.get("SQ_ACTIVE_INST_VMEM/SQ_WAVE_CYCLES", 0), der.get("SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES", 0),
                 der.get("mfma_busy_share_of_wave_time (one wave per SIMD)", 0), hb / 1e6, hb / 1e3 / us if us else 0,
                 m.get("SQ_WAVES", 0), m.get("SQ_INSTS_VALU", 0) / 1e6, m.get("SQ_INSTS_MFMA", 0) / 1e6, m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
rows.sort(reverse=True)
print("%-52s %7s %5s %5s %5s %5s %5s %7s %6s %6s %7s %6s %5s" % ("kernel", "us", "valu", "lds", "vmem", "wait", "mfma", "MB", "GB/s", "waves", "Mvalu", "Mmfma", "bank"))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("%-52s %7.1f %5.2f %5.2f %5.2f %5.2f %5.2f %7.0f %6.0f %6d %7.1f %6.2f %5.2f" % r[1:])
