"""Print the key fields of a bench.py JSON line read from stdin (development aid)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("rays/s %.0f  ms/step %.4f  sdf frac %.3f (%.1f us, %d pts)  inference frac %.3f (%.1f us)" % (
    d["value"], d["ms_per_step"], r.get("frac", 0), 1e3 * r.get("kernel_ms", 0), r.get("points", 0),
    (r.get("inference_launch") or {}).get("frac", 0), 1e3 * (r.get("inference_launch") or {}).get("kernel_ms", 0)))
g = d.get("roofline_dw_gemm")
if g:
    print("dw_gemm %.1f us  %.0f GB/s algorithmic (frac %.3f)  %.0f TFLOP/s" % (1e3 * g["kernel_ms"], g["achieved"], g["frac"], g.get("tflops", 0)))
