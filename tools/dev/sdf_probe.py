"""The fused SDF kernel alone (HIP events, back to back): inference launch (65 536 points), training launch over all rows, the
sdf-only pass (32 768 points), the one-launch shading kernel. For A/B runs of library variants (VDN_LIB).  usage: sdf_probe.py [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse
import numpy as np
import torch
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
os.environ.setdefault("VDN_SIDE_STREAM", "0")
os.environ.setdefault("VDN_OVERLAP", "0")
leg = bench.Leg(argparse.Namespace(batch=512), torch.device("cuda:0"), 1, 0, "bf16", False, 8)
for i in range(40):
    leg.step(i)
torch.cuda.synchronize()
eng, rend = leg.trainer.engine, leg.rend
o, d = leg.batches[0][0], leg.batches[0][1]
bgc = torch.ones(3, device="cuda:0")

def med(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), min(ts)
with torch.no_grad():
    inf = med(lambda: rend.sdf_network._run(1, rays=(o, d, eng.w["mid_z"])))
    z64 = eng.w["mid_z"][:, :64]
    m0 = med(lambda: rend.sdf_network._run(0, rays=(o, d, z64)))
    sh = med(lambda: rend._shade(o, d, eng.w["dists"], eng.w["mid_z"], None, bgc, 0.5))
fgc, eng._fg_compact = eng._fg_compact, False
tr = med(lambda: eng._sdf_forward(o, d))
eng._fg_compact = fgc
F1 = bench.F_SDF + bench.F_GRAD
print("%-8s inference %.1f us (min %.1f) = %.3f of peak | training launch 65 536 rows %.1f (min %.1f) = %.3f | sdf-only 32 768 pts %.1f (min %.1f) | one-launch shading %.1f (min %.1f)"
      % (tag, inf[0], inf[1], F1 * 65536 / (inf[0] * 1e-6) / 2.5e15, tr[0], tr[1], F1 * 65536 / (tr[0] * 1e-6) / 2.5e15, m0[0], m0[1], sh[0], sh[1]))
