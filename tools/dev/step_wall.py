"""Wall time per training step at the steady-state work lists, for A/B runs of process-wide switches (environment variables
read once per process): run the same command once per setting inside ONE gpurun call.
usage: step_wall.py [tag] [steps_per_region=40] [regions=6] [config: white|wdepth] [crop|-] [bf16|fp32]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse
import numpy as np
import torch
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
R = int(sys.argv[3]) if len(sys.argv) > 3 else 6
wdepth = len(sys.argv) > 4 and sys.argv[4] == "wdepth"
crop = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != "-" else None
prec = sys.argv[6] if len(sys.argv) > 6 else "bf16"
args = argparse.Namespace(batch=512)
rc = bench.real_cameras() if os.environ.get("VDN_REAL_CAMS", "0") == "1" else None      # (the rig of a scene the reference ships: bench.py real_cameras)
leg = bench.Leg(args, torch.device("cuda:0"), 1, 0, prec, wdepth, 64, crop=crop, cams=None if rc is None else rc[0], focal=None if rc is None else rc[1])
for i in range(700 if prec == "bf16" else 120):
    leg.step(i)
torch.cuda.synchronize()
res, host = [], []
for r in range(R):
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(K):
        leg.step(700 + r * K + i)
    t1 = time.time()
    torch.cuda.synchronize()
    res.append((time.time() - t0) / K * 1e6)
    host.append((t1 - t0) / K * 1e6)
eng = leg.trainer.engine
print("%-14s wall %.1f us/step (min %.1f, max %.1f; host enqueue %.1f)  rows fg %d bg %d  loss %.5f" % (
    tag, float(np.median(res)), min(res), max(res), float(np.median(host)), int(eng.w["fg_active"][1].item()), int(eng.w["bg_active"][1].item()),
    float(leg.trainer.scalars[0].item())))
