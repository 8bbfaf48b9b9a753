"""A/B of the Trainer's two-stream schedule (overlap on / off) in ONE process, interleaved regions: wall time per step and
host issue time per step.  usage: overlap_probe.py [steps_per_region=40] [rounds=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse
import numpy as np
import torch
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
args = argparse.Namespace(batch=512)
dev = torch.device("cuda:0")
legs = {}
for name, ov in (("overlap", "1"), ("inorder", "0")):
    os.environ["VDN_OVERLAP"] = ov
    legs[name] = bench.Leg(args, dev, 1, 0, "bf16", False, 64)
for leg in legs.values():
    for i in range(700):                     # past the first ~600 steps: the work lists have reached their steady size
        leg.step(i)
torch.cuda.synchronize()
res = {k: [] for k in legs}
issue = {k: [] for k in legs}
for r in range(R):
    for name, leg in legs.items():
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(K):
            leg.step(700 + r * K + i)
        t1 = time.time()
        torch.cuda.synchronize()
        res[name].append((time.time() - t0) / K * 1e6)
        issue[name].append((t1 - t0) / K * 1e6)
for name in legs:
    eng = legs[name].trainer.engine
    print("%-8s wall %.1f us/step (min %.1f)   host issue %.1f us/step   rows fg %d bg %d" % (
        name, float(np.median(res[name])), min(res[name]), float(np.median(issue[name])), int(eng.w["fg_active"][1].item()), int(eng.w["bg_active"][1].item())))
