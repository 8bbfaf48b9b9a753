"""Sweep of the rows per K split of the two weight-gradient GEMM launches (SDF group / rest group): wall time per step at the
steady-state work lists, all settings interleaved in one process.  usage: dw_split_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse
import numpy as np
import torch
import bench
args = argparse.Namespace(batch=512)
dev = torch.device("cuda:0")
# (sdf, background network, heads): the three launches of the default schedule
settings = [(6144, 4096, 4096), (6144, 4608, 4096), (6144, 5120, 4096), (6144, 5504, 4096), (6144, 4608, 2048), (6144, 4608, 2752), (6144, 4608, 1600),
            (6144, 4608, 6144), (6144, 4096, 2048), (5632, 4608, 4096), (6656, 4608, 4096)]
legs = {}
for sdf, nerf, heads in settings:
    os.environ["VDN_DW_SPLIT_PTS_SDF"], os.environ["VDN_DW_SPLIT_PTS_NERF"], os.environ["VDN_DW_SPLIT_PTS_HEADS"] = str(sdf), str(nerf), str(heads)
    rest = (sdf, nerf, heads)
    sdf = rest[0]
    leg = bench.Leg(args, dev, 1, 0, "bf16", False, 48)
    for i in range(650):
        leg.step(i)
    legs[rest] = leg
torch.cuda.synchronize()
K, R = 30, 5
res = {k: [] for k in legs}
for r in range(R):
    for k, leg in legs.items():
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(K):
            leg.step(650 + r * K + i)
        torch.cuda.synchronize()
        res[k].append((time.time() - t0) / K * 1e6)
for k in legs:
    eng = legs[k].trainer.engine
    print("sdf %5d nerf %5d heads %5d : wall %.1f us/step (min %.1f)  WGs sdf %d nerf %d heads %d" % (k[0], k[1], k[2], float(np.median(res[k])), min(res[k]),
          eng.dw_groups["sdf"][2], eng.dw_groups["nerf"][2], eng.dw_groups["heads"][2]))
