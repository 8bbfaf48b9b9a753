"""C2 chain (renderer.py:239-315 on 65 536 points) and whole render() throughput, fused launch against the separate launches.
usage: fwd_c2_probe.py"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np
import torch
import bench

dev = torch.device("cuda:0")
leg = bench.Leg(argparse.Namespace(batch=512), dev, 1, 0, "bf16", False, 32)
for i in range(300):
    leg.step(i)
torch.cuda.synchronize()
eng, rend = leg.trainer.engine, leg.rend
o, d = leg.batches[0][0], leg.batches[0][1]
bgc = torch.ones(3, device=dev)
bg = torch.ones(1, 3, device=dev)
for fused in ("0", "1", "0", "1"):
    os.environ["VDN_SHADE_FUSED"] = fused
    with torch.no_grad():
        t = bench.time_kernel_stats(lambda: rend._shade(o, d, eng.w["dists"], eng.w["mid_z"], None, bgc, 0.5), iters=30)
        for i in range(3):
            rend.render(*leg.batches[i][:4], background_rgb=bg, cos_anneal_ratio=0.5)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(200):
            rend.render(*leg.batches[i % 32][:4], background_rgb=bg, cos_anneal_ratio=0.5)
        torch.cuda.synchronize()
        dt = time.time() - t0
    print("VDN_SHADE_FUSED=%s  C2 chain (no background): median %.1f us (min %.1f, max %.1f)   render(): %.0f rays/s" % (
        fused, t["median"] * 1e6, t["min"] * 1e6, t["max"] * 1e6, 512 * 200 / dt), flush=True)
