"""The hierarchical sampler alone (NeuSRenderer._sample: coarse depths, first SDF pass + round, three rounds), whole render() calls and
the training step, for A/B runs of process-wide switches (VDN_SDF_FIRST_SPLIT ...).  usage: sampler_probe.py [tag]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse
import numpy as np
import torch
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
leg = bench.Leg(argparse.Namespace(batch=512), torch.device("cuda:0"), 1, 0, "bf16", False, 64)
for i in range(700):
    leg.step(i)
torch.cuda.synchronize()
rend = leg.rend
o, d, near, far = leg.batches[0][:4]
bg = torch.ones(1, 3, device="cuda:0")
with torch.no_grad():
    def sample():
        rend._sample(o, d, near.reshape(-1), far.reshape(-1), rend.perturb, None, None, None, defer_last_merge=True)
    for _ in range(5):
        sample()
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sample(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    # back to back (the device never idles between the calls)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        sample()
    e1.record()
    torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) * 10.0
    for _ in range(5):
        rend.render(o, d, near, far, background_rgb=bg, cos_anneal_ratio=0.5)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(300):
        rend.render(*leg.batches[i % 64][:4], background_rgb=bg, cos_anneal_ratio=0.5)
    torch.cuda.synchronize()
    rate = 300 * 512 / (time.time() - t0)
res = []
for r in range(5):
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(40):
        leg.step(700 + r * 40 + i)
    torch.cuda.synchronize()
    res.append((time.time() - t0) / 40 * 1e6)
print("%-12s sampler %.1f us (min %.1f; back to back %.1f)  render() %.3f M rays/s  step %.1f us (min %.1f)" % (
    tag, float(np.median(ts)), min(ts), b2b, rate / 1e6, float(np.median(res)), min(res)))
