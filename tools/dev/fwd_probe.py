"""Inference render(): host enqueue time vs device time per ray batch (dev probe)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200)
ap.add_argument("--batch", type=int, default=512)
a = ap.parse_args()
from vdn_train import synth, factory
dev = torch.device("cuda", 0)
st = synth.make_all_states(0, wdepth=False)
rend = factory.build_renderer(wdepth=False, device=dev, states=st, precision="bf16")
cams = synth.make_cameras(0)
g = lambda x: torch.tensor(x).to(dev)
bs = []
for s in range(8):
    o, d = synth.random_pixel_batch(0, s, s % len(cams), a.batch, rank=0, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    bs.append((g(o), g(d), g(near), g(far)))
bg = torch.ones(1, 3, device=dev)
with torch.no_grad():
    for i in range(5):
        rend.render(*bs[i % 8], background_rgb=bg, cos_anneal_ratio=0.5)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.time()
        for i in range(a.n):
            rend.render(*bs[i % 8], background_rgb=bg, cos_anneal_ratio=0.5)
        t1 = time.time()
        torch.cuda.synchronize()
        t2 = time.time()
        print("B=%d host enqueue %.1f us/batch, wall %.1f us/batch -> %.0f rays/s" % (a.batch, (t1 - t0) / a.n * 1e6, (t2 - t0) / a.n * 1e6, a.batch * a.n / (t2 - t0)))

    plan = rend.plan(a.batch, background_rgb=bg, cos_anneal_ratio=0.5)
    for i in range(5):
        plan(*bs[i % 8])
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.time()
        for i in range(a.n):
            plan(*bs[i % 8])
        t1 = time.time()
        torch.cuda.synchronize()
        t2 = time.time()
        print("PLAN B=%d host enqueue %.1f us/batch, wall %.1f us/batch -> %.0f rays/s" % (a.batch, (t1 - t0) / a.n * 1e6, (t2 - t0) / a.n * 1e6, a.batch * a.n / (t2 - t0)))
