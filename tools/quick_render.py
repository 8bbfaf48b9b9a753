"""Render B rays a few times (forward only) - used under rocprofv3 for per-kernel timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np
import torch
from vdn_train import synth, factory

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wdepth = (sys.argv[3] == "wdepth") if len(sys.argv) > 3 else False
dev = torch.device("cuda:0")
st = synth.make_all_states(0, wdepth=wdepth)
rend = factory.build_renderer(wdepth=wdepth, device=dev, states=st)
cams = synth.make_cameras(0)
o, d = synth.random_pixel_batch(0, 0, 0, B, cams=cams)
near, far = synth.near_far_from_sphere(o, d)
g = lambda x: torch.tensor(x).to(dev)
o, d, near, far = g(o), g(d), g(near), g(far)
bg = torch.ones(1, 3, device=dev)
for it in range(iters + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.time()
    out = rend.render(o, d, near, far, background_rgb=bg, cos_anneal_ratio=0.5)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
print("B=%d wdepth=%s  %.3f ms/render  %.0f rays/s  color mean %.4f" % (B, wdepth, dt * 1e3, B / dt, out["color_fine"].mean().item()))
