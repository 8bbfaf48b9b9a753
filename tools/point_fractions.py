import os, sys
sys.path.insert(0, "vdn-nerf_amd")
import numpy as np, torch
from vdn_train import synth, factory
from vdn_train.trainer import Trainer
dev = torch.device("cuda:0")
B = 512
rend = factory.build_renderer(device=dev, states=synth.make_all_states(0), precision="bf16")
tr = Trainer(rend, B, dev)
cams = synth.make_cameras(0)
g = lambda x: torch.tensor(x).to(dev)
for crop in (None, 420):
    fr1 = fr12 = act = 0
    for step in range(8):
        o, d = synth.random_pixel_batch(0, step, step, B, cams=cams, crop=crop)
        near, far = synth.near_far_from_sphere(o, d)
        tr.train_step(g(o), g(d), g(near), g(far), g(synth.target_colors(o, d)))
        mz = tr.engine.w["mid_z"]
        p = g(o)[:, None, :] + g(d)[:, None, :] * mz[:, :, None]
        pn = p.norm(dim=-1)
        fr1 += (pn >= 1.0).float().mean().item(); fr12 += (pn >= 1.2).float().mean().item()
        act += tr.engine.w["bg_active"][1].item() / tr.engine.Q
    print("crop", crop, "inside samples with |p|>=1: %.3f  |p|>=1.2: %.3f  background active frac %.3f" % (fr1 / 8, fr12 / 8, act / 8))
