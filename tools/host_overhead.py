"""Host-side issue time of a training step vs its GPU time (is the step launch-bound?). usage: host_overhead.py [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
from vdn_train.trainer import Trainer
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
B = 512
tr = Trainer(factory.build_renderer(device=dev, states=synth.make_all_states(0), precision=prec), B, dev)
cams = synth.make_cameras(0)
g = lambda x: torch.tensor(x).to(dev)
bs = []
for s in range(24):
    o, d = synth.random_pixel_batch(0, s, s, B, cams=cams)
    n, f = synth.near_far_from_sphere(o, d)
    bs.append((g(o), g(d), g(n), g(f), g(synth.target_colors(o, d))))
for b in bs[:4]:
    tr.train_step(*b)
torch.cuda.synchronize()
t0 = time.time()
for b in bs[4:]:
    tr.train_step(*b)
t_issue = time.time() - t0
torch.cuda.synchronize()
t_all = time.time() - t0
print("steps %d: host issue %.3f ms/step, wall %.3f ms/step" % (len(bs) - 4, t_issue / 20 * 1e3, t_all / 20 * 1e3))
