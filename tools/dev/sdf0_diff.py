import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np
import torch
from vdn_train import synth, factory
dev = torch.device("cuda", 0)
tag = os.environ.get("TAG", "x")
prec = os.environ.get("PREC", "bf16")
rend = factory.build_renderer(wdepth=False, device=dev, states=synth.make_all_states(0), precision=prec)
net = rend.sdf_network
g = torch.Generator(device=dev).manual_seed(1)
pts = (torch.rand(8192, 3, device=dev, generator=g) * 2 - 1) * 1.1
with torch.no_grad():
    runs = [net._run(0, pts=pts).cpu().numpy() for _ in range(4)]
print(tag, "self-consistent:", all(np.array_equal(runs[0], r) for r in runs[1:]))
np.save(os.path.join(ROOT, "gpurun_out", "sdf0d_%s.npy" % tag), runs[0])
