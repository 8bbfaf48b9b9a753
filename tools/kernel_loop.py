"""Launch one kernel of the path repeatedly (for rocprofv3 --pmc runs).  usage: kernel_loop.py <sdf1|sdf1t|sdf0|nerf|color> [P] [precision] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
which = sys.argv[1]; P = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"; iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
rend = factory.build_renderer(device=dev, states=synth.make_all_states(0), precision=prec)
B = P // 128
cams = synth.make_cameras(0)
o, d = synth.random_pixel_batch(0, 0, 0, B, cams=cams)
near, far = synth.near_far_from_sphere(o, d)
g = lambda x: torch.tensor(x).to(dev)
o, d = g(o), g(d)
z = (g(near) + (g(far) - g(near)) * torch.linspace(0, 1, 128, device=dev)[None, :]).contiguous()
eng = None
if which == "sdf1t":                      # the fused kernel as the training step launches it (saves S, H and the PE planes)
    from vdn_train.trainer import Trainer
    eng = Trainer(rend, batch_size=B, device=dev).engine
    eng.w["mid_z"].copy_(z)
with torch.no_grad():
    for i in range(iters):
        if which == "sdf1t":
            eng._sdf_forward(o, d)
        elif which == "sdf1":
            sdf, feat, nrm = rend.sdf_network._run(1, rays=(o, d, z))
        elif which == "sdf0":
            rend.sdf_network._run(0, rays=(o, d, z))
        elif which == "nerf":
            rend.nerf._run(rays=(o, d, z))
        elif which == "color":
            if i == 0:
                sdf, feat, nrm = rend.sdf_network._run(1, rays=(o, d, z))
            rend.color_network._run(nrm, feat, rays=(o, d, z))
torch.cuda.synchronize()
print("done", which, P, prec)
