"""SDF value kernel (mode 0) at the sampler's sizes: time per launch and a dump of the values (dev probe; run once per
VDN_SDF0_SPLIT_MAX setting and compare the dumps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np
import torch
from vdn_train import synth, factory

dev = torch.device("cuda", 0)
rend = factory.build_renderer(wdepth=False, device=dev, states=synth.make_all_states(0), precision="bf16")
net = rend.sdf_network
out = {}
g = torch.Generator(device=dev).manual_seed(1)
for P in (100, 8192, 16384, 32768, 65536, 262144):
    pts = (torch.rand(P, 3, device=dev, generator=g) * 2 - 1) * 1.1
    with torch.no_grad():
        for _ in range(5):
            sdf = net._run(0, pts=pts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            sdf = net._run(0, pts=pts)
        e1.record()
        torch.cuda.synchronize()
    out["p%d" % P] = sdf.cpu().numpy()
    print("P=%6d  %.1f us / launch (back to back)   sdf[:3] %s" % (P, e0.elapsed_time(e1) * 10, sdf[:3].tolist()))
# rays form with a column slice (the sampler's calls)
B, n = 512, 16
o = torch.rand(B, 3, device=dev, generator=g) - 0.5
d = torch.nn.functional.normalize(torch.rand(B, 3, device=dev, generator=g) - 0.5, dim=1)
zbuf = torch.rand(B, 128, device=dev, generator=g) * 2
sbuf = torch.zeros(B, 128, device=dev)
with torch.no_grad():
    net._run(0, rays=(o, d, zbuf[:, 64:80]), sdf_out=sbuf[:, 64:80])
out["rays"] = sbuf.cpu().numpy()
np.savez(os.path.join(ROOT, "gpurun_out", "sdf0_%s.npz" % os.environ.get("TAG", "x")), **out)
