import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from vdn_train import synth, factory
dev = torch.device("cuda", 0)
rend = factory.build_renderer(wdepth=False, device=dev, states=synth.make_all_states(0), precision="bf16")
cams = synth.make_cameras(0)
o, d = synth.random_pixel_batch(0, 0, 0, 512, rank=0, cams=cams)
near, far = synth.near_far_from_sphere(o, d)
b = tuple(torch.tensor(x).to(dev) for x in (o, d, near, far))
bg = torch.ones(1, 3, device=dev)
with torch.no_grad():
    for _ in range(3):
        rend.render(*b, background_rgb=bg, cos_anneal_ratio=0.5)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as p:
        rend.render(*b, background_rgb=bg, cos_anneal_ratio=0.5)
for e in p.events():
    if e.name in ("aten::fill_", "aten::zeros", "aten::zero_", "aten::ones", "aten::full", "aten::copy_"):
        st = [s for s in (e.stack or []) if "repo" in s][:2]
        print(e.name, e.input_shapes, st)
