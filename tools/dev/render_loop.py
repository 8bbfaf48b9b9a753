"""A loop of NeuSRenderer.render() calls on one 512-ray batch (inference path, bf16), for a rocprofv3 kernel trace
(tools/dev/timeline.py prints one call of it), a counter run or a wall-clock figure.
usage: render_loop.py [calls=200] [B=512] [crop=420|0: full frame]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import torch
from vdn_train import synth, factory
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
crop = int(sys.argv[3]) if len(sys.argv) > 3 else 420
dev = torch.device("cuda", 0)
rend = factory.build_renderer(wdepth=False, device=dev, states=synth.make_all_states(0, variance=0.4), precision="bf16")
cams = synth.make_cameras(0)
o, d = synth.random_pixel_batch(0, 0, 0, B, rank=0, cams=cams, crop=crop if crop > 0 else None)
near, far = synth.near_far_from_sphere(o, d)
b = tuple(torch.tensor(x).to(dev) for x in (o, d, near, far))
bg = torch.ones(1, 3, device=dev)
with torch.no_grad():
    for _ in range(20):
        rend.render(*b, background_rgb=bg, cos_anneal_ratio=0.5)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        out = rend.render(*b, background_rgb=bg, cos_anneal_ratio=0.5)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
print("render(): %.1f us per call of %d rays = %.3f M rays/s" % (dt * 1e6, B, B / dt / 1e6))
