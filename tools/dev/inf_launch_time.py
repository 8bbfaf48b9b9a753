"""The fused SDF kernel's inference launch (65 536 points, ray form) and forward render() on the library VDN_LIB names: medians, for
same-box A/B runs of library variants (two processes, alternating).  usage: inf_launch_time.py [launches=60]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
rend = factory.build_renderer(device=dev, states=synth.make_all_states(0), precision="bf16")
B = 512
cams = synth.make_cameras(0)
o, d = synth.random_pixel_batch(0, 0, 0, B, cams=cams)
near, far = synth.near_far_from_sphere(o, d)
g = lambda x: torch.tensor(x).to(dev)
o, d, near, far = g(o), g(d), g(near), g(far)
z = (near + (far - near) * torch.linspace(0, 1, 128, device=dev)[None, :]).contiguous()
with torch.no_grad():
    for _ in range(200):          # ~ 30 ms of back-to-back launches: the clock the kernel holds under its own load
        rend.sdf_network._run(1, rays=(o, d, z))
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rend.sdf_network._run(1, rays=(o, d, z)); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    bg = torch.ones(1, 3, device=dev)
    for _ in range(20):
        rend.render(o, d, near, far, background_rgb=bg, cos_anneal_ratio=0.5)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(300):
        rend.render(o, d, near, far, background_rgb=bg, cos_anneal_ratio=0.5)
    torch.cuda.synchronize()
    rps = 300 * B / (time.time() - t0)
print("%-28s inference launch median %.1f us min %.1f (%.3f of 2.5 PFLOP/s)   render() %.0f K rays/s" % (
    os.path.basename(os.environ.get("VDN_LIB", "libvdn_render.so")), np.median(ts), np.min(ts), 1967104.0 * 65536 / (np.median(ts) * 1e-6) / 2.5e15, rps / 1e3))
