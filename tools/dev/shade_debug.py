import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vdn_train import synth, factory
dev = torch.device("cuda:0")
g = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev)

def run(rend, fused, rays, **kw):
    os.environ["VDN_SHADE_FUSED"] = fused
    with torch.no_grad():
        return rend.render(*rays, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=0.5, **kw)

cams = synth.make_cameras(0)
for B in (37, 255, 256, 257, 512):
    o, d = synth.random_pixel_batch(0, 3, 2, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(0, 3, B)
    rays = (g(o), g(d), g(near), g(far))
    kw = dict(t_rand=g(t1), t_rand_out=g(t2))
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(0, variance=0.4), precision="bf16")
    a = run(rend, "0", rays, **kw)
    b = run(rend, "1", rays, **kw)
    c = run(rend, "1", rays, **kw)
    line = "B %d:" % B
    for k in a:
        if a[k] is None:
            continue
        dab = (a[k] - b[k]).abs().max().item()
        dbc = (b[k] - c[k]).abs().max().item()
        if dab or dbc:
            nbad = int(((a[k] - b[k]).abs().reshape(a[k].shape[0] if a[k].dim() else 1, -1).max(dim=1)[0] > 0).sum().item())
            line += "  %s d(unfused,fused) %.2e [rows differing %d] d(fused,fused) %.2e" % (k, dab, nbad, dbc)
    print(line, flush=True)
