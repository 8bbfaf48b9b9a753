import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from vdn_train import synth, factory
dev = torch.device("cuda:0")
g = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev)
fx = load_golden("white_v03_c0")

def run(rend, fused, rays, **kw):
    os.environ["VDN_SHADE_FUSED"] = fused
    with torch.no_grad():
        return rend.render(*rays, background_rgb=torch.ones(1, 3, device=dev), **kw)

rays = (g(fx["rays_o"]), g(fx["rays_d"]), g(fx["near"]), g(fx["far"]))
for seed in (0, 1):
    for var in (0.3, 0.4):
        for first in ("0", "1"):
            for inject in (False, True):
                for car in (0.0, 0.5):
                    st = synth.make_all_states(seed, variance=var)
                    rend = factory.build_renderer(device=dev, states=st, precision="bf16")
                    kw = dict(cos_anneal_ratio=car, t_rand=g(fx["t_rand"]), t_rand_out=g(fx["t_rand_out"]))
                    if inject:
                        kw["z_vals_inject"] = g(fx["z_vals_inside"])
                    a = run(rend, first, rays, **kw)
                    b = run(rend, "1" if first == "0" else "0", rays, **kw)
                    dc = (a["color_fine"] - b["color_fine"]).abs().max().item()
                    dw = (a["weights"] - b["weights"]).abs().max().item()
                    dg = (a["gradients"] - b["gradients"]).abs().max().item()
                    print("seed %d var %.1f first=%s inject=%d car %.1f: dcolor %.2e dweights %.2e dgrad %.2e" % (seed, var, first, inject, car, dc, dw, dg), flush=True)
