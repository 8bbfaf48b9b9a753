"""render() under grad as graph replays against eager launches over MANY optimizer steps (injected jitter, changing batches, annealing
ratio and background): final parameters and every 10th step's outputs must be bit-identical.  usage: graph_long_run.py [steps=300] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
import torch.nn.functional as F
from vdn_train import synth, factory
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
cams = synth.make_cameras(9)
tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
B = 512
data = []
for it in range(16):
    o, d = synth.random_pixel_batch(9, it, it % len(cams), B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(9, it, B)
    data.append(tuple(tt(x) for x in (o, d, near, far, t1, t2, synth.target_colors(o, d))))


def run(graphs):
    os.environ["VDN_RENDER_GRAPHS"] = "1" if graphs else "0"
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(9, variance=0.35), precision=prec)
    params = rend._all_parameters()
    opt = torch.optim.Adam(params, lr=5e-4)
    keep = []
    for it in range(steps):
        o, d, near, far, t1, t2, rgb = data[it % len(data)]
        out = rend.render(o, d, near, far, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=min(1.0, it / 100.0), t_rand=t1, t_rand_out=t2)
        mask = torch.ones(B, 1, device=dev)
        loss = (out["color_fine"] - rgb).abs().sum() / B + 0.1 * out["gradient_error"] + 0.0 * F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1 - 1e-3), mask)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 10 == 0:
            keep.append((out["color_fine"].detach().clone(), out["weights"].detach().clone(), float(loss)))
    return keep, [p.detach().clone() for p in params]

ka, pa = run(True)
kb, pb = run(False)
same_out = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(ka, kb))
same_par = all(torch.equal(a, b) for a, b in zip(pa, pb))
print("%d steps, %s: outputs of every 10th step bit-identical: %s; final parameters bit-identical: %s; loss %.6f -> %.6f (graphs) / %.6f (eager)" % (
    steps, prec, same_out, same_par, ka[0][2], ka[-1][2], kb[-1][2]))
sys.exit(0 if (same_out and same_par) else 1)
