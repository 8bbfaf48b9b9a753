"""Host time of the unchanged-runner flow (bench.py::runner_flow's step) by function: cProfile over 200 steps, device work left
asynchronous (what is listed is what the Python thread spends issuing a step). usage: runner_host_profile.py [bf16|fp32]"""
import cProfile, os, pstats, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
import torch.nn.functional as F
from vdn_train import synth, factory
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
seed, B = 0, 512
rend = factory.build_renderer(wdepth=False, device=dev, states=synth.make_all_states(seed), precision=prec)
params = rend._all_parameters()
opt = torch.optim.Adam(params, lr=5e-4)
cams = synth.make_cameras(seed)
g = lambda x: torch.tensor(x).to(dev)
batches = []
for s in range(16):
    o, d = synth.random_pixel_batch(seed, s, s % len(cams), B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    batches.append((g(o), g(d), g(near), g(far), g(synth.target_colors(o, d))))
bg = torch.ones([1, 3], device=dev)
def step(i):
    rays_o, rays_d, near, far, true_rgb = batches[i % len(batches)]
    mask = torch.ones(B, 1, device=dev)
    mask_sum = mask.sum() + 1e-5
    out = rend.render(rays_o, rays_d, near, far, background_rgb=bg, cos_anneal_ratio=0.5, depth_before_color=False)
    color_error = (out["color_fine"] - true_rgb) * mask
    color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / mask_sum
    mask_loss = F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
    loss = color_fine_loss + out["gradient_error"] * 0.1 + mask_loss * 0.0
    opt.zero_grad()
    loss.backward()
    opt.step()
for i in range(30):
    step(i)
torch.cuda.synchronize()
# host time per step with the device left behind (no sync inside): wall of issuing N steps
t0 = time.time()
for i in range(100):
    step(i)
t_issue = (time.time() - t0) / 100
torch.cuda.synchronize()
t_all = (time.time() - t0) / 100
print("issue %.0f us/step, with the device drained %.0f us/step" % (t_issue * 1e6, t_all * 1e6))
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
