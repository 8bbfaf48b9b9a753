"""Host wall time of the pieces of the unchanged-runner step (perf_counter around each, device left asynchronous)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
import torch.nn.functional as F
from vdn_train import synth, factory
from vdn_hip import train as T
from dpt_models import renderer as R
acc = collections.defaultdict(float)
def timed(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[tag] += time.perf_counter() - t0
    setattr(obj, name, g)
timed(T.TrainEngine, "forward", "engine.forward")
timed(T.TrainEngine, "backward", "engine.backward")
timed(T.TrainEngine, "param_grads", "param_grads")
timed(T.TrainEngine, "outputs_clone", "outputs_clone")
timed(R.NeuSRenderer, "_sample", "_sample")
timed(R.NeuSRenderer, "_render_train", "_render_train")
timed(R.NeuSRenderer, "render", "render")
timed(R._TrainPlan, "stage", "plan.stage")
timed(R._TrainPlan, "replay_forward", "plan.replay_forward")
timed(R._TrainPlan, "replay_backward", "plan.replay_backward")
dev = torch.device("cuda:0")
seed, B = 0, 512
rend = factory.build_renderer(wdepth=False, device=dev, states=synth.make_all_states(seed), precision=sys.argv[1] if len(sys.argv) > 1 else "bf16")
opt = torch.optim.Adam(rend._all_parameters(), lr=5e-4)
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
    t0 = time.perf_counter()
    mask = torch.ones(B, 1, device=dev)
    mask_sum = mask.sum() + 1e-5
    out = rend.render(rays_o, rays_d, near, far, background_rgb=bg, cos_anneal_ratio=0.5, depth_before_color=False)
    t1 = time.perf_counter()
    color_error = (out["color_fine"] - true_rgb) * mask
    color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / mask_sum
    mask_loss = F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
    loss = color_fine_loss + out["gradient_error"] * 0.1 + mask_loss * 0.0
    t2 = time.perf_counter()
    opt.zero_grad()
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    for k, v in (("A render+mask", t1 - t0), ("B loss ops", t2 - t1), ("C zero_grad", t3 - t2), ("D loss.backward", t4 - t3), ("E opt.step", t5 - t4)):
        acc[k] += v
for i in range(50):
    step(i)
torch.cuda.synchronize()
acc.clear()
N = 300
t0 = time.perf_counter()
for i in range(N):
    step(i)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print("issue %.0f us/step; drained %.0f us/step" % (t_issue / N * 1e6, (time.perf_counter() - t0) / N * 1e6))
for k in sorted(acc):
    print("%-18s %7.0f us/step" % (k, acc[k] / N * 1e6))
