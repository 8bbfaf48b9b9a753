"""Pose refinement through the ray adjoint: targets are rendered by the (frozen) networks from the true camera; a perturbed
LearnPose is optimised against them with torch.optim.Adam. Prints loss and pose error over the steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
from dpt_models.poses import LearnPose
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
seed, B, steps = 0, 512, int(sys.argv[2]) if len(sys.argv) > 2 else 300
st = synth.make_all_states(seed, wdepth=False, variance=0.3)
rend = factory.build_renderer(device=dev, states=st, precision=prec)
for p in rend._all_parameters():
    p.requires_grad_(False)
cams = torch.tensor(np.asarray(synth.make_cameras(seed)[:4], np.float32)).to(dev)
Kinv = torch.tensor(synth.intrinsics_inv().astype(np.float32)).to(dev)
pose_net = LearnPose(4, True, True, init_c2w=cams.clone()).to(dev)
true_r = torch.tensor([0.02, -0.015, 0.01], device=dev)
true_t = torch.tensor([0.05, -0.04, 0.03], device=dev)
cam = 2
from dpt_models.lie_group_helper import make_c2w
true_pose = make_c2w(true_r, true_t) @ cams[cam]


def rays(pose, px, py):
    p = torch.stack([px, py, torch.ones_like(py)], dim=-1)
    p = torch.matmul(Kinv[None], p[:, :, None]).squeeze(-1)
    v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
    v = torch.matmul(pose[None, :3, :3], v[:, :, None]).squeeze(-1)
    return pose[None, :3, 3].expand(v.shape), v


def near_far(o, d):
    mid = 0.5 * (-(2.0 * (o * d).sum(-1, keepdim=True))) / (d * d).sum(-1, keepdim=True)
    return mid - 1.0, mid + 1.0


opt = torch.optim.Adam([pose_net.r, pose_net.t], lr=2e-3)
g = torch.Generator(device="cpu").manual_seed(1)
for it in range(steps):
    px = (torch.rand(B, generator=g) * 500 + 150).floor().to(dev)
    py = (torch.rand(B, generator=g) * 500 + 150).floor().to(dev)
    with torch.no_grad():
        o, d = rays(true_pose, px, py)
        n, f = near_far(o, d)
        target = rend.render(o, d, n, f, perturb_overwrite=0, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=1.0)["color_fine"]
    o, d = rays(pose_net(cam), px, py)
    n, f = near_far(o, d)
    out = rend.render(o, d, n, f, perturb_overwrite=0, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=1.0)
    loss = (out["color_fine"] - target).abs().mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
    if it % 25 == 0 or it == steps - 1:
        er = (pose_net.r[cam].detach() - true_r).norm().item()
        et = (pose_net.t[cam].detach() - true_t).norm().item()
        print("step %3d  loss %.5f  |r - r*| %.4f  |t - t*| %.4f" % (it, loss.item(), er, et), flush=True)
