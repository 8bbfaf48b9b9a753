"""Learnable cameras of the reference's dpt_models/poses.py on MI355X: LearnPose (16-47), LearnIntrin (50-93) and the
learnable branch of RaysGenerator.gen_random_rays_at / gen_rays_at (168-212).

The modules are parameter holders (6 numbers per camera, one focal coefficient); rays built from them carry a graph, and
NeuSRenderer.render() differentiates through rays_o / rays_d / near / far with the hand-written ray adjoint
(vdn_ray_adjoint + the networks' input adjoints, include/vdn_render.h), so `loss.backward()` reaches LearnPose.r / .t as in
the reference's `*_learn_*` configurations. The fixed-pose generator with resident images is vdn_train.rays.RaysGenerator.
"""
import numpy as np
import torch
import torch.nn as nn

from dpt_models.lie_group_helper import make_c2w


class LearnPose(nn.Module):
    def __init__(self, num_cams, learn_R, learn_t, init_c2w=None):
        """poses.py:16-36. init_c2w: [N,4,4] tensor, a .npy path, or None."""
        super().__init__()
        self.num_cams = num_cams
        self.init_c2w = None
        if isinstance(init_c2w, str):
            init_c2w = torch.stack([torch.from_numpy(p) for p in np.load(init_c2w).astype(np.float32)])
        if init_c2w is not None:
            self.init_c2w = nn.Parameter(init_c2w, requires_grad=False)
        self.r = nn.Parameter(torch.zeros(size=(num_cams, 3), dtype=torch.float32), requires_grad=learn_R)
        self.t = nn.Parameter(torch.zeros(size=(num_cams, 3), dtype=torch.float32), requires_grad=learn_t)

    def forward(self, cam_id):
        c2w = make_c2w(self.r[cam_id], self.t[cam_id])             # poses.py:38-47: a delta on the initial pose
        if self.init_c2w is not None:
            c2w = c2w @ self.init_c2w[cam_id]
        return c2w


class LearnIntrin(nn.Module):
    def __init__(self, H, W, req_grad, fx_only=True, order=2, init_focal=None):
        """poses.py:50-78. fx is the focal coefficient: focal = fx^order * W."""
        super().__init__()
        self.H, self.W, self.order = H, W, order
        if isinstance(init_focal, str):
            init_focal = np.load(init_focal)
        if init_focal is None:
            self.fx = nn.Parameter(torch.tensor(1.0, dtype=torch.float32), requires_grad=req_grad)
        else:
            init_focal = torch.as_tensor(init_focal, dtype=torch.float32)
            if order == 2:
                coe_x = torch.sqrt(init_focal / float(W)).clone().detach().float()
            elif order == 1:
                coe_x = (init_focal / float(W)).clone().detach().float()
            else:
                raise ValueError("Focal init order need to be 1 or 2")
            self.fx = nn.Parameter(coe_x, requires_grad=req_grad)

    def forward(self, i=None):
        # poses.py:80-93 builds the matrix from fx.item(): the intrinsics the rays see are a constant of the step (no
        # gradient reaches fx through them in the reference either)
        fx = self.fx.item()
        f = fx ** 2 * self.W if self.order == 2 else fx * self.W
        k = np.array([[f, 0.0, self.W / 2, 0.0], [0.0, f, self.H / 2, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
        return torch.from_numpy(k).to(self.fx.device)


class LearnableRays:
    """The learnable branch of RaysGenerator (poses.py:168-212): rays from pose_net(img_idx) and inverse(intrin_net()),
    attached to the pose parameters. Pixel colours / masks / features come from the resident arrays of a
    vdn_train.rays.RaysGenerator (`pixels`), so an iteration stays on the device."""

    def __init__(self, pose_net, intrin_net, pixels):
        self.pose_net, self.intrin_net, self.pixels = pose_net, intrin_net, pixels
        self.H, self.W, self.device = pixels.H, pixels.W, pixels.device

    def _rays(self, img_idx, px, py):
        pose = self.pose_net(img_idx)
        intrinsic_inv = torch.inverse(self.intrin_net())
        p = torch.stack([px, py, torch.ones_like(py)], dim=-1).float()
        p = torch.matmul(intrinsic_inv[None, :3, :3], p[:, :, None]).squeeze(-1)
        rays_v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
        rays_v = torch.matmul(pose[None, :3, :3], rays_v[:, :, None]).squeeze(-1)
        rays_o = pose[None, :3, 3].expand(rays_v.shape)
        return rays_o, rays_v

    def gen_random_rays_at(self, img_idx, batch_size, pixels=None):
        """poses.py:189-212 -> [B, 10 + C] = rays_o | rays_d | mask | rgb | feats; columns 0..5 carry the pose graph."""
        if pixels is None:
            px = torch.randint(low=0, high=self.W, size=[batch_size], device=self.device).float()
            py = torch.randint(low=0, high=self.H, size=[batch_size], device=self.device).float()
        else:
            px, py = (torch.as_tensor(p, dtype=torch.float32).to(self.device).contiguous() for p in pixels)
        fixed = self.pixels.gen_random_rays_at(int(img_idx), batch_size, pixels=(px, py))       # mask | rgb | feats gathers
        rays_o, rays_v = self._rays(img_idx, px, py)
        return torch.cat([rays_o, rays_v, fixed[:, 6:]], dim=-1)

    def gen_rays_at(self, img_idx, resolution_level=1):
        """poses.py:168-187 -> rays_o, rays_v [H/l, W/l, 3]."""
        l = resolution_level
        tx = torch.linspace(0, self.W - 1, self.W // l)
        ty = torch.linspace(0, self.H - 1, self.H // l)
        pixels_x, pixels_y = torch.meshgrid(tx, ty, indexing="ij")
        px, py = pixels_x.reshape(-1).to(self.device), pixels_y.reshape(-1).to(self.device)
        o, v = self._rays(img_idx, px, py)
        shape = (self.W // l, self.H // l, 3)
        return o.reshape(shape).transpose(0, 1), v.reshape(shape).transpose(0, 1)
