"""so(3) helpers of the learnable-pose path: the torch functions of the reference's dpt_models/lie_group_helper.py that
poses.py uses (vec2skew 47-58, Exp 61-71, make_c2w 74-83, convert3x4_4x4 27-44). A pose is 6 numbers per camera; this is
autograd plumbing on 3-vectors, the per-ray work is in the kernels (vdn_gen_rays for fixed poses, the ray adjoint
vdn_ray_adjoint behind NeuSRenderer.render for d loss / d rays)."""
import torch


def convert3x4_4x4(m):
    """[3,4] or [N,3,4] -> homogeneous [4,4] / [N,4,4] (lie_group_helper.py:27-44, torch branch)."""
    if m.dim() == 3:
        out = torch.cat([m, torch.zeros_like(m[:, 0:1])], dim=1)
        out[:, 3, 3] = 1.0
        return out
    return torch.cat([m, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=m.dtype, device=m.device)], dim=0)


def vec2skew(v):
    """(3,) -> the 3x3 cross-product matrix [v]x (lie_group_helper.py:47-58)."""
    zero = torch.zeros(1, dtype=torch.float32, device=v.device)
    return torch.stack([torch.cat([zero, -v[2:3], v[1:2]]),
                        torch.cat([v[2:3], zero, -v[0:1]]),
                        torch.cat([-v[1:2], v[0:1], zero])], dim=0)


def Exp(r):
    """Rodrigues: axis-angle (3,) -> rotation [3,3], with the reference's regulariser |r| + 1e-15 (lie_group_helper.py:61-71)."""
    K = vec2skew(r)
    n = r.norm() + 1e-15
    eye = torch.eye(3, dtype=torch.float32, device=r.device)
    return eye + (torch.sin(n) / n) * K + ((1 - torch.cos(n)) / n ** 2) * (K @ K)


def make_c2w(r, t):
    """axis-angle (3,), translation (3,) -> camera-to-world [4,4] (lie_group_helper.py:74-83)."""
    return convert3x4_4x4(torch.cat([Exp(r), t.unsqueeze(1)], dim=1))
