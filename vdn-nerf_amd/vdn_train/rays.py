"""On-device ray source: the fixed-pose part of the reference's RaysGenerator
(dpt_models/poses.py:96-212) with images, masks and VDN target features resident in HBM, so a training
iteration never leaves the GPU (the reference indexes CPU images and does `.cpu() ... .cuda()` every
iteration, poses.py:212). Decoding image / camera files stays out of scope: pass decoded arrays."""
import numpy as np
import torch

from vdn_hip import lib


_stream = lib.stream_handle          # the HIP handle of torch's current stream


class RaysGenerator:
    def __init__(self, images, masks, pose_all, intrinsics_all, depth_feats=None, device="cuda"):
        """images [n,H,W,3] in [0,1]; masks [n,H,W,1|3] or None; pose_all [n,4,4] c2w; intrinsics_all [n,4,4] or [4,4];
        depth_feats [n,H,W,C] or None (already normalised / up-sampled as poses.py:133-146 does)."""
        dev = torch.device(device)
        f = lambda x: None if x is None else torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev).contiguous()
        self.device = dev
        self.images, self.masks, self.depth_feats = f(images), f(masks), f(depth_feats)
        self.pose_all = f(pose_all)
        K = torch.as_tensor(np.asarray(intrinsics_all), dtype=torch.float32)
        if K.dim() == 2:
            K = K[None].expand(self.pose_all.shape[0], -1, -1)
        self.intrin_inv = torch.inverse(K)[:, :3, :3].contiguous().to(dev)        # poses.py:107
        self.n_images, self.H, self.W = self.images.shape[0], self.images.shape[1], self.images.shape[2]
        self.with_depth = self.depth_feats is not None
        self.C = self.depth_feats.shape[-1] if self.with_depth else 0

    def _launch(self, img_idx, px, py, with_pixels=True):
        B = px.numel()
        ld = (10 + (self.C if self.with_depth else 1)) if with_pixels else 6
        out = torch.zeros(B, ld, dtype=torch.float32, device=self.device)
        near = torch.empty(B, 1, dtype=torch.float32, device=self.device)
        far = torch.empty(B, 1, dtype=torch.float32, device=self.device)
        a = lib.VdnGenRaysArgs()
        a.pixels_x, a.pixels_y = px.data_ptr(), py.data_ptr()
        a.intrinsic_inv, a.pose = self.intrin_inv[img_idx].data_ptr(), self.pose_all[img_idx].data_ptr()
        if with_pixels:
            a.image = self.images[img_idx].data_ptr()
            if self.masks is not None:
                a.mask, a.mask_ch = self.masks[img_idx].data_ptr(), self.masks.shape[-1]
            if self.with_depth:
                a.feats, a.C = self.depth_feats[img_idx].data_ptr(), self.C
        a.out, a.near, a.far = out.data_ptr(), near.data_ptr(), far.data_ptr()
        a.B, a.H, a.W, a.out_ld = B, self.H, self.W, ld
        lib.call("vdn_gen_rays", a, _stream())
        return out, near, far

    def gen_random_rays_at(self, img_idx, batch_size, pixels=None, return_near_far=False):
        """poses.py:189-212: -> [B, 10 + C] = rays_o | rays_d | mask | rgb | feats (feats is one zero column when the
        generator holds no depth features, as the reference's `torch.zeros([B,1])`)."""
        if pixels is None:
            px = torch.randint(low=0, high=self.W, size=[batch_size], device=self.device).float()
            py = torch.randint(low=0, high=self.H, size=[batch_size], device=self.device).float()
        else:
            px, py = (torch.as_tensor(p, dtype=torch.float32).to(self.device).contiguous() for p in pixels)
        out, near, far = self._launch(int(img_idx), px, py)
        return (out, near, far) if return_near_far else out

    def gen_rays_at(self, img_idx, resolution_level=1):
        """poses.py:168-187: -> rays_o, rays_v [H/l, W/l, 3]."""
        l = resolution_level
        tx = torch.linspace(0, self.W - 1, self.W // l)
        ty = torch.linspace(0, self.H - 1, self.H // l)
        pixels_x, pixels_y = torch.meshgrid(tx, ty, indexing="ij")            # [W/l, H/l]
        px = pixels_x.reshape(-1).to(self.device).contiguous()
        py = pixels_y.reshape(-1).to(self.device).contiguous()
        out, _, _ = self._launch(int(img_idx), px, py, with_pixels=False)
        o = out[:, 0:3].reshape(self.W // l, self.H // l, 3).transpose(0, 1)
        v = out[:, 3:6].reshape(self.W // l, self.H // l, 3).transpose(0, 1)
        return o, v

    def image_at(self, idx, resolution_level=1):
        """poses.py:254-256: image `idx` in 0..255 at 1/resolution_level size, channel order as loaded (BGR). cv.resize's default
        INTER_LINEAR = bilinear taps at half-pixel centres, edges clamped, no anti-aliasing (F.interpolate, align_corners=False)."""
        img = self.images[int(idx)]
        l = int(resolution_level)
        if l != 1:
            img = torch.nn.functional.interpolate(img.permute(2, 0, 1)[None], size=(self.H // l, self.W // l), mode="bilinear",
                                                  align_corners=False, antialias=False)[0].permute(1, 2, 0)
        return (img.cpu().numpy() * 255).clip(0, 255)

    def mask_at(self, idx, resolution_level=1):
        """poses.py:258-261: mask `idx` at 1/resolution_level size, [H/l, W/l, 1] (cv.resize drops the single channel of the alpha mask,
        the reference puts it back)."""
        msk = self.masks[int(idx)][..., :1]
        l = int(resolution_level)
        if l != 1:
            msk = torch.nn.functional.interpolate(msk.permute(2, 0, 1)[None], size=(self.H // l, self.W // l), mode="bilinear",
                                                  align_corners=False, antialias=False)[0].permute(1, 2, 0)
        return msk.cpu().numpy()

    def gen_rays_between(self, ratio, idx_0, idx_1, resolution_level=1):
        """poses.py:214-252: rays of a camera interpolated between two views - translation linearly, rotation by Slerp, both
        on the world-to-camera side as the reference does - with the first camera's intrinsics. -> rays_o, rays_v [H/l, W/l, 3].
        (The pose algebra is a 4x4 on the host, as in the reference; the rays come from vdn_gen_rays.)"""
        from scipy.spatial.transform import Rotation as Rot, Slerp
        l = resolution_level
        pose_0 = self.pose_all[int(idx_0)].detach().cpu().numpy().astype(np.float64)
        pose_1 = self.pose_all[int(idx_1)].detach().cpu().numpy().astype(np.float64)
        pose_0, pose_1 = np.linalg.inv(pose_0), np.linalg.inv(pose_1)
        rot = Slerp([0, 1], Rot.from_matrix(np.stack([pose_0[:3, :3], pose_1[:3, :3]])))(ratio)
        pose = np.diag([1.0, 1.0, 1.0, 1.0]).astype(np.float32)
        pose[:3, :3] = rot.as_matrix()
        pose[:3, 3] = ((1.0 - ratio) * pose_0 + ratio * pose_1)[:3, 3]
        pose = np.linalg.inv(pose)
        tx = torch.linspace(0, self.W - 1, self.W // l)
        ty = torch.linspace(0, self.H - 1, self.H // l)
        pixels_x, pixels_y = torch.meshgrid(tx, ty, indexing="ij")
        px = pixels_x.reshape(-1).to(self.device).contiguous()
        py = pixels_y.reshape(-1).to(self.device).contiguous()
        B = px.numel()
        out = torch.zeros(B, 6, dtype=torch.float32, device=self.device)
        near = torch.empty(B, 1, dtype=torch.float32, device=self.device)
        far = torch.empty(B, 1, dtype=torch.float32, device=self.device)
        pose_t = torch.tensor(pose, dtype=torch.float32, device=self.device).contiguous()
        a = lib.VdnGenRaysArgs()
        a.pixels_x, a.pixels_y = px.data_ptr(), py.data_ptr()
        a.intrinsic_inv, a.pose = self.intrin_inv[0].data_ptr(), pose_t.data_ptr()          # poses.py:229: intrin_inv[0]
        a.out, a.near, a.far = out.data_ptr(), near.data_ptr(), far.data_ptr()
        a.B, a.H, a.W, a.out_ld = B, self.H, self.W, 6
        lib.call("vdn_gen_rays", a, _stream())
        o = out[:, 0:3].reshape(self.W // l, self.H // l, 3).transpose(0, 1)
        v = out[:, 3:6].reshape(self.W // l, self.H // l, 3).transpose(0, 1)
        return o, v

    @staticmethod
    def near_far_from_sphere(rays_o, rays_d):
        """dataset.py:111-118 (torch ops on device tensors; the kernel also returns them fused)."""
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return mid - 1.0, mid + 1.0
