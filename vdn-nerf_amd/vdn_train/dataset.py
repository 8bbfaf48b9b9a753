"""On-disk formats of a VDN-NeRF scene, decoded on the host once and then resident in HBM:

  <data_dir>/<render_cameras_name>          cameras_sphere*.npz: world_mat_<name>, scale_mat_<name>   (dataset.py:68-91)
  <data_dir>/<img_dir>/<name>.png           RGB or RGBA images                                        (poses.py:116-127)
  <data_dir>/<img_dir>/mask/<name>.png      masks (only read for 3-channel images)                    (dataset.py:61)
  <data_dir>/<img_dir>/<depth_dir>/<name>.npy   wavelet-encoder features [1,96,H/2,W/2]               (dataset.py:62, poses.py:133-146)
  <data_dir>/<img_dir>/depth_from_sdf/sdf_<name>.npy   written back for the wavelet fine-tuning loop   (dpt_runner.py:449-453)

The reference decodes with OpenCV (absent here); Pillow + scipy do the same arithmetic. Channel order follows the
reference: cv.imread yields BGR(A), and trained colour networks / checkpoints are in that order, so images are
flipped to BGR after decoding.
"""
import os
from glob import glob

import numpy as np
import torch


def load_K_Rt_from_P(P):
    """dataset.py:14-35 (IDR's helper) without cv.decomposeProjectionMatrix: P[3,4] = K [R | -R c] up to scale ->
    (intrinsics [4,4] with K[2,2] = 1, pose [4,4] camera-to-world). RQ-decomposes the left 3x3 block with a positive
    diagonal of K and a proper rotation, which is the decomposition OpenCV returns."""
    from scipy.linalg import rq
    P = np.asarray(P, dtype=np.float64)[:3, :4]
    M = P[:, :3]
    if np.linalg.det(M) < 0:                     # P is homogeneous: fix the sign so that R can be a rotation
        P, M = -P, -M
    K, R = rq(M)
    sign = np.diag(np.sign(np.diag(K)))          # K S S R with S = diag(+-1): make diag(K) > 0
    K, R = K @ sign, sign @ R
    centre = -np.linalg.solve(M, P[:, 3])        # camera centre: M c + p4 = 0
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K / K[2, 2]
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.T
    pose[:3, 3] = centre
    return intrinsics, pose


def _read_png(path, unchanged=True):
    """The array cv.imread yields (BGR / BGRA order, which the reference trains in).
    unchanged=True  = cv.imread(path, -1) (poses.py:114, images): the file's own channels, palettes expanded, 16-bit scaled here;
    unchanged=False = cv.imread(path)     (poses.py:125, masks):  always 8-bit 3-channel, whatever the file holds."""
    from PIL import Image
    im = Image.open(path)
    if not unchanged:
        im = im.convert("RGB")                      # 1-bit, palette, grey, RGBA masks -> 3 x 8 bit like IMREAD_COLOR
    elif im.mode in ("P", "1"):
        im = im.convert("RGBA" if "transparency" in im.info else "RGB")
    elif im.mode == "LA":
        im = im.convert("RGBA")
    a = np.asarray(im)
    if a.ndim == 2:
        a = np.repeat(a[:, :, None], 3, axis=2)
    if a.dtype == np.uint16 or (a.dtype.kind in "iu" and a.dtype.itemsize > 1):
        a = (a / 257.0)
    a = a.astype(np.float64)
    return np.concatenate([a[:, :, 2::-1], a[:, :, 3:]], axis=2) if a.shape[2] == 4 else a[:, :, ::-1]


def composite_on_white(decoded, mask_files=None):
    """poses.py:114-129 on decoded 0..255 arrays (what cv.imread yields): 4-channel images are composited on white with their
    own alpha, which becomes the mask (117-122); 3-channel images with the mask files (123-127). float64 arithmetic, one cast
    to float32 at the end, as the reference does -> (images [n,H,W,3], masks [n,H,W,1|3]), float32."""
    images = np.asarray(decoded, dtype=np.float64) / 255.0
    if images.shape[-1] == 4:
        pic, a = images[..., :3], images[..., 3:]
        images, masks = pic * a + (1 - a), a
    else:
        if mask_files is None:
            raise ValueError("3-channel images need their mask files (poses.py:123-127)")
        masks = np.asarray(mask_files, dtype=np.float64) / 255.0
        images = images * masks + (1 - masks)
    return images.astype(np.float32), masks.astype(np.float32)


def normalise_depth_feats(stack, image_size):
    """poses.py:133-146: global mean / std over the whole stack -> sigmoid -> bilinear up-sampling to the image size
    (nn.Upsample(size, mode='bilinear'), align_corners=False) -> [n, H, W, C]."""
    stack = np.asarray(stack)
    m, s = np.mean(stack), np.std(stack)
    feats = torch.sigmoid(torch.from_numpy(((stack - m) / s).astype(np.float32)))
    if feats.dim() == 3:
        feats = feats.unsqueeze(1)
    feats = torch.nn.functional.interpolate(feats, size=tuple(image_size), mode="bilinear", align_corners=False)
    return feats.permute(0, 2, 3, 1).contiguous()


class SceneData:
    """Decoded scene: what the reference's Dataset + RaysGenerator constructors hold (dataset.py:38-108, poses.py:96-152)."""

    def __init__(self, data_dir, img_dir="image", depth_dir="wavelet_feats/0", render_cameras_name="cameras_sphere.npz",
                 with_depth=False):
        self.data_dir, self.img_dir, self.depth_dir = data_dir, img_dir, depth_dir
        self.images_lis = sorted(glob(os.path.join(data_dir, img_dir, "*.png")))
        if not self.images_lis:
            raise FileNotFoundError("no *.png under %s" % os.path.join(data_dir, img_dir))
        names = [os.path.basename(f)[:-4] for f in self.images_lis]
        self.names = names
        self.masks_lis = [os.path.join(data_dir, img_dir, "mask", n + ".png") for n in names]
        self.depth_lis = [os.path.join(data_dir, img_dir, depth_dir, n + ".npy") for n in names]
        self.n_images = len(names)

        cams = np.load(os.path.join(data_dir, render_cameras_name))
        self.world_mats_np = [cams["world_mat_" + n].astype(np.float32) for n in names]
        self.scale_mats_np = [cams["scale_mat_" + n].astype(np.float32) for n in names]
        intr, pose = [], []
        for scale_mat, world_mat in zip(self.scale_mats_np, self.world_mats_np):
            k, p = load_K_Rt_from_P((world_mat @ scale_mat)[:3, :4])
            intr.append(k.astype(np.float32))
            pose.append(p)
        self.intrinsics_all, self.pose_all = np.stack(intr), np.stack(pose)
        self.focal = float(self.intrinsics_all[0][0, 0])

        decoded = np.stack([_read_png(f) for f in self.images_lis])
        mask_files = None if decoded.shape[-1] == 4 else np.stack([_read_png(f, unchanged=False) for f in self.masks_lis])
        self.images, self.masks = composite_on_white(decoded, mask_files)
        self.H, self.W = self.images.shape[1:3]
        self.depth_feats = None
        if with_depth:
            stack = np.stack([np.squeeze(np.load(f)) for f in self.depth_lis])
            self.depth_feats = normalise_depth_feats(stack, (self.H, self.W)).numpy()
            if self.depth_feats.shape[:3] != self.images.shape[:3]:
                raise ValueError("depth features %s do not match images %s" % (self.depth_feats.shape, self.images.shape))

        # region of interest for mesh extraction (dataset.py:98-106)
        bmin, bmax = np.array([-1.01, -1.01, -1.01, 1.0]), np.array([1.01, 1.01, 1.01, 1.0])
        obj = self.scale_mats_np[0]
        inv0 = np.linalg.inv(self.scale_mats_np[0])
        self.object_bbox_min = (inv0 @ obj @ bmin[:, None])[:3, 0]
        self.object_bbox_max = (inv0 @ obj @ bmax[:, None])[:3, 0]

    def rays_generator(self, device="cuda"):
        """The on-device ray source over this scene (vdn_train.rays.RaysGenerator)."""
        from vdn_train.rays import RaysGenerator
        g = RaysGenerator(self.images, self.masks, self.pose_all, self.intrinsics_all, self.depth_feats, device=device)
        g.images_lis = self.images_lis
        return g

    def depth_from_sdf_path(self, idx):
        return os.path.join(self.data_dir, self.img_dir, "depth_from_sdf", "sdf_%s.npy" % self.names[idx])


def write_cameras_npz(path, names, world_mats, scale_mats):
    """Writer of the cameras_sphere.npz schema (colmap_preprocess/gen_cameras.py:43-100 emits the same keys)."""
    d = {}
    for n, w, s in zip(names, world_mats, scale_mats):
        d["world_mat_" + n] = np.asarray(w, dtype=np.float64)
        d["scale_mat_" + n] = np.asarray(s, dtype=np.float64)
    np.savez(path, **d)
