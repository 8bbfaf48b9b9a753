"""Whole-image rendering through the HIP renderer: the arithmetic of Runner.val_img (dpt_runner.py:417-491) - rays of
one camera in batches, colour image, L1 / PSNR against the ground truth, and the weight-argmax depth written back as
`depth_from_sdf/sdf_<name>.npy` for the wavelet fine-tuning loop (dpt_runner.py:449-453). Everything stays on the device
until the final image copy (the reference copies every batch to the host)."""
import os

import numpy as np
import torch


@torch.no_grad()
def render_image(renderer, rays_gen, idx, resolution_level=1, batch_size=512, cos_anneal_ratio=1.0, white_bkgd=True,
                 gen_depth_for_finetune=False):
    """-> dict(img_fine [H,W,3] float32 in [0,1], gradient_error [n_batches], weight_depth [H,W,1] | None)."""
    rays_o, rays_d = rays_gen.gen_rays_at(idx, resolution_level=resolution_level)
    H, W, _ = rays_o.shape
    rays_o, rays_d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
    dev = rays_o.device
    bg = torch.ones(1, 3, device=dev) if white_bkgd else None
    rgb = torch.empty(H * W, 3, device=dev)
    depth = torch.empty(H * W, 1, device=dev) if gen_depth_for_finetune else None
    eik = []
    for s in range(0, H * W, batch_size):
        o, d = rays_o[s:s + batch_size], rays_d[s:s + batch_size]
        near, far = rays_gen.near_far_from_sphere(o, d)
        out = renderer.render(o, d, near, far, cos_anneal_ratio=cos_anneal_ratio, background_rgb=bg)
        rgb[s:s + o.shape[0]] = out["color_fine"]
        eik.append(out["gradient_error"])
        if gen_depth_for_finetune:                                   # dpt_runner.py:449-455
            inside = out["inside_sphere"]
            w = out["weights"][:, :inside.shape[1]] * inside
            depth[s:s + o.shape[0]] = out["z_vals"].gather(1, torch.argmax(w, dim=-1, keepdim=True))
        del out
    return {"img_fine": rgb.reshape(H, W, 3).cpu().numpy(), "gradient_error": torch.stack(eik).cpu().numpy(),
            "weight_depth": depth.reshape(H, W, 1).cpu().numpy() if gen_depth_for_finetune else None}


def image_metrics(img_fine, gt, mask=None):
    """color L1 and PSNR of dpt_runner.py:470-474 (mask = ones when masks are not used)."""
    mask = np.ones_like(gt[..., :1]) if mask is None else (mask > 0.1).astype(np.float32)
    mask_sum = mask.sum() + 1e-5
    l1 = np.abs((img_fine - gt) * mask).sum() / mask_sum
    psnr = 20.0 * np.log10(1.0 / np.sqrt(((img_fine - gt) ** 2 * mask).sum() / (mask_sum * 3.0)))
    return float(l1), float(psnr)


def val_img(renderer, scene, rays_gen, idx, resolution_level=1, batch_size=512, cos_anneal_ratio=1.0, white_bkgd=True,
            use_mask=False, gen_depth_for_finetune=False):
    """Runner.val_img for a vdn_train.dataset.SceneData: -> (color_loss, psnr, gradient_error, img_fine); with
    gen_depth_for_finetune also writes depth_from_sdf/sdf_<name>.npy under the scene directory."""
    res = render_image(renderer, rays_gen, idx, resolution_level, batch_size, cos_anneal_ratio, white_bkgd, gen_depth_for_finetune)
    step = resolution_level
    H, W = res["img_fine"].shape[:2]
    gt = _resize(scene.images[idx], H, W, step)
    mask = _resize(scene.masks[idx], H, W, step)[..., :1] if use_mask else None
    l1, psnr = image_metrics(res["img_fine"], gt, mask)
    if gen_depth_for_finetune:
        path = scene.depth_from_sdf_path(idx)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.save(path, res["weight_depth"])
    return l1, psnr, res["gradient_error"], res["img_fine"]


def _resize(img, H, W, level):
    """Ground truth at the rendered resolution. The rendered pixel (i, j) looks through full-resolution pixel
    (linspace(0, H-1, H//l)[i], linspace(0, W-1, W//l)[j]) (poses.py:173-175), so sample the image there (bilinear)."""
    if level == 1:
        return img
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None]
    out = torch.nn.functional.interpolate(t, size=(H, W), mode="bilinear", align_corners=True)
    return out[0].permute(1, 2, 0).numpy()
