"""Whole-image rendering through the HIP renderer: Runner.validate_image (dpt_runner.py:520-587: colour + normal image),
Runner.render_novel_image (589-616: interpolated view) and the arithmetic of Runner.val_img (dpt_runner.py:417-491) - rays of
one camera in batches, colour image, L1 / PSNR against the ground truth, and the weight-argmax depth written back as
`depth_from_sdf/sdf_<name>.npy` for the wavelet fine-tuning loop (dpt_runner.py:449-453). Everything stays on the device
until the final image copy (the reference copies every batch to the host)."""
import os

import numpy as np
import torch


def _render_batches(renderer, rays_gen, rays_o, rays_d, batch_size, jitter=None, **kw):
    """render() over the rays of one image in batches -> yields (first ray, n rays, outputs). With VDN_RENDER_GRAPH=1 the
    full batches replay one captured plan (dpt_models.renderer.RenderPlan: a fifth of the host time per batch, ~3% more device
    time) and only the ragged last batch is a plain call. The outputs of a replay are overwritten by the next one: consume
    them before advancing. jitter: optional list of (t_rand [n,1], t_rand_out [n,O]) per batch, injected in place of the two
    torch.rand draws of render() (renderer.py:348,355) - how the parity tests reproduce the reference's runs."""
    n = rays_o.shape[0]
    plan = None
    if os.environ.get("VDN_RENDER_GRAPH", "0") == "1" and n >= 2 * batch_size and jitter is None:
        plan = renderer.plan(batch_size, **kw)
    for b, s in enumerate(range(0, n, batch_size)):
        o, d = rays_o[s:s + batch_size], rays_d[s:s + batch_size]
        near, far = rays_gen.near_far_from_sphere(o, d)
        if plan is not None and o.shape[0] == batch_size:
            yield s, batch_size, plan(o, d, near, far)
        elif jitter is not None:
            yield s, o.shape[0], renderer.render(o, d, near, far, t_rand=jitter[b][0], t_rand_out=jitter[b][1], **kw)
        else:
            yield s, o.shape[0], renderer.render(o, d, near, far, **kw)


@torch.no_grad()
def render_image(renderer, rays_gen, idx, resolution_level=1, batch_size=512, cos_anneal_ratio=1.0, white_bkgd=True,
                 gen_depth_for_finetune=False, jitter=None):
    """-> dict(img_fine [H,W,3] float32 in [0,1], gradient_error [n_batches], weight_depth [H,W,1] | None)."""
    rays_o, rays_d = rays_gen.gen_rays_at(idx, resolution_level=resolution_level)
    H, W, _ = rays_o.shape
    rays_o, rays_d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
    dev = rays_o.device
    bg = torch.ones(1, 3, device=dev) if white_bkgd else None
    rgb = torch.empty(H * W, 3, device=dev)
    depth = torch.empty(H * W, 1, device=dev) if gen_depth_for_finetune else None
    eik = []
    for s, n, out in _render_batches(renderer, rays_gen, rays_o, rays_d, batch_size, jitter=jitter, cos_anneal_ratio=cos_anneal_ratio, background_rgb=bg):
        rgb[s:s + n] = out["color_fine"]
        eik.append(out["gradient_error"].clone())
        if gen_depth_for_finetune:                                   # dpt_runner.py:449-455
            inside = out["inside_sphere"]
            w = out["weights"][:, :inside.shape[1]] * inside
            depth[s:s + n] = out["z_vals"].gather(1, torch.argmax(w, dim=-1, keepdim=True))
        del out
    return {"img_fine": rgb.reshape(H, W, 3).cpu().numpy(), "gradient_error": torch.stack(eik).cpu().numpy(),
            "weight_depth": depth.reshape(H, W, 1).cpu().numpy() if gen_depth_for_finetune else None}


@torch.no_grad()
def validate_image(renderer, rays_gen, idx, resolution_level=1, batch_size=512, cos_anneal_ratio=1.0, white_bkgd=True,
                   depth_before_color=False, out_dir=None, iter_step=0, jitter=None):
    """Runner.validate_image (dpt_runner.py:520-587): the colour image and the normal image of one camera.
    -> (img_fine [H,W,3] in 0..255, normal_img [H,W,3] in 0..255): normals = sum_i gradients_i * weights_i * inside_sphere_i
    per ray (553-557), rotated into the camera frame by inv(pose[:3,:3]) and mapped by * 128 + 128 (570-573). With `out_dir`
    the two PNGs are written as the runner writes them (575-587): validations_fine/ holds the render stacked over the
    ground-truth image (`image_at`), normals/ the normal image. The pipeline's channel order is cv.imread's BGR
    (dataset._read_png) and cv.imwrite takes its array as BGR, so the files show true colours: PIL gets the array reversed."""
    rays_o, rays_d = rays_gen.gen_rays_at(idx, resolution_level=resolution_level)
    H, W, _ = rays_o.shape
    rays_o, rays_d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
    dev = rays_o.device
    bg = torch.ones(1, 3, device=dev) if white_bkgd else None
    rgb = torch.empty(H * W, 3, device=dev)
    nrm = torch.empty(H * W, 3, device=dev)
    n_in = renderer.n_samples + renderer.n_importance
    for s, n, out in _render_batches(renderer, rays_gen, rays_o, rays_d, batch_size, jitter=jitter, cos_anneal_ratio=cos_anneal_ratio,
                                     background_rgb=bg, depth_before_color=depth_before_color):
        rgb[s:s + n] = out["color_fine"]
        nrm[s:s + n] = (out["gradients"] * out["weights"][:, :n_in, None] * out["inside_sphere"][..., None]).sum(dim=1)
        del out
    img_fine = (rgb.reshape(H, W, 3).cpu().numpy() * 255).clip(0, 255)
    rot = np.linalg.inv(rays_gen.pose_all[idx, :3, :3].detach().cpu().numpy())
    normal_img = (np.matmul(rot[None, :, :], nrm.cpu().numpy()[:, :, None]).reshape(H, W, 3) * 128 + 128).clip(0, 255)
    if out_dir is not None:
        from PIL import Image
        val_im = np.concatenate([img_fine, rays_gen.image_at(idx, resolution_level=resolution_level)])      # dpt_runner.py:577-578
        for sub, arr in (("validations_fine", val_im), ("normals", normal_img)):
            os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
            # cv.imwrite(path, float array) = the array read as BGR, converted with saturate_cast<uchar> (round to nearest):
            # hand PIL the RGB view of that
            Image.fromarray(np.ascontiguousarray(np.rint(arr).clip(0, 255).astype(np.uint8)[..., ::-1])).save(
                os.path.join(out_dir, sub, "{:0>8d}_{}_{}.png".format(iter_step, 0, idx)))
    return img_fine, normal_img


@torch.no_grad()
def render_novel_image(renderer, rays_gen, idx_0, idx_1, ratio, resolution_level=1, batch_size=512, cos_anneal_ratio=1.0,
                       white_bkgd=True, depth_before_color=False):
    """Runner.render_novel_image (dpt_runner.py:589-616): the view interpolated between two cameras -> uint8 [H,W,3]
    (the runner's `* 256` clip). Argument order as RaysGenerator.gen_rays_between defines it (ratio first, poses.py:214);
    the runner itself calls it with (idx_0, idx_1, ratio), SURVEY.md Appendix A."""
    rays_o, rays_d = rays_gen.gen_rays_between(ratio, idx_0, idx_1, resolution_level=resolution_level)
    H, W, _ = rays_o.shape
    rays_o, rays_d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
    dev = rays_o.device
    bg = torch.ones(1, 3, device=dev) if white_bkgd else None
    rgb = torch.empty(H * W, 3, device=dev)
    for s, n, out in _render_batches(renderer, rays_gen, rays_o, rays_d, batch_size, cos_anneal_ratio=cos_anneal_ratio,
                                     background_rgb=bg, depth_before_color=depth_before_color):
        rgb[s:s + n] = out["color_fine"]
    return (rgb.reshape(H, W, 3).cpu().numpy() * 256).clip(0, 255).astype(np.uint8)


def image_metrics(img_fine, gt, mask=None):
    """color L1 and PSNR of dpt_runner.py:470-474 (mask = ones when masks are not used)."""
    mask = np.ones_like(gt[..., :1]) if mask is None else (mask > 0.1).astype(np.float32)
    mask_sum = mask.sum() + 1e-5
    l1 = np.abs((img_fine - gt) * mask).sum() / mask_sum
    psnr = 20.0 * np.log10(1.0 / np.sqrt(((img_fine - gt) ** 2 * mask).sum() / (mask_sum * 3.0)))
    return float(l1), float(psnr)


def weight_max_image(weight_depth):
    """The runner's picture of the weight-argmax depths (dpt_runner.py:463-468): stretched between their 50th and 95th percentile."""
    lb, ub = np.percentile(weight_depth, [50, 95])
    return ((weight_depth - lb) / (ub - lb) * 255).clip(0, 255)


def val_img(renderer, scene, rays_gen, idx, resolution_level=1, batch_size=512, cos_anneal_ratio=1.0, white_bkgd=True,
            use_mask=False, gen_depth_for_finetune=False, jitter=None, out_dir=None, iter_step=0):
    """Runner.val_img (dpt_runner.py:417-491) for a vdn_train.dataset.SceneData (or None: nothing is written into a scene
    directory then): -> (color_loss, psnr, gradient_error, img_fine). The ground truth is the generator's `image_at` / `mask_at`
    (poses.py:254-262: cv.resize of the loaded image) as in the reference; with gen_depth_for_finetune the weight-argmax depths
    go to depth_from_sdf/sdf_<name>.npy under the scene directory (449-453) and, with `out_dir`, their picture to
    weight_max/weight_max_<iter>_<idx>.png (463-468)."""
    res = render_image(renderer, rays_gen, idx, resolution_level, batch_size, cos_anneal_ratio, white_bkgd, gen_depth_for_finetune, jitter=jitter)
    gt = rays_gen.image_at(idx, resolution_level=resolution_level) / 255.0
    mask = rays_gen.mask_at(idx, resolution_level=resolution_level) if use_mask else None
    l1, psnr = image_metrics(res["img_fine"], gt, mask)
    if gen_depth_for_finetune:
        if scene is not None:
            path = scene.depth_from_sdf_path(idx)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            np.save(path, res["weight_depth"])
        if out_dir is not None:
            from PIL import Image
            os.makedirs(os.path.join(out_dir, "weight_max"), exist_ok=True)
            Image.fromarray(np.rint(weight_max_image(res["weight_depth"])[..., 0]).astype(np.uint8)).save(
                os.path.join(out_dir, "weight_max", "weight_max_{}_{}.png".format(iter_step, idx)))
    return l1, psnr, res["gradient_error"], res["img_fine"]
