"""Measure the bf16 path against the fp32 oracle: stage errors, render PSNR, gradient agreement, step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import oracle.neus_oracle as orc
from vdn_train import synth, factory
from vdn_train.trainer import Trainer

dev = torch.device("cuda:0")
g = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev)
for variance in (0.3, 0.65):
    B, seed = 64, 5
    st = synth.make_all_states(seed, wdepth=True, variance=variance)
    nets = orc.nets_from_numpy(st, requires_grad=True)
    cams = synth.make_cameras(seed)
    px = np.floor(synth.uniform(seed, "p/x", (B,)) * 500) + 150
    py = np.floor(synth.uniform(seed, "p/y", (B,)) * 500) + 150
    o, d = synth.pixel_rays(cams[0], px, py)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, 0, B)
    tt = torch.tensor
    res = {}
    for prec in ("fp32", "bf16"):
        rend = factory.build_renderer(wdepth=True, device=dev, states=st, precision=prec)
        if prec == "fp32":
            with torch.no_grad():
                z, _ = rend._sample(g(o), g(d), g(near).reshape(-1), g(far).reshape(-1), 1.0, g(t1), g(t2), None)
            ref = orc.render(nets, tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3), cos_anneal_ratio=0.5,
                             t_rand=tt(t1), t_rand_out=tt(t2), z_vals_inject=z.cpu())
            target = tt(synth.target_colors(o, d))
            rl = (ref["color_fine"] - target).abs().sum() / B + 0.1 * ref["gradient_error"]
            named = orc.all_params(nets)
            gs = torch.autograd.grad(rl, [p for _, p in named], allow_unused=True)
            refg = torch.cat([(torch.zeros_like(p) if gr is None else gr).reshape(-1) for (_, p), gr in zip(named, gs)])
        pts = g((synth.uniform(seed, "p/pts", (4096, 3)) * 2 - 1) * 0.9)
        with torch.no_grad():
            out_sdf = rend.sdf_network(pts).cpu()
            nrm = rend.sdf_network.gradient(pts).cpu()[:, 0]
        oo, og = orc.sdf_forward(nets.sdf, pts.cpu(), nets.sdf_conf, with_gradient=True)
        oo, og = oo.detach(), og.detach()
        out = rend.render(g(o), g(d), g(near), g(far), background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=0.5,
                          t_rand=g(t1), t_rand_out=g(t2), z_vals_inject=z)
        loss = (out["color_fine"] - g(target)).abs().sum() / B + 0.1 * out["gradient_error"]
        loss.backward()
        gg = torch.cat([p.grad.reshape(-1) for p in rend._all_parameters()]).cpu()
        col = out["color_fine"].detach().cpu()
        mse = ((col - ref["color_fine"].detach()) ** 2).mean().item()
        print("variance %.2f %-5s | sdf abs err max %.2e mean %.2e | feat rel %.2e | normal abs max %.2e | color abs max %.2e PSNR-vs-oracle %.1f dB | "
              "weights abs max %.2e | loss %.6f (ref %.6f) | grad cos %.6f rel-l2 %.2e"
              % (variance, prec, (out_sdf[:, 0] - oo[:, 0]).abs().max(), (out_sdf[:, 0] - oo[:, 0]).abs().mean(),
                 (out_sdf[:, 1:] - oo[:, 1:]).abs().max() / oo[:, 1:].abs().max(), (nrm - og).abs().max(),
                 (col - ref["color_fine"].detach()).abs().max(), 10 * np.log10(1.0 / max(mse, 1e-20)),
                 (out["weights"].detach().cpu() - ref["weights"].detach()).abs().max(), loss.item(), rl.item(),
                 torch.nn.functional.cosine_similarity(gg, refg, dim=0).item(), ((gg - refg).norm() / refg.norm()).item()), flush=True)
# speed at B = 512
for prec in ("fp32", "bf16"):
    st = synth.make_all_states(0, wdepth=False)
    rend = factory.build_renderer(device=dev, states=st, precision=prec)
    tr = Trainer(rend, 512, dev)
    cams = synth.make_cameras(0)
    o, d = synth.random_pixel_batch(0, 0, 0, 512, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    args = (g(o), g(d), g(near), g(far), g(synth.target_colors(o, d)))
    for _ in range(3):
        tr.train_step(*args)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        sc = tr.train_step(*args)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    with torch.no_grad():
        for _ in range(2):
            rend.render(*args[:4], background_rgb=torch.ones(1, 3, device=dev))
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10):
            rend.render(*args[:4], background_rgb=torch.ones(1, 3, device=dev))
        torch.cuda.synchronize(); df = (time.time() - t0) / 10
    print("%s: train step %.3f ms (%.0f rays/s), forward render %.3f ms (%.0f rays/s), loss %.4f" % (prec, dt * 1e3, 512 / dt, df * 1e3, 512 / df, sc[0].item()), flush=True)
