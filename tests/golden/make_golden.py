#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (read-only at /root/reference) on CPU.

Runs only in the build container (the reference does not travel to the GPU box). It
  1. stubs the two absent third-party imports of dpt_models/renderer.py:6-7 (mcubes, icecream),
  2. builds the reference's own SDFNetwork / RenderingNetwork / NeRF / SingleVarianceNetwork /
     NeuSRenderer at the shipped config shapes and loads this repo's deterministic synthetic
     weights (vdn_train.synth) through load_state_dict,
  3. replaces torch.rand by a queue of injected jitter tensors for the duration of render(),
  4. writes inputs + reference outputs as small .npz fixtures (weights are NOT stored: they are
     regenerated from (seed, name) by vdn_train.synth),
  5. cross-checks the oracle (oracle/neus_oracle.py) against the reference and prints the errors.

Usage:  python tests/golden/make_golden.py [--check-only]
"""
import argparse
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")


REFERENCE = "/root/reference"


def import_reference():
    """Load the reference's three hot-path modules straight from their files, under the package name they import each
    other by (`dpt_models`, fields.py:5). The reference's dpt_models/ has no __init__.py (a namespace package), so a
    plain `import dpt_models.fields` resolves to THIS repo's regular package of the same name whenever it is on
    sys.path; loading by file path cannot be shadowed, and the asserts below prove where the code came from."""
    import importlib.util
    sys.modules.setdefault("mcubes", types.ModuleType("mcubes"))
    ic = types.ModuleType("icecream")
    ic.ic = lambda *a, **k: None
    sys.modules.setdefault("icecream", ic)
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "dpt_models" or k.startswith("dpt_models.")}
    pkg = types.ModuleType("dpt_models")
    pkg.__path__ = [os.path.join(REFERENCE, "dpt_models")]
    sys.modules["dpt_models"] = pkg
    mods = {}
    try:
        for name in ("embedder", "fields", "renderer"):
            path = os.path.join(REFERENCE, "dpt_models", name + ".py")
            spec = importlib.util.spec_from_file_location("dpt_models." + name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules["dpt_models." + name] = mod
            spec.loader.exec_module(mod)
            assert os.path.realpath(mod.__file__).startswith(REFERENCE + os.sep), mod.__file__
            mods[name] = mod
        assert mods["fields"].get_embedder is mods["embedder"].get_embedder
    finally:
        # the reference modules stay reachable through `mods` only; this repo's dpt_models (if it was imported) comes back
        for k in [k for k in sys.modules if k == "dpt_models" or k.startswith("dpt_models.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    return mods["fields"], mods["renderer"], mods["embedder"]


MODE_KW = {"idr": dict(mode="idr", d_in=9, multires_view=4), "no_normal": dict(mode="no_normal", d_in=6, multires_view=4),
           "no_view_dir": dict(mode="no_view_dir", d_in=6, multires_view=0)}


def build_reference(fields, renderer, states, wdepth, dtype, n_importance=64, n_outside=32, color_mode="idr", weight_norm=True,
                    depth_before_color=False):
    from vdn_train import synth
    tt = lambda d: {k: torch.tensor(v, dtype=dtype) for k, v in d.items()}
    vs = lambda key, mode: synth.variant_state(states[key], mode, weight_norm)
    nerf = fields.NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
                       rgb_dims=3, use_viewdirs=True, gen_depth_feats=wdepth, dpt_dim=96).to(dtype)
    sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5,
                            scale=1.0, geometric_init=True, weight_norm=weight_norm).to(dtype)
    var = fields.SingleVarianceNetwork(init_val=0.3).to(dtype)
    col = fields.RenderingNetwork(d_feature=352 if depth_before_color else 256, d_out=3, d_hidden=256, n_layers=4,
                                  weight_norm=weight_norm, squeeze_out=True, **MODE_KW[color_mode]).to(dtype)
    vdn = None
    if wdepth:
        vdn = fields.RenderingNetwork(d_feature=256, d_out=96, d_hidden=256, n_layers=4, weight_norm=weight_norm, squeeze_out=True,
                                      **MODE_KW[color_mode]).to(dtype)
        vdn.load_state_dict(tt(vs("depth_network_fine", color_mode)))
    nerf.load_state_dict(tt(states["nerf"]))
    sdf.load_state_dict(tt(vs("sdf_network_fine", "idr")))
    var.load_state_dict(tt(states["variance_network_fine"]))
    col.load_state_dict(tt(vs("color_network_fine", color_mode)))
    rend = renderer.NeuSRenderer(nerf, sdf, var, col, vdn, n_samples=64, n_importance=n_importance,
                                 n_outside=n_outside, up_sample_steps=4, perturb=1.0)
    return rend


class RandQueue:
    """Context manager: torch.rand(shape) pops the next injected tensor (renderer.py:348,355)."""

    def __init__(self, tensors):
        self.q = list(tensors)

    def __enter__(self):
        self.orig = torch.rand

        def fake(shape, *a, **k):
            t = self.q.pop(0)
            assert list(t.shape) == list(shape), (t.shape, shape)
            return t
        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self.orig


def make_rays(seed, B, center_crop=520):
    from vdn_train import synth
    cams = synth.make_cameras(seed)
    px = np.floor(synth.uniform(seed, "gold/x", (B,)) * center_crop) + (synth.W_IMG - center_crop) // 2
    py = np.floor(synth.uniform(seed, "gold/y", (B,)) * center_crop) + (synth.H - center_crop) // 2
    o, d = synth.pixel_rays(cams[seed % len(cams)], px, py)
    near, far = synth.near_far_from_sphere(o, d)
    return o, d, near, far


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def run_case(fields, renderer, name, seed, B, wdepth, variance, cos_anneal, perturb, dtype=torch.float32,
             n_importance=64, with_grads=True, white=True, depth_before_color=False, color_mode="idr", weight_norm=True):
    from vdn_train import synth
    import oracle.neus_oracle as orc
    torch.set_default_dtype(dtype)
    states = synth.make_all_states(seed, wdepth=wdepth, variance=variance, depth_before_color=depth_before_color)
    rend = build_reference(fields, renderer, states, wdepth, dtype, n_importance=n_importance, color_mode=color_mode, weight_norm=weight_norm,
                           depth_before_color=depth_before_color)
    o, d, near, far = make_rays(seed, B)
    t_rand, t_rand_out = synth.jitter(seed, 0, B)
    tt = lambda x: torch.tensor(x, dtype=dtype)
    bg_rgb = torch.ones(1, 3, dtype=dtype) if white else None
    q = [tt(t_rand), tt(t_rand_out)] if perturb > 0 else []
    captured = {}
    orig_core = rend.render_core

    def spy_core(rays_o_, rays_d_, z_vals_, *a, **k):      # records the reference's own inside z_vals
        captured["z"] = z_vals_.detach().clone()
        return orig_core(rays_o_, rays_d_, z_vals_, *a, **k)
    rend.render_core = spy_core
    with RandQueue(q):
        out = rend.render(tt(o), tt(d), tt(near), tt(far), perturb_overwrite=(-1 if perturb > 0 else 0),
                          background_rgb=bg_rgb, cos_anneal_ratio=cos_anneal,
                          depth_before_color=depth_before_color)
    true_rgb = tt(synth.target_colors(o, d))
    gt_feats = tt(synth.uniform(seed, "gold/feats", (B, 96)).astype(np.float32)) if wdepth else None
    # loss exactly as dpt_runner.py:208-243 (use_mask False, mask_weight 0, igr 0.1, depth ramp 0.7)
    mask = torch.ones(B, 1, dtype=dtype)
    mask_sum = mask.sum() + 1e-5
    color_err = (out["color_fine"] - true_rgb) * mask
    loss = torch.nn.functional.l1_loss(color_err, torch.zeros_like(color_err), reduction="sum") / mask_sum
    loss = loss + out["gradient_error"] * 0.1
    if wdepth:
        derr = (out["render_feats"] - gt_feats) * mask
        loss = loss + torch.nn.functional.l1_loss(derr, torch.zeros_like(derr), reduction="sum") / mask_sum * 0.7
    psnr = 20.0 * torch.log10(1.0 / (((out["color_fine"] - true_rgb) ** 2 * mask).sum() / (mask_sum * 3.0)).sqrt())
    fx = {
        "seed": seed, "B": B, "wdepth": wdepth, "variance": variance, "cos_anneal": cos_anneal,
        "perturb": perturb, "n_importance": n_importance, "white": white,
        "rays_o": o, "rays_d": d, "near": near, "far": far, "t_rand": t_rand, "t_rand_out": t_rand_out,
        "true_rgb": true_rgb.numpy(), "loss": loss.item(), "psnr": psnr.item(),
    }
    if wdepth:
        fx["gt_feats"] = gt_feats.numpy()
    for k, v in out.items():
        if v is not None:
            fx["out_" + k] = v.detach().numpy()
    if with_grads:
        mods = [("nerf", rend.nerf), ("sdf", rend.sdf_network), ("variance", rend.deviation_network),
                ("color", rend.color_network)] + ([("vdn", rend.depth_network)] if wdepth else [])
        params = [(mn + "." + pn if mn != "variance" else "variance", p) for mn, m in mods for pn, p in m.named_parameters()]
        grads = torch.autograd.grad(loss, [p for _, p in params], allow_unused=True)
        for (n, p), g in zip(params, grads):
            g = torch.zeros_like(p) if g is None else g
            gf = g.detach().reshape(-1)
            fx["grad_norm/" + n] = float(gf.norm())
            fx["grad_sum/" + n] = float(gf.sum())
            idx = np.unique(np.linspace(0, gf.numel() - 1, 16).astype(np.int64))
            fx["grad_idx/" + n] = idx
            fx["grad_val/" + n] = gf[idx].numpy()
    fx["z_vals_inside"] = captured["z"].numpy()          # the REFERENCE's inside z (input of render_core)
    if depth_before_color:
        fx["depth_before_color"] = True
    if color_mode != "idr" or not weight_norm:
        fx["color_mode"], fx["weight_norm"] = color_mode, weight_norm
        # constructor variants outside the shipped configurations: the oracle restates the shipped ones only; the fixture is
        # the reference's outputs and gradient samples
        torch.set_default_dtype(torch.float32)
        return fx, {}
    # ---- oracle cross-check
    nets = orc.nets_from_numpy(states, dtype=dtype, requires_grad=with_grads)
    conf = orc.RendererConf(n_importance=n_importance)
    rec = {}
    oo = orc.render(nets, tt(o), tt(d), tt(near), tt(far), conf, perturb_overwrite=(-1 if perturb > 0 else 0),
                    background_rgb=bg_rgb, cos_anneal_ratio=cos_anneal, depth_before_color=depth_before_color,
                    t_rand=tt(t_rand), t_rand_out=tt(t_rand_out), record=rec)
    errs = {k: rel(oo[k].detach().numpy(), out[k].detach().numpy()) for k in out if out[k] is not None}
    fx["z_vals_inside"] = captured["z"].numpy()          # the REFERENCE's inside z (input of render_core)
    errs["z_inside"] = rel(rec["z_vals_inside"].numpy(), fx["z_vals_inside"])
    oi = orc.render(nets, tt(o), tt(d), tt(near), tt(far), conf, perturb_overwrite=(-1 if perturb > 0 else 0),
                    background_rgb=bg_rgb, cos_anneal_ratio=cos_anneal, depth_before_color=depth_before_color,
                    t_rand=tt(t_rand), t_rand_out=tt(t_rand_out), z_vals_inject=captured["z"])
    for k in ("weights", "cdf_fine", "gradients", "color_fine"):
        errs["inj_" + k] = rel(oi[k].detach().numpy(), out[k].detach().numpy())
    if "coarse_sdf" in rec:
        fx["coarse_sdf"] = rec["coarse_sdf"].numpy()
        for i in range(4):
            fx["z_round%d" % i] = rec["z_round%d" % i].numpy()
    if with_grads:
        # gradients are compared with the reference's own z injected: the sampler is a no-grad,
        # ill-conditioned stage and would otherwise dominate the difference
        lo = orc.loss_from_render(oi, true_rgb, gt_feats=gt_feats, depth_ramp=0.7 if wdepth else None)
        errs["loss"] = abs(lo["loss"].item() - loss.item()) / abs(loss.item())
        named = orc.all_params(nets)
        og = torch.autograd.grad(lo["loss"], [p for _, p in named], allow_unused=True)
        worst = 0.0
        for (n, p), g in zip(named, og):
            g = torch.zeros_like(p) if g is None else g
            gf = g.detach().reshape(-1)
            ref_n = fx["grad_norm/" + n]
            e = float(np.abs(gf[fx["grad_idx/" + n]].numpy() - fx["grad_val/" + n]).max() / (np.abs(fx["grad_val/" + n]).max() + 1e-30))
            en = abs(float(gf.norm()) - ref_n) / (ref_n + 1e-30)
            worst = max(worst, e if ref_n > 0 else 0.0, en if ref_n > 0 else 0.0)
        errs["param_grads(worst)"] = worst
    print("[%s] oracle vs reference:" % name, {k: "%.2e" % v for k, v in errs.items()})
    torch.set_default_dtype(torch.float32)
    return fx, errs


def stage_fixture(fields, renderer, embedder, seed=3):
    """Per-stage known-answer vectors from the reference modules (fp32)."""
    from vdn_train import synth
    import oracle.neus_oracle as orc
    torch.set_default_dtype(torch.float32)
    states = synth.make_all_states(seed, wdepth=True, variance=0.3)
    rend = build_reference(fields, renderer, states, True, torch.float32)
    P = 96
    pts = torch.tensor(((synth.uniform(seed, "st/pts", (P, 3)) * 2 - 1) * 0.9).astype(np.float32))
    dirs = synth.normal(seed, "st/dirs", (P, 3))
    dirs = torch.tensor((dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)).astype(np.float32))
    fx = {"seed": seed, "pts": pts.numpy(), "dirs": dirs.numpy()}
    for (L, dd) in ((6, 3), (10, 4), (4, 3)):
        fn, od = embedder.get_embedder(L, input_dims=dd)
        x = pts if dd == 3 else torch.cat([pts, pts[:, :1] * 0.5], -1)
        fx["pe_%d_%d" % (L, dd)] = fn(x).numpy()
    out = rend.sdf_network(pts)
    fx["sdf_out"] = out.detach().numpy()
    g = rend.sdf_network.gradient(pts.clone()).squeeze(1)
    fx["sdf_grad"] = g.detach().numpy()
    feat = out[:, 1:].detach()
    fx["color"] = rend.color_network(pts, g.detach(), dirs, feat).detach().numpy()
    fx["vdn"] = rend.depth_network(pts, g.detach(), dirs, feat).detach().numpy()
    pts4 = torch.cat([pts, torch.tensor(synth.uniform(seed, "st/w", (P, 1)).astype(np.float32))], -1)
    a, rgb, ft = rend.nerf(pts4, dirs)
    fx["pts4"] = pts4.numpy()
    fx["nerf_alpha"], fx["nerf_rgb"], fx["nerf_feat"] = a.detach().numpy(), rgb.detach().numpy(), ft.detach().numpy()
    # sample_pdf known-answer
    B, M = 8, 80
    bins = np.sort(synth.uniform(seed, "st/bins", (B, M)) * 2 + 1, -1).astype(np.float32)
    w = (synth.uniform(seed, "st/w2", (B, M - 1)) ** 8).astype(np.float32)
    w[0, :] = 0.0                      # flat-CDF row: exercises the denom<1e-5 branch (renderer.py:70)
    w[1, 10:] = 0.0
    fx["spdf_bins"], fx["spdf_w"] = bins, w
    fx["spdf_out"] = renderer.sample_pdf(torch.tensor(bins), torch.tensor(w), 16, det=True).numpy()
    # SDF lattice (renderer.py:10-30) at resolution 20 (non multiple of 64 -> single ragged block)
    u = renderer.extract_fields(torch.tensor([-0.8, -0.7, -0.6]), torch.tensor([0.7, 0.8, 0.9]), 20,
                                lambda p: -rend.sdf_network.sdf(p))
    fx["lattice"] = u
    # oracle check
    nets = orc.nets_from_numpy(states)
    oo, og = orc.sdf_forward(nets.sdf, pts, nets.sdf_conf, with_gradient=True)
    e = {"sdf_out": rel(oo.numpy(), fx["sdf_out"]), "sdf_grad": rel(og.numpy(), fx["sdf_grad"]),
         "color": rel(orc.rendering_forward(nets.color, pts, g.detach(), dirs, feat, nets.color_conf).numpy(), fx["color"]),
         "vdn": rel(orc.rendering_forward(nets.vdn, pts, g.detach(), dirs, feat, nets.vdn_conf).numpy(), fx["vdn"]),
         "spdf": rel(orc.sample_pdf_det(torch.tensor(bins), torch.tensor(w), 16).numpy(), fx["spdf_out"]),
         "lattice": rel(orc.extract_fields(nets, [-0.8, -0.7, -0.6], [0.7, 0.8, 0.9], 20).numpy(), u)}
    na, nr, nf = orc.nerf_forward(nets.nerf, pts4, dirs, nets.nerf_conf)
    e["nerf"] = max(rel(na.numpy(), fx["nerf_alpha"]), rel(nr.numpy(), fx["nerf_rgb"]), rel(nf.numpy(), fx["nerf_feat"]))
    print("[stages] oracle vs reference:", {k: "%.2e" % v for k, v in e.items()})
    return fx, e


def adam_fixture(fields, renderer, seed=5, B=12, steps=3):
    """Params after 3 Adam steps of the a-R loop (dpt_runner.py:228-257, 310-319), fp32."""
    from vdn_train import synth
    torch.set_default_dtype(torch.float32)
    states = synth.make_all_states(seed, wdepth=False, variance=0.3)
    rend = build_reference(fields, renderer, states, False, torch.float32)
    mods = [rend.nerf, rend.sdf_network, rend.deviation_network, rend.color_network]
    params = [p for m in mods for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=5e-4)
    o, d, near, far = make_rays(seed, B)
    tt = torch.tensor
    true_rgb = tt(synth.target_colors(o, d))
    fx = {"seed": seed, "B": B, "steps": steps, "rays_o": o, "rays_d": d, "near": near, "far": far,
          "true_rgb": true_rgb.numpy()}
    losses = []
    for it in range(steps):
        lr_factor = (it + 100) / 5000.0            # warm-up branch of dpt_runner.py:311-312 at iter_step = it+100
        for gq in opt.param_groups:
            gq["lr"] = 5e-4 * lr_factor
        t1, t2 = synth.jitter(seed, it, B)
        with RandQueue([tt(t1), tt(t2)]):
            out = rend.render(tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3),
                              cos_anneal_ratio=min(1.0, (it + 100) / 50000.0))
        mask_sum = B + 1e-5
        loss = (out["color_fine"] - true_rgb).abs().sum() / mask_sum + out["gradient_error"] * 0.1
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    fx["losses"] = np.asarray(losses)
    names = ["nerf", "sdf", "variance", "color"]
    for mn, m in zip(names, mods):
        for pn, p in m.named_parameters():
            key = mn + "." + pn if mn != "variance" else "variance"
            pf = p.detach().reshape(-1)
            idx = np.unique(np.linspace(0, pf.numel() - 1, 8).astype(np.int64))
            fx["p_idx/" + key] = idx
            fx["p_val/" + key] = pf[idx].numpy()
            fx["p_norm/" + key] = float(pf.norm())
    print("[adam] reference losses:", losses)
    return fx


def raygrad_fixture(fields, renderer):
    """d loss / d (rays_o, rays_d, near, far) by the REFERENCE's autograd, the way a learnable-pose run differentiates
    (poses.py:198-208 hands over rays with a graph; dpt_runner.py:201-217 derives near / far from them and calls render):
    the four inputs are leaves here, so each partial is pinned separately. Two cases: 64 coarse samples only (every depth
    depends on near / far) and the full 64 + 64 sampler (the reference's own inside depths are stored for injection).
    Plus known answers of the so(3) helpers the pose modules use (lie_group_helper.py:47-83)."""
    from vdn_train import synth
    import importlib.util
    fx = {}
    for tag, seed, B, n_imp, cos_anneal in (("n64", 8, 16, 0, 0.3), ("full", 2, 16, 64, 0.5)):
        torch.set_default_dtype(torch.float32)
        states = synth.make_all_states(seed, wdepth=False, variance=0.3)
        rend = build_reference(fields, renderer, states, False, torch.float32, n_importance=n_imp)
        o, d, near, far = make_rays(seed, B)
        t_rand, t_rand_out = synth.jitter(seed, 0, B)
        tt = lambda x: torch.tensor(x, dtype=torch.float32)
        leaves = [tt(x).requires_grad_(True) for x in (o, d, near, far)]
        captured = {}
        orig_core = rend.render_core

        def spy_core(rays_o_, rays_d_, z_vals_, *a, **k):
            captured["z"] = z_vals_.detach().clone()
            return orig_core(rays_o_, rays_d_, z_vals_, *a, **k)
        rend.render_core = spy_core
        with RandQueue([tt(t_rand), tt(t_rand_out)]):
            out = rend.render(leaves[0], leaves[1], leaves[2], leaves[3], perturb_overwrite=-1,
                              background_rgb=torch.ones(1, 3), cos_anneal_ratio=cos_anneal)
        true_rgb = tt(synth.target_colors(o, d))
        mask_sum = B + 1e-5
        loss = (out["color_fine"] - true_rgb).abs().sum() / mask_sum + out["gradient_error"] * 0.1
        # (with importance sampling `near` is not part of the graph: the inside depths leave a no_grad block, renderer.py:367-386)
        grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        fx["%s/near_in_graph" % tag] = int(grads[2] is not None)
        grads = [torch.zeros_like(x) if gr is None else gr for x, gr in zip(leaves, grads)]
        for k, v in (("seed", seed), ("B", B), ("n_importance", n_imp), ("cos_anneal", cos_anneal), ("rays_o", o), ("rays_d", d),
                     ("near", near), ("far", far), ("t_rand", t_rand), ("t_rand_out", t_rand_out), ("true_rgb", true_rgb.numpy()),
                     ("loss", loss.item()), ("z_vals_inside", captured["z"].numpy()), ("color_fine", out["color_fine"].detach().numpy())):
            fx["%s/%s" % (tag, k)] = v
        for nm, gr in zip(("rays_o", "rays_d", "near", "far"), grads):
            fx["%s/grad_%s" % (tag, nm)] = gr.numpy()
        print("[raygrad %s] loss %.6f  |d rays_o| %.3e  |d rays_d| %.3e  |d near| %.3e  |d far| %.3e" %
              ((tag, loss.item()) + tuple(float(gr.abs().max()) for gr in grads)))
    spec = importlib.util.spec_from_file_location("ref_lie_group_helper", os.path.join(REFERENCE, "dpt_models", "lie_group_helper.py"))
    lg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lg)
    assert os.path.realpath(lg.__file__).startswith(REFERENCE + os.sep)
    rs = (synth.uniform(11, "pose/r", (6, 3)) * 2.0 - 1.0).astype(np.float32)
    rs[0] = 0.0                                     # the regularised zero rotation (|r| + 1e-15)
    rs[1] *= 1e-4
    ts = (synth.uniform(11, "pose/t", (6, 3)) * 2.0 - 1.0).astype(np.float32)
    fx["pose/r"], fx["pose/t"] = rs, ts
    fx["pose/c2w"] = np.stack([lg.make_c2w(torch.tensor(r), torch.tensor(t)).numpy() for r, t in zip(rs, ts)])
    return fx


def rays_fixture():
    """The reference's own ray source - RaysGenerator (poses.py:96-252) and Dataset.near_far_from_sphere (dataset.py:111-118) -
    run on CPU: `cv2` (absent here) is a stub whose imread() hands back injected arrays, the `.cuda()` at the end of
    gen_random_rays_at / gen_rays_between (poses.py:212, 248-249) is the identity while the generator runs, torch.randint is
    seeded and its draws are recorded. Both constructor branches (RGBA images composited on white, poses.py:117-122; 3-channel
    images with mask files, 123-127) and the wavelet-feature branch (global mean / std -> sigmoid -> bilinear up-sampling,
    133-146) run. load_K_Rt_from_P (dataset.py:13-34) needs cv.decomposeProjectionMatrix and stays out: parity unpinned there."""
    import importlib.util, tempfile
    from vdn_train import synth
    fx = {}
    store = {}
    cv = types.ModuleType("cv2")
    cv.imread = lambda name, flag=None: store[name].copy()
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "dpt_models" or k.startswith("dpt_models.") or k == "cv2"}
    sys.modules["cv2"] = cv
    pkg = types.ModuleType("dpt_models")
    pkg.__path__ = [os.path.join(REFERENCE, "dpt_models")]
    sys.modules["dpt_models"] = pkg
    mods = {}
    orig_cuda = torch.Tensor.cuda
    try:
        for name in ("lie_group_helper", "poses", "dataset"):
            path = os.path.join(REFERENCE, "dpt_models", name + ".py")
            spec = importlib.util.spec_from_file_location("dpt_models." + name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules["dpt_models." + name] = mod
            spec.loader.exec_module(mod)
            assert os.path.realpath(mod.__file__).startswith(REFERENCE + os.sep), mod.__file__
            mods[name] = mod
        poses, dataset = mods["poses"], mods["dataset"]
        torch.Tensor.cuda = lambda self, *a, **k: self
        n, H, W, C = 3, 40, 56, 5
        rng = np.random.RandomState(5)
        cams = synth.make_cameras(3, n=n).astype(np.float32)
        K = np.eye(4, dtype=np.float32)
        K[:3, :3] = np.linalg.inv(synth.intrinsics_inv(focal=60.0, h=H, w=W)).astype(np.float32)
        pose_all, intr_all = torch.tensor(cams), torch.tensor(np.stack([K] * n))
        fx["pose_all"], fx["intrinsics_all"] = cams, np.stack([K] * n)
        tmp = tempfile.mkdtemp(prefix="vdn_rays_fx_")
        # what cv.imread would hand back: 8-bit images (BGR / BGRA as they lie), 3-channel masks, [1, C, H/2, W/2] feature files
        bgr = rng.randint(0, 256, (n, H, W, 3)).astype(np.uint8)
        alpha = (rng.rand(n, H, W, 1) > 0.3).astype(np.uint8) * 255
        alpha[:, ::7] = 128                                  # some partial coverage
        bgra = np.concatenate([bgr, alpha], -1)
        msk = np.repeat((rng.rand(n, H, W, 1) > 0.4).astype(np.uint8) * 255, 3, -1)
        feats = rng.randn(n, 1, C, H // 2, W // 2).astype(np.float32) * 2.0 + 0.5
        fx["bgr"], fx["bgra"], fx["mask_files"], fx["feat_files"] = bgr, bgra, msk, feats
        img3, img4, mk, dp = [], [], [], []
        for i in range(n):
            for lis, arr, tag in ((img3, bgr[i], "rgb"), (img4, bgra[i], "rgba"), (mk, msk[i], "mask")):
                lis.append("%s/%s_%d.png" % (tmp, tag, i))
                store[lis[-1]] = arr
            dp.append("%s/feat_%d.npy" % (tmp, i))
            np.save(dp[-1], feats[i])
        for tag, imgs, with_depth in (("rgba", img4, True), ("rgbmask", img3, False)):
            gen = poses.RaysGenerator(imgs, mk, dp, pose_all, intr_all, learnable=False, with_depth=with_depth)
            fx[tag + "/images"], fx[tag + "/masks"] = gen.images.numpy(), gen.masks.numpy()
            if with_depth:
                fx[tag + "/depth_feats"] = gen.depth_feats.numpy()
            for idx, seed, B in ((1, 11, 64), (2, 12, 33)):
                torch.manual_seed(seed)
                data = gen.gen_random_rays_at(idx, B)
                torch.manual_seed(seed)                     # the two draws of poses.py:193-194, in order
                px = torch.randint(low=0, high=W, size=[B])
                py = torch.randint(low=0, high=H, size=[B])
                k = "%s/rand_%d" % (tag, idx)
                fx[k + "/img_idx"], fx[k + "/pixels_x"], fx[k + "/pixels_y"], fx[k + "/data"] = idx, px.numpy(), py.numpy(), data.numpy()
                near, far = dataset.Dataset.near_far_from_sphere(None, data[:, :3], data[:, 3:6])
                fx[k + "/near"], fx[k + "/far"] = near.numpy(), far.numpy()
        for idx, l in ((0, 1), (2, 2), (1, 4)):
            o, v = gen.gen_rays_at(idx, resolution_level=l)
            fx["at_%d_l%d/rays_o" % (idx, l)], fx["at_%d_l%d/rays_v" % (idx, l)] = o.contiguous().numpy(), v.contiguous().numpy()
        for ratio, i0, i1, l in ((0.0, 0, 2, 2), (0.3, 0, 2, 2), (0.75, 1, 2, 4), (1.0, 0, 1, 2)):
            o, v = gen.gen_rays_between(ratio, i0, i1, resolution_level=l)
            k = "between_%d_%d_r%03d_l%d" % (i0, i1, int(round(ratio * 100)), l)
            fx[k + "/ratio"], fx[k + "/rays_o"], fx[k + "/rays_v"] = ratio, o.contiguous().numpy(), v.contiguous().numpy()
        print("[rays] generator fixtures: %d keys; images %s, feats %s" % (len(fx), fx["rgba/images"].shape, fx["rgba/depth_feats"].shape))
    finally:
        torch.Tensor.cuda = orig_cuda
        for k in [k for k in sys.modules if k == "dpt_models" or k.startswith("dpt_models.") or k == "cv2"]:
            del sys.modules[k]
        sys.modules.update(saved)
    return fx


def pnf_fixture():
    """The reference-held DATA on this path: the learned camera parameters it ships, pretrained-models/*/*/pnf_300000.pth
    (schema: dpt_runner.py:383-401 - `pose_param_net` = LearnPose.state_dict() {init_c2w, r, t}, `intrin_net` = LearnIntrin {fx}).
    Each checkpoint is loaded into the REFERENCE's own LearnPose / LearnIntrin (poses.py:16-93), and its learnable ray branch
    (RaysGenerator.gen_random_rays_at with learnable=True, poses.py:189-212) runs on CPU with seeded, recorded pixels; the
    reference's autograd gives d loss / d (r, t) for a fixed linear loss on the rays. Stored: the checkpoint's parameters
    (a few hundred floats each: the reference does not travel), every camera's c2w and the intrinsics as the reference's modules
    return them, rays and pose gradients. The image size is not part of the checkpoint (the scenes are not shipped):
    H x W = 24 x 32 here - LearnIntrin's K is fx^2 W on the diagonal and (W/2, H/2) as principal point for any size."""
    import glob
    import importlib.util
    fx = {}
    store = {}
    cv = types.ModuleType("cv2")
    cv.imread = lambda name, flag=None: store[name].copy()
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "dpt_models" or k.startswith("dpt_models.") or k == "cv2"}
    sys.modules["cv2"] = cv
    pkg = types.ModuleType("dpt_models")
    pkg.__path__ = [os.path.join(REFERENCE, "dpt_models")]
    sys.modules["dpt_models"] = pkg
    mods = {}
    orig_cuda = torch.Tensor.cuda
    orig_to = torch.Tensor.to
    try:
        for name in ("lie_group_helper", "poses"):
            path = os.path.join(REFERENCE, "dpt_models", name + ".py")
            spec = importlib.util.spec_from_file_location("dpt_models." + name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules["dpt_models." + name] = mod
            spec.loader.exec_module(mod)
            assert os.path.realpath(mod.__file__).startswith(REFERENCE + os.sep), mod.__file__
            mods[name] = mod
        poses = mods["poses"]
        torch.Tensor.cuda = lambda self, *a, **k: self
        # LearnIntrin.forward moves its matrix to torch.device('cuda') (poses.py:54, 89): the identity while this runs
        torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], torch.device) and a[0].type == "cuda") else orig_to(self, *a, **k)
        H, W, B = 24, 32, 16
        files = sorted(glob.glob(os.path.join(REFERENCE, "pretrained-models", "*", "*", "pnf_300000.pth")))
        assert len(files) == 10, files
        names = []
        rng = np.random.RandomState(17)
        for f in files:
            tag = "%s.%s" % tuple(f.split(os.sep)[-3:-1])
            names.append(tag)
            ck = torch.load(f, map_location="cpu", weights_only=False)
            sd_pose, sd_intr = ck["pose_param_net"], ck["intrin_net"]
            n = sd_pose["r"].shape[0]
            pose_net = poses.LearnPose(n, True, True, torch.zeros(n, 4, 4))          # dpt_runner.py:76-84 builds it with the scene's poses
            pose_net.load_state_dict(sd_pose)
            intrin_net = poses.LearnIntrin(H, W, req_grad=True)
            intrin_net.load_state_dict(sd_intr)
            for k in ("init_c2w", "r", "t"):
                fx["%s/%s" % (tag, k)] = sd_pose[k].numpy()
            fx[tag + "/fx"] = sd_intr["fx"].numpy()
            fx[tag + "/poses_iter_step"] = int(ck["poses_iter_step"])
            with torch.no_grad():
                fx[tag + "/c2w"] = np.stack([pose_net(i).numpy() for i in range(n)])
                fx[tag + "/intrinsic"] = intrin_net().numpy()
            # images: what cv.imread would return (RGBA, 8 bit); only the gathers of colour / mask touch them
            imgs = []
            bgra = rng.randint(0, 256, (n, H, W, 4)).astype(np.uint8)
            for i in range(n):
                imgs.append("/pnf/%s/%03d.png" % (tag, i))
                store[imgs[-1]] = bgra[i]
            gen = poses.RaysGenerator(imgs, None, None, pose_net, intrin_net, learnable=True, with_depth=False)
            fx[tag + "/bgra"] = bgra[:3]                 # (the three cameras' images used below)
            cams = [0, 1, 2]
            for j, idx in enumerate(cams):
                # swap the tested camera's image into slot idx: only three images are stored
                seed = 100 + 7 * len(names) + j
                torch.manual_seed(seed)
                data = gen.gen_random_rays_at(idx, B)
                torch.manual_seed(seed)
                px = torch.randint(low=0, high=W, size=[B])
                py = torch.randint(low=0, high=H, size=[B])
                wgt = torch.tensor(rng.randn(B, 6).astype(np.float32))
                loss = (data[:, :6] * wgt).sum()
                gr, gt = torch.autograd.grad(loss, [pose_net.r, pose_net.t])
                k = "%s/cam%d" % (tag, idx)
                fx[k + "/pixels_x"], fx[k + "/pixels_y"], fx[k + "/data"] = px.numpy(), py.numpy(), data.detach().numpy()
                fx[k + "/loss_weights"], fx[k + "/loss"] = wgt.numpy(), float(loss)
                fx[k + "/grad_r"], fx[k + "/grad_t"] = gr[idx].numpy(), gt[idx].numpy()
                assert all(float(g[i].abs().sum()) == 0.0 for g in (gr, gt) for i in range(n) if i != idx)       # only that camera's row
        fx["names"] = np.array(names)
        fx["H"], fx["W"] = H, W
        print("[pnf] %d checkpoints, %d keys; cameras per scene: %s" % (len(names), len(fx), [fx[n + "/r"].shape[0] for n in names]))
    finally:
        torch.Tensor.cuda = orig_cuda
        torch.Tensor.to = orig_to
        for k in [k for k in sys.modules if k == "dpt_models" or k.startswith("dpt_models.") or k == "cv2"]:
            del sys.modules[k]
        sys.modules.update(saved)
    return fx


def runner_fixture():
    """The reference RUNNER's own image loops - Runner.val_img with gen_depth_for_finetune=True (dpt_runner.py:417-491: batched render of
    one camera, L1 / PSNR against the image, the weight-argmax depth written as depth_from_sdf/sdf_<name>.npy and as the
    weight_max PNG) and Runner.validate_image (520-587: colour and normal images as written by cv.imwrite) - run by the reference
    itself on CPU. dpt_runner.py is loaded by file path with its absent third-party imports stubbed (cv2: imread hands back the
    injected arrays, imwrite records what it is given, resize is the identity at resolution level 1 and raises otherwise; trimesh,
    pyhocon, tensorboard, icecream: empty), `dpt_models.*` bound to the reference's own modules; the Runner object is made
    without its constructor (which parses a conf file and builds a dataset from disk) and given exactly the attributes the two
    methods read. torch.rand inside render() pops injected jitter (one pair per batch)."""
    import importlib.util
    import tempfile
    from types import SimpleNamespace
    from vdn_train import synth
    fx = {}
    store, written = {}, {}
    cv = types.ModuleType("cv2")
    cv.imread = lambda name, flag=None: store[name].copy()

    def resize(img, size, *a, **k):
        if (img.shape[1], img.shape[0]) != tuple(size):
            raise NotImplementedError("cv.resize stub: resolution level 1 only")
        return img.reshape(img.shape[0], img.shape[1]) if (img.ndim == 3 and img.shape[2] == 1) else img       # cv drops a single channel
    cv.resize = resize
    cv.imwrite = lambda path, arr: written.__setitem__(path, np.array(arr)) or True
    stubs = {"cv2": cv, "trimesh": types.ModuleType("trimesh"), "pyhocon": types.ModuleType("pyhocon"),
             "torch.utils.tensorboard": types.ModuleType("torch.utils.tensorboard")}
    stubs["pyhocon"].ConfigFactory = object
    stubs["torch.utils.tensorboard"].SummaryWriter = object
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "dpt_models" or k.startswith("dpt_models.") or k in stubs}
    sys.modules.update(stubs)
    sys.modules.setdefault("mcubes", types.ModuleType("mcubes"))
    ic = types.ModuleType("icecream")
    ic.ic = lambda *a, **k: None
    sys.modules.setdefault("icecream", ic)
    pkg = types.ModuleType("dpt_models")
    pkg.__path__ = [os.path.join(REFERENCE, "dpt_models")]
    sys.modules["dpt_models"] = pkg
    orig_cuda = torch.Tensor.cuda
    try:
        mods = {}
        for name in ("embedder", "lie_group_helper", "fields", "renderer", "dataset", "poses"):
            path = os.path.join(REFERENCE, "dpt_models", name + ".py")
            spec = importlib.util.spec_from_file_location("dpt_models." + name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules["dpt_models." + name] = mod
            spec.loader.exec_module(mod)
            assert os.path.realpath(mod.__file__).startswith(REFERENCE + os.sep), mod.__file__
            mods[name] = mod
        spec = importlib.util.spec_from_file_location("ref_dpt_runner", os.path.join(REFERENCE, "dpt_runner.py"))
        rn = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rn)
        assert os.path.realpath(rn.__file__).startswith(REFERENCE + os.sep)
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.set_default_dtype(torch.float32)
        seed, n, H, W, BS = 13, 2, 10, 12, 50
        states = synth.make_all_states(seed, wdepth=False, variance=0.45)
        rend = build_reference(mods["fields"], mods["renderer"], states, False, torch.float32)
        rng = np.random.RandomState(seed)
        cams = synth.make_cameras(seed, n=n).astype(np.float32)
        K = np.eye(4, dtype=np.float32)
        K[:3, :3] = np.linalg.inv(synth.intrinsics_inv(focal=1.39 * W, h=H, w=W)).astype(np.float32)
        bgra = rng.randint(0, 256, (n, H, W, 4)).astype(np.uint8)
        bgra[..., 3] = (rng.rand(n, H, W) > 0.25) * 255
        tmp = tempfile.mkdtemp(prefix="vdn_runner_fx_")
        imgs = []
        for i in range(n):
            imgs.append("%s/%03d.png" % (tmp, i))
            store[imgs[-1]] = bgra[i]
        gen = mods["poses"].RaysGenerator(imgs, None, None, torch.tensor(cams), torch.tensor(np.stack([K] * n)), learnable=False, with_depth=False)
        runner = object.__new__(rn.Runner)
        runner.rays_generator, runner.renderer = gen, rend
        runner.dataset = SimpleNamespace(near_far_from_sphere=lambda o, d: mods["dataset"].Dataset.near_far_from_sphere(None, o, d),
                                         data_dir=tmp, img_dir="image", pose_all=torch.tensor(cams), n_images=n)
        runner.batch_size, runner.use_mask, runner.use_white_bkgd, runner.depth_before_color = BS, False, True, False
        runner.base_exp_dir, runner.iter_step, runner.anneal_end, runner.rgb_dims, runner.validate_resolution_level = tmp, 20000, 50000.0, 3, 1
        idx = 1
        n_batches = (H * W + BS - 1) // BS
        sizes = [min(BS, H * W - b * BS) for b in range(n_batches)]
        jit = [synth.jitter(seed, 100 + b, sizes[b]) for b in range(n_batches)]
        tt = lambda x: torch.tensor(x, dtype=torch.float32)
        queue = lambda: [t for a, b in jit for t in (tt(a), tt(b))]
        with RandQueue(queue()):
            closs, psnr, eik, _, _ = rn.Runner.val_img(runner, idx, resolution_level=1, gen_depth_for_finetune=True)
        depth_file = os.path.join(tmp, "image", "depth_from_sdf", "sdf_%03d.npy" % idx)
        fx["val_img/weight_depth"] = np.load(depth_file)
        (wm_path, wm), = [(k, v) for k, v in written.items() if "weight_max" in k]
        fx["val_img/weight_max_png"], fx["val_img/weight_max_name"] = wm, os.path.relpath(wm_path, tmp)
        fx["val_img/color_fine_loss"], fx["val_img/psnr"] = float(closs), float(psnr)
        fx["val_img/gradient_error"] = np.stack([np.asarray(e) for e in eik])
        written.clear()
        with RandQueue(queue()):
            rn.Runner.validate_image(runner, idx, resolution_level=1)
        for k, v in written.items():
            sub = os.path.relpath(k, tmp).split(os.sep)[0]
            fx["validate_image/%s" % sub], fx["validate_image/%s_name" % sub] = v, os.path.relpath(k, tmp)
        assert {"validate_image/validations_fine", "validate_image/normals"} <= set(fx)
        for k, v in (("seed", seed), ("variance", 0.45), ("idx", idx), ("H", H), ("W", W), ("batch_size", BS), ("iter_step", 20000), ("anneal_end", 50000.0),
                     ("bgra", bgra), ("pose_all", cams), ("intrinsic", K), ("images", gen.images.numpy()), ("masks", gen.masks.numpy())):
            fx[k] = v
        for b, (a, c) in enumerate(jit):
            fx["jitter/%d/t_rand" % b], fx["jitter/%d/t_rand_out" % b] = a, c
        print("[runner] val_img: L1 %.5f PSNR %.3f; weight_depth %s; files %s" % (closs, psnr, fx["val_img/weight_depth"].shape,
              [fx[k] for k in fx if k.endswith("_name")]))
    finally:
        torch.Tensor.cuda = orig_cuda
        for k in [k for k in sys.modules if k == "dpt_models" or k.startswith("dpt_models.") or k in stubs]:
            del sys.modules[k]
        sys.modules.update(saved)
    return fx


CASES = [
    # name, seed, B, wdepth, variance, cos_anneal, perturb, kwargs
    ("white_v03_c0", 1, 24, False, 0.3, 0.0, 1.0, {}),
    ("white_v03_c05_det", 2, 24, False, 0.3, 0.5, 0.0, {}),
    ("white_v065_c1", 4, 24, False, 0.65, 1.0, 1.0, {}),
    ("wdepth_v03_c05", 6, 16, True, 0.3, 0.5, 1.0, {}),
    ("wdepth_v065_c1", 7, 16, True, 0.65, 1.0, 1.0, {}),
    ("white_n64_v03", 8, 24, False, 0.3, 0.3, 1.0, {"n_importance": 0}),
    ("black_v03", 9, 16, False, 0.3, 1.0, 1.0, {"white": False, "with_grads": False}),
    # constructor variants no shipped configuration uses (fields.py:154-158, 65-66 / 141-142)
    ("white_nonormal_plain", 10, 16, False, 0.3, 0.5, 1.0, {"color_mode": "no_normal", "weight_norm": False}),
    ("wdepth_noviewdir", 11, 12, True, 0.3, 0.5, 1.0, {"color_mode": "no_view_dir"}),
    # renderer.py:247-248: the colour network fed cat([feature_vector, VDN output]) (no shipped configuration sets it)
    ("wdepth_dbc", 12, 12, True, 0.3, 0.5, 1.0, {"depth_before_color": True}),
]
F64_COMPANIONS = ("white_v03_c0", "white_v065_c1", "wdepth_v065_c1")


def generate(only=None):
    """Run the reference and return {fixture name: {key: array}} (all fixtures, or the names in `only`)."""
    fields, renderer, embedder = import_reference()
    want = (lambda n: True) if only is None else (lambda n: n in only)
    torch.manual_seed(0)
    out = {}
    if want("stages"):
        out["stages"], _ = stage_fixture(fields, renderer, embedder)
    for (name, seed, B, wd, var, ca, pt, kw) in CASES:
        if want(name):
            out[name], _ = run_case(fields, renderer, name, seed, B, wd, var, ca, pt, **kw)
        if name in F64_COMPANIONS and want(name + "_f64"):
            # fp64 companion: calibrates tolerances (SURVEY.md 4 noise floor)
            fx64, _ = run_case(fields, renderer, name + "_f64", seed, B, wd, var, ca, pt, dtype=torch.float64, **kw)
            keep = {k: v for k, v in fx64.items() if k.startswith("out_") or k.startswith("grad_") or
                    k in ("loss", "psnr", "z_vals_inside")}
            out[name + "_f64"] = keep
    if want("adam3"):
        out["adam3"] = adam_fixture(fields, renderer)
    if want("raygrad"):
        out["raygrad"] = raygrad_fixture(fields, renderer)
    if want("rays"):
        out["rays"] = rays_fixture()
    if want("pnf_rays"):
        out["pnf_rays"] = pnf_fixture()
    if want("runner"):
        out["runner"] = runner_fixture()
    return out


def compare_with_committed(out):
    """Bit-for-bit comparison of freshly generated fixtures with the committed .npz files -> list of mismatches."""
    bad = []
    for name, fx in out.items():
        path = os.path.join(HERE, name + ".npz")
        if not os.path.exists(path):
            bad.append((name, "<file missing>"))
            continue
        have = dict(np.load(path, allow_pickle=False))
        new = {k.replace("/", "__"): np.asarray(v) for k, v in fx.items()}
        if set(have) != set(new):
            bad.append((name, "key sets differ: %s" % sorted(set(have) ^ set(new))[:5]))
            continue
        for k in sorted(new):
            a, b = have[k], new[k]
            # NaN-aware equality only exists for inexact dtypes (string / integer keys such as `color_mode` raise in isnan)
            same = a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=np.issubdtype(a.dtype, np.inexact))
            if not same:
                bad.append((name, k))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true",
                    help="regenerate from the reference, compare with the committed fixtures bit for bit, write nothing")
    ap.add_argument("--only", default=None, help="comma-separated fixture names (default: all)")
    args = ap.parse_args()
    out = generate(None if args.only is None else set(args.only.split(",")))
    if args.check_only:
        bad = compare_with_committed(out)
        for name, key in bad:
            print("MISMATCH %s: %s" % (name, key))
        print("checked %d fixtures against the reference: %s" % (len(out), "all bit-identical" if not bad else "%d differences" % len(bad)))
        sys.exit(1 if bad else 0)
    for name, fx in out.items():
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **{k.replace("/", "__"): np.asarray(v) for k, v in fx.items()})
        print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
