"""The oracle (oracle/neus_oracle.py) against the golden vectors produced by the reference itself
(tests/golden/make_golden.py). CPU only. Tolerances: 1e-4 rel on per-ray outputs end to end and on
per-sample tensors with the reference's z injected; 1e-5 on single stages."""
import numpy as np
import pytest
import torch

import oracle.neus_oracle as orc
from vdn_train import synth
from conftest import relmax

CASES = ["white_v03_c0", "white_v03_c05_det", "white_v065_c1", "wdepth_v03_c05", "wdepth_v065_c1",
         "white_n64_v03", "black_v03"]


def _run(fx, dtype=torch.float32, inject=False, grads=False):
    st = synth.make_all_states(int(fx["seed"]), wdepth=bool(fx["wdepth"]), variance=float(fx["variance"]))
    nets = orc.nets_from_numpy(st, dtype=dtype, requires_grad=grads)
    tt = lambda x: torch.tensor(x, dtype=dtype)
    conf = orc.RendererConf(n_importance=int(fx["n_importance"]))
    out = orc.render(nets, tt(fx["rays_o"]), tt(fx["rays_d"]), tt(fx["near"]), tt(fx["far"]), conf,
                     perturb_overwrite=(-1 if fx["perturb"] > 0 else 0),
                     background_rgb=torch.ones(1, 3, dtype=dtype) if fx["white"] else None,
                     cos_anneal_ratio=float(fx["cos_anneal"]), t_rand=tt(fx["t_rand"]), t_rand_out=tt(fx["t_rand_out"]),
                     z_vals_inject=tt(fx["z_vals_inside"]) if (inject and fx["n_importance"] > 0) else None)
    return nets, out


@pytest.mark.parametrize("name", CASES)
def test_render_per_ray(golden, name):
    fx = golden(name)
    _, out = _run(fx)
    keys = ["color_fine", "weight_sum", "s_val", "z_vals", "gradient_error", "inside_sphere"]
    if fx["wdepth"]:
        keys.append("render_feats")
    for k in keys:
        assert relmax(out[k].detach().numpy(), fx["out_" + k]) < 1e-4, k
    assert set(k for k in out if not k.startswith("eik_")) == set(
        ["render_feats", "color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "gradients", "weights",
         "z_vals", "gradient_error", "inside_sphere"])           # renderer.py:426-439
    for k in ("weights", "cdf_fine", "gradients", "z_vals", "inside_sphere"):
        assert tuple(out[k].shape) == fx["out_" + k].shape


@pytest.mark.parametrize("name", CASES)
def test_render_per_sample_injected_z(golden, name):
    fx = golden(name)
    _, out = _run(fx, inject=True)
    for k in ("weights", "cdf_fine", "gradients", "weight_max", "color_fine"):
        assert relmax(out[k].detach().numpy(), fx["out_" + k]) < 1e-4, k


@pytest.mark.parametrize("name", ["white_v03_c0", "white_v03_c05_det", "wdepth_v03_c05", "white_n64_v03"])
def test_loss_and_param_grads(golden, name):
    fx = golden(name)
    nets, out = _run(fx, inject=True, grads=True)
    tt = torch.tensor
    lo = orc.loss_from_render(out, tt(fx["true_rgb"]), gt_feats=tt(fx["gt_feats"]) if fx["wdepth"] else None,
                              depth_ramp=0.7 if fx["wdepth"] else None)
    assert abs(lo["loss"].item() - float(fx["loss"])) < 1e-5 * abs(float(fx["loss"]))
    assert abs(lo["psnr"].item() - float(fx["psnr"])) < 1e-3
    named = orc.all_params(nets)
    gs = torch.autograd.grad(lo["loss"], [p for _, p in named], allow_unused=True)
    for (n, p), g in zip(named, gs):
        ref_norm = float(fx["grad_norm/" + n])
        g = torch.zeros_like(p) if g is None else g
        if ref_norm == 0:
            assert float(g.norm()) == 0
            continue
        gv = g.reshape(-1)[fx["grad_idx/" + n]].numpy()
        rv = fx["grad_val/" + n]
        # rel-to-max over the sampled entries; 3e-4 = fp32-vs-fp32 floor between two orderings of the same math
        assert np.abs(gv - rv).max() <= 3e-4 * np.abs(rv).max() + 1e-9, n
        assert abs(float(g.norm()) - ref_norm) <= 1e-4 * ref_norm, n


def test_fp64_oracle_matches_fp64_reference(golden):
    """In fp64 the restatement and the reference agree to rounding: the algorithm is the same one."""
    for name in ("white_v03_c0", "white_v065_c1", "wdepth_v065_c1"):
        fx, fx64 = golden(name), golden(name + "_f64")
        _, out = _run(fx, dtype=torch.float64)
        for k in ("color_fine", "weights", "gradients", "cdf_fine", "weight_sum"):
            assert relmax(out[k].detach().numpy(), fx64["out_" + k]) < 1e-9, (name, k)


def test_stage_vectors(golden):
    fx = golden("stages")
    st = synth.make_all_states(int(fx["seed"]), wdepth=True, variance=0.3)
    nets = orc.nets_from_numpy(st)
    pts, dirs = torch.tensor(fx["pts"]), torch.tensor(fx["dirs"])
    for (L, dd) in ((6, 3), (10, 4), (4, 3)):
        x = pts if dd == 3 else torch.cat([pts, pts[:, :1] * 0.5], -1)
        pe = orc.embed(x, L)
        assert pe.shape[1] == dd * (1 + 2 * L)
        assert np.array_equal(pe.numpy(), fx["pe_%d_%d" % (L, dd)])
    out, g = orc.sdf_forward(nets.sdf, pts, nets.sdf_conf, with_gradient=True)
    assert relmax(out.numpy(), fx["sdf_out"]) < 1e-5
    assert relmax(g.numpy(), fx["sdf_grad"]) < 1e-5
    assert relmax(orc.sdf_only(nets.sdf, pts, nets.sdf_conf).numpy(), fx["sdf_out"][:, :1]) < 1e-5
    feat, grad = torch.tensor(fx["sdf_out"][:, 1:]), torch.tensor(fx["sdf_grad"])
    assert relmax(orc.rendering_forward(nets.color, pts, grad, dirs, feat, nets.color_conf).numpy(), fx["color"]) < 1e-5
    assert relmax(orc.rendering_forward(nets.vdn, pts, grad, dirs, feat, nets.vdn_conf).numpy(), fx["vdn"]) < 1e-5
    a, rgb, ft = orc.nerf_forward(nets.nerf, torch.tensor(fx["pts4"]), dirs, nets.nerf_conf)
    assert relmax(a.numpy(), fx["nerf_alpha"]) < 1e-5
    assert relmax(rgb.numpy(), fx["nerf_rgb"]) < 1e-5
    assert relmax(ft.numpy(), fx["nerf_feat"]) < 1e-5
    sp = orc.sample_pdf_det(torch.tensor(fx["spdf_bins"]), torch.tensor(fx["spdf_w"]), 16)
    assert relmax(sp.numpy(), fx["spdf_out"]) < 1e-6           # includes the flat-CDF rows (renderer.py:70)
    lat = orc.extract_fields(nets, [-0.8, -0.7, -0.6], [0.7, 0.8, 0.9], 20)
    assert relmax(lat.numpy(), fx["lattice"]) < 1e-5


def test_sampler_rounds_are_sorted_and_grow(golden):
    fx = golden("white_v03_c0")
    for i, m in enumerate((80, 96, 112, 128)):
        z = fx["z_round%d" % i]
        assert z.shape[1] == m and np.all(np.diff(z, axis=1) >= 0)


def test_schedules():
    assert orc.learning_rate_factor(0) == 0.0 and abs(orc.learning_rate_factor(2500) - 0.5) < 1e-12
    assert abs(orc.learning_rate_factor(300000) - 0.05) < 1e-12
    assert orc.cos_anneal_ratio(25000) == 0.5 and orc.cos_anneal_ratio(10 ** 6) == 1.0
    assert abs(orc.depth_iter_weight(2500) - 0.5) < 1e-12


def test_adam3_reference_trajectory(golden):
    """Three Adam steps of the a-R loop with the oracle reproduce the reference's losses and parameters."""
    fx = golden("adam3")
    B = int(fx["B"])
    st = synth.make_all_states(int(fx["seed"]), wdepth=False, variance=0.3)
    nets = orc.nets_from_numpy(st, requires_grad=True)
    named = orc.all_params(nets)
    opt = torch.optim.Adam([p for _, p in named], lr=5e-4)
    tt = torch.tensor
    o, d, near, far = (tt(fx[k]) for k in ("rays_o", "rays_d", "near", "far"))
    for it in range(int(fx["steps"])):
        for g in opt.param_groups:
            g["lr"] = 5e-4 * orc.learning_rate_factor(it + 100)
        t1, t2 = synth.jitter(int(fx["seed"]), it, B)
        out = orc.render(nets, o, d, near, far, background_rgb=torch.ones(1, 3),
                         cos_anneal_ratio=orc.cos_anneal_ratio(it + 100), t_rand=tt(t1), t_rand_out=tt(t2))
        lo = orc.loss_from_render(out, tt(fx["true_rgb"]))
        opt.zero_grad()
        lo["loss"].backward()
        opt.step()
        assert abs(lo["loss"].item() - fx["losses"][it]) < 2e-5 * abs(fx["losses"][it])
    for n, p in named:
        ref = fx["p_val/" + n]
        got = p.detach().reshape(-1)[fx["p_idx/" + n]].numpy()
        # Adam's first steps move every weight by ~lr regardless of gradient size: compare the
        # displacement scale (lr*3 = 3e-5) rather than demanding bitwise-equal sign decisions.
        assert np.abs(got - ref).max() < 2e-5, n
