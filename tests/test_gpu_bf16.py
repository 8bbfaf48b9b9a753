"""bf16-MFMA throughput path (precision='bf16') against the fp32 oracle. bf16 operands carry 8 significant bits, so
this path is NOT held to 1e-4: the bounds below are ~3x the errors measured on MI355X (tests/probes/bf16_probe.py:
sdf abs err max 5.5e-3 / mean 1e-3, colour PSNR 70 dB at inv_s 20 and 57 dB at inv_s 665, gradient cosine >= 0.9995).
The fp32 path (tests/test_gpu_parity.py, test_gpu_grads.py) is the one that carries the 1e-4 parity claim."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def g(x, dev):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev)


def test_bf16_stages_vs_oracle(golden):
    import oracle.neus_oracle as orc
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    fx = golden("stages")
    st = synth.make_all_states(int(fx["seed"]), wdepth=True)
    rend = factory.build_renderer(wdepth=True, device=dev, states=st, precision="bf16")
    pts, dirs = g(fx["pts"], dev), g(fx["dirs"], dev)
    out = rend.sdf_network(pts).cpu().numpy()
    # sdf: f32 last-layer row + the inputs' bf16 residue in the spare contraction slots (csrc/k_sdf_fwd2.h); what remains is the
    # operand rounding of the eight hidden bf16 layers
    assert np.abs(out[:, 0] - fx["sdf_out"][:, 0]).max() < 6e-3 and np.abs(out[:, 0] - fx["sdf_out"][:, 0]).mean() < 1e-3
    assert np.abs(out[:, 1:] - fx["sdf_out"][:, 1:]).max() < 3e-2 * np.abs(fx["sdf_out"][:, 1:]).max()
    nrm = rend.sdf_network.gradient(pts).cpu().numpy()[:, 0]
    assert np.abs(nrm - fx["sdf_grad"]).max() < 5e-2
    assert np.abs(rend.sdf_network.sdf(pts).cpu().numpy() - fx["sdf_out"][:, :1]).max() < 6e-3
    col = rend.color_network(pts, g(fx["sdf_grad"], dev), dirs, g(fx["sdf_out"][:, 1:], dev)).cpu().numpy()
    assert np.abs(col - fx["color"]).max() < 2e-2
    vdn = rend.depth_network(pts, g(fx["sdf_grad"], dev), dirs, g(fx["sdf_out"][:, 1:], dev)).cpu().numpy()
    assert np.abs(vdn - fx["vdn"]).max() < 2e-2
    a, rgb, ft = rend.nerf(g(fx["pts4"], dev), dirs)
    assert np.abs(a.cpu().numpy() - fx["nerf_alpha"]).max() < 3e-2 * max(1.0, np.abs(fx["nerf_alpha"]).max())
    assert np.abs(rgb.cpu().numpy() - fx["nerf_rgb"]).max() < 3e-2 and np.abs(ft.cpu().numpy() - fx["nerf_feat"]).max() < 3e-2


@pytest.mark.parametrize("name,min_psnr", [("white_v03_c0", 55.0), ("wdepth_v03_c05", 55.0), ("wdepth_v065_c1", 42.0), ("white_n64_v03", 55.0)])
def test_bf16_render_psnr_vs_reference(golden, name, min_psnr):
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    fx = golden(name)
    st = synth.make_all_states(int(fx["seed"]), wdepth=bool(fx["wdepth"]), variance=float(fx["variance"]))
    rend = factory.build_renderer(wdepth=bool(fx["wdepth"]), device=dev, states=st, precision="bf16", n_importance=int(fx["n_importance"]))
    for mode in (False, True):                      # inference path and training path
        with torch.set_grad_enabled(mode):
            out = rend.render(g(fx["rays_o"], dev), g(fx["rays_d"], dev), g(fx["near"], dev), g(fx["far"], dev),
                              background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=float(fx["cos_anneal"]),
                              t_rand=g(fx["t_rand"], dev), t_rand_out=g(fx["t_rand_out"], dev),
                              z_vals_inject=g(fx["z_vals_inside"], dev) if fx["n_importance"] > 0 else None)
        mse = float(((out["color_fine"].detach().cpu().numpy() - fx["out_color_fine"]) ** 2).mean())
        assert 10 * np.log10(1.0 / max(mse, 1e-20)) > min_psnr
        assert np.abs(out["weight_sum"].detach().cpu().numpy() - fx["out_weight_sum"]).max() < 2e-2
        if fx["wdepth"]:
            assert np.abs(out["render_feats"].detach().cpu().numpy() - fx["out_render_feats"]).max() < 2e-2
        assert abs(out["gradient_error"].item() - float(fx["out_gradient_error"])) < 5e-2 * max(float(fx["out_gradient_error"]), 1e-2)


def test_bf16_gradients_and_training_trajectory(golden):
    import oracle.neus_oracle as orc
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    from test_gpu_grads import _gpu_grads, _oracle_grads
    dev = torch.device("cuda:0")
    fx = golden("wdepth_v03_c05")
    # same helper as the fp32 test, but with bf16 networks
    import vdn_train.factory as fac
    orig = fac.build_renderer
    try:
        fac.build_renderer = lambda *a, **k: orig(*a, precision="bf16", **k)
        loss, named, _ = _gpu_grads(fx, dev)
    finally:
        fac.build_renderer = orig
    ref_loss, ref = _oracle_grads(fx, torch.float64)
    assert abs(loss - ref_loss) < 2e-3 * abs(ref_loss)
    for net in ("nerf", "sdf", "color", "vdn"):
        a = torch.cat([p.grad.reshape(-1).cpu().double() for n, p in named if n.startswith(net + ".")])
        b = torch.cat([ref[n].reshape(-1).double() for n, p in named if n.startswith(net + ".")])
        cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
        assert cos > 0.99, (net, cos)
    # 3 Adam steps in bf16 follow the reference's loss trajectory to 1e-3
    fa = golden("adam3")
    B, seed = int(fa["B"]), int(fa["seed"])
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(seed), precision="bf16")
    tr = Trainer(rend, B, dev)
    tr.iter_step, tr._adam_step_offset = 100, 100
    o, d, near, far, rgb = (g(fa[k], dev) for k in ("rays_o", "rays_d", "near", "far", "true_rgb"))
    for it in range(int(fa["steps"])):
        t1, t2 = synth.jitter(seed, it, B)
        sc = tr.train_step(o, d, near, far, rgb, t_rand=g(t1, dev), t_rand_out=g(t2, dev))
        assert abs(sc[0].item() - fa["losses"][it]) < 2e-3 * abs(fa["losses"][it]), (it, sc[0].item(), fa["losses"][it])


def test_bf16_trained_weights_render_the_same_on_the_fp32_kernels():
    """Cross-precision check: train on the bf16 path, then render the SAME weights with the fp32 (parity) kernels and with
    the bf16 kernels. As inv_s grows during training the alpha multiplies the SDF error by it, so agreement at initialisation
    (test above) does not imply agreement later: the two renders of a held-out view must still agree to >= 45 dB."""
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed, steps = 512, 0, 2000
    # Seeded: the geometric init and the per-ray jitter come from torch's generators, and whether the variance takes off within
    # 2 000 steps of this scene depends on the draw - 5 of 32 seeds stay at inv_s ~ 20 (tests/probes/inv_s_growth.py), on
    # the fp32 kernels exactly as on the bf16 ones (same seeds, same trajectories), so it is the optimisation's basin, not
    # the precision. With a fixed seed the run is bit-reproducible across processes (same probe).
    torch.manual_seed(0)
    rend = factory.build_renderer(device=dev, precision="bf16")          # the reference's geometric init
    tr = Trainer(rend, B, dev, conf=dict(warm_up_end=200, end_iter=steps, anneal_end=steps // 4))
    cams = synth.make_cameras(seed)
    gg = lambda x: torch.tensor(x).to(dev)
    order = (synth.uniform(seed, "trainperm", (steps,)) * 40).astype(np.int64) % 40
    for it in range(steps):
        img = int(order[it]) if int(order[it]) != 7 else 8           # view 7 is held out
        o, d = synth.random_pixel_batch(seed, it, img, B, cams=cams, crop=420)
        near, far = synth.near_far_from_sphere(o, d)
        tr.train_step(gg(o), gg(d), gg(near), gg(far), gg(synth.target_colors(o, d, 0.5)))
    inv_s = float(torch.exp(rend.deviation_network.variance * 10).item())
    assert inv_s > 25.0                                               # it has moved from its initial 20.1
    ref = factory.build_renderer(device=dev, precision="fp32")
    for a, b in ((ref.nerf, rend.nerf), (ref.sdf_network, rend.sdf_network), (ref.deviation_network, rend.deviation_network),
                 (ref.color_network, rend.color_network)):
        a.load_state_dict({k: v.detach().clone() for k, v in b.state_dict().items()})
    vx, vy = np.meshgrid(np.linspace(190, 610, 48), np.linspace(190, 610, 48))
    vo, vd = synth.pixel_rays(cams[7], vx.reshape(-1), vy.reshape(-1))
    vn, vf = synth.near_far_from_sphere(vo, vd)
    cols = {}
    with torch.no_grad():
        for name, r in (("bf16", rend), ("fp32", ref)):
            out = []
            for i in range(0, vo.shape[0], 512):
                o_ = r.render(gg(vo[i:i + 512]), gg(vd[i:i + 512]), gg(vn[i:i + 512]), gg(vf[i:i + 512]), perturb_overwrite=0,
                              background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=1.0)
                out.append(o_["color_fine"])
            cols[name] = torch.cat(out)
    mse = ((cols["bf16"] - cols["fp32"]) ** 2).mean().item()
    psnr = 10.0 * np.log10(1.0 / max(mse, 1e-20))
    print("bf16-trained weights, inv_s %.1f: bf16 render vs fp32 render %.1f dB" % (inv_s, psnr))
    assert psnr >= 45.0, (psnr, inv_s)
