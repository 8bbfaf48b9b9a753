"""GPU parity of the hand-derived backward: d loss / d every parameter through NeuSRenderer.render
(custom autograd node -> HIP kernels) against the oracle's autograd and the reference's own gradient
samples in the golden fixtures. z is injected (the sampler is a no-grad stage). Tolerance: 1e-4 of the
largest entry of each gradient tensor vs the fp64 oracle (BASELINE north-star; SURVEY.md 4 'backward')."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def g(x, dev):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev)


def _loss(out, true_rgb, gt_feats, wdepth, mask_weight=0.0):
    B = true_rgb.shape[0]
    mask_sum = B + 1e-5
    loss = (out["color_fine"] - true_rgb).abs().sum() / mask_sum + out["gradient_error"] * 0.1
    if wdepth:
        loss = loss + (out["render_feats"] - gt_feats).abs().sum() / mask_sum * 0.7
    if mask_weight:
        loss = loss + torch.nn.functional.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3),
                                                               torch.ones_like(out["weight_sum"])) * mask_weight
    return loss


def _gpu_grads(fx, dev, mask_weight=0.0, **kw):
    from vdn_train import synth, factory
    wdepth = bool(fx["wdepth"])
    st = synth.make_all_states(int(fx["seed"]), wdepth=wdepth, variance=float(fx["variance"]))
    rend = factory.build_renderer(wdepth=wdepth, device=dev, states=st, n_importance=int(fx["n_importance"]), **kw)
    z_inj = g(fx["z_vals_inside"], dev) if fx["n_importance"] > 0 else None
    out = rend.render(g(fx["rays_o"], dev), g(fx["rays_d"], dev), g(fx["near"], dev), g(fx["far"], dev),
                      perturb_overwrite=(-1 if fx["perturb"] > 0 else 0),
                      background_rgb=torch.ones(1, 3, device=dev) if fx["white"] else None,
                      cos_anneal_ratio=float(fx["cos_anneal"]), t_rand=g(fx["t_rand"], dev), t_rand_out=g(fx["t_rand_out"], dev),
                      z_vals_inject=z_inj)
    loss = _loss(out, g(fx["true_rgb"], dev), g(fx["gt_feats"], dev) if wdepth else None, wdepth, mask_weight)
    loss.backward()
    named = []
    for key, mod in (("nerf", rend.nerf), ("sdf", rend.sdf_network), ("variance", rend.deviation_network),
                     ("color", rend.color_network), ("vdn", rend.depth_network)):
        if mod is None:
            continue
        for n, p in mod.named_parameters():
            named.append((key + "." + n if key != "variance" else "variance", p))
    return loss.item(), named, out


def _oracle_grads(fx, dtype, mask_weight=0.0):
    import oracle.neus_oracle as orc
    from vdn_train import synth
    wdepth = bool(fx["wdepth"])
    st = synth.make_all_states(int(fx["seed"]), wdepth=wdepth, variance=float(fx["variance"]))
    nets = orc.nets_from_numpy(st, dtype=dtype, requires_grad=True)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=dtype)
    out = orc.render(nets, tt(fx["rays_o"]), tt(fx["rays_d"]), tt(fx["near"]), tt(fx["far"]),
                     orc.RendererConf(n_importance=int(fx["n_importance"])),
                     perturb_overwrite=(-1 if fx["perturb"] > 0 else 0),
                     background_rgb=torch.ones(1, 3, dtype=dtype) if fx["white"] else None,
                     cos_anneal_ratio=float(fx["cos_anneal"]), t_rand=tt(fx["t_rand"]), t_rand_out=tt(fx["t_rand_out"]),
                     z_vals_inject=tt(fx["z_vals_inside"]) if fx["n_importance"] > 0 else None)
    loss = _loss(out, tt(fx["true_rgb"]), tt(fx["gt_feats"]) if wdepth else None, wdepth, mask_weight)
    named = orc.all_params(nets)
    gs = torch.autograd.grad(loss, [p for _, p in named], allow_unused=True)
    return loss.item(), {n: (torch.zeros_like(p) if gr is None else gr).detach() for (n, p), gr in zip(named, gs)}


def _compare(named, ref, tol, ref32=None):
    """rel-to-max error of every gradient tensor vs the fp64 oracle. Where the fp32 oracle (= what the fp32
    reference computes) is itself further than `tol` from fp64 - tiny, cancellation-dominated gradients such as
    the background NeRF's first layers (|g| ~ 1e-5) - the bound is 3x that measured fp32 floor instead."""
    #
    # ReLU sign flips: a hidden unit whose pre-activation is within ~1e-7 of zero is 'on' in one fp32 evaluation
    # and 'off' in another (or in fp64). With ~3e6 unit-points per evaluation a few such flips are inevitable; a flip
    # in layer l changes one row of dW_l and, through the delta chain, every layer below it, by one point's
    # contribution - a visible fraction of the total only because the fixtures hold 16-24 rays. A tensor that misses
    # the strict bound is therefore accepted when (a) its own max error stays below 1e-2 and (b) the whole network's
    # gradient (all its tensors concatenated) is within 1e-3 relative L2 of the oracle's.
    # test_param_grads_larger_batch applies the strict per-tensor bound where flips are diluted.
    rows, bad, net_num, net_den = [], [], {}, {}
    for n, p in named:
        gg = p.grad
        assert gg is not None, n
        a, b = gg.detach().cpu().double().numpy(), ref[n].double().numpy()
        assert a.shape == b.shape, n
        net = n.split(".")[0]
        net_num[net] = net_num.get(net, 0.0) + float(((a - b) ** 2).sum())
        net_den[net] = net_den.get(net, 0.0) + float((b ** 2).sum())
    net_rel = {k: (net_num[k] / (net_den[k] + 1e-300)) ** 0.5 for k in net_num}
    for n, p in named:
        a, b = p.grad.detach().cpu().double().numpy(), ref[n].double().numpy()
        scale = np.abs(b).max()
        err = np.abs(a - b).max() / (scale + 1e-30) if scale > 0 else np.abs(a).max()
        floor = 0.0
        if ref32 is not None and scale > 0:
            floor = np.abs(ref32[n].double().numpy() - b).max() / scale
        rows.append((err, n, scale, floor))
        strict = err < max(tol, 3 * floor)
        ok = strict or (err < 1e-2 and net_rel[n.split(".")[0]] < 1e-3)
        if not ok:
            bad.append(n)
        rows[-1] = rows[-1] + (strict,)
    rows.sort(reverse=True)
    report = "\n".join("%-36s err %.2e  fp32-floor %.2e  |g|max %.2e" % (n, e, fl, s) for e, n, s, fl, _ in rows[:12])
    assert not bad, "gradient errors beyond max(%.0e, 3x fp32 floor) in %s (network rel-L2 %s)\n%s" % (tol, bad, net_rel, report)
    return rows


@pytest.mark.parametrize("name", ["white_v03_c0", "white_v03_c05_det", "wdepth_v03_c05", "white_n64_v03"])
def test_param_grads_vs_oracle_fp64(golden, name):
    dev = torch.device("cuda:0")
    fx = golden(name)
    loss, named, _ = _gpu_grads(fx, dev)
    ref_loss, ref = _oracle_grads(fx, torch.float64)
    _, ref32 = _oracle_grads(fx, torch.float32)
    assert abs(loss - ref_loss) < 2e-5 * abs(ref_loss)
    assert abs(loss - float(fx["loss"])) < 2e-5 * abs(float(fx["loss"]))      # the reference's own loss
    rows = _compare(named, ref, 1e-4, ref32)
    # the bulk of the parameters must be at 1e-4 outright (not only within the floor)
    assert sum(1 for r in rows if r[0] < 1e-4) >= 0.6 * len(rows)
    # the reference's own gradients (fp32 CPU autograd; sampled entries + norms in the fixture)
    floors = {n: fl for _, n, _, fl, _ in rows}
    strict_ok = {n: st for _, n, _, _, st in rows}
    for n, p in named:
        if not strict_ok[n]:
            continue                      # accepted through the ReLU-flip rule above: not comparable entry by entry
        rv = fx["grad_val/" + n]
        gv = p.grad.detach().cpu().reshape(-1)[torch.as_tensor(fx["grad_idx/" + n])].numpy()
        tol = max(3e-4, 6 * floors[n])
        assert np.abs(gv - rv).max() <= tol * np.abs(ref[n].numpy()).max() + 1e-12, n
        rn = float(fx["grad_norm/" + n])
        assert abs(float(p.grad.norm()) - rn) <= max(1e-4, 3 * floors[n]) * rn + 1e-12, n


def test_param_grads_with_mask_loss_and_high_inv_s(golden):
    """weight_sum (BCE mask loss, dpt_runner.py:233) and the inv_s = 665 regime. At inv_s = 665 the fp32 reference
    itself is 1e-2 away from fp64 on some SDF gradients (tests/golden/make_golden.py), so the comparison is
    against the fp32 oracle with the noise floor measured between the fp32 and fp64 oracles."""
    dev = torch.device("cuda:0")
    fx = golden("white_v03_c0")
    loss, named, _ = _gpu_grads(fx, dev, mask_weight=0.3)
    ref_loss, ref = _oracle_grads(fx, torch.float64, mask_weight=0.3)
    _, ref32 = _oracle_grads(fx, torch.float32, mask_weight=0.3)
    assert abs(loss - ref_loss) < 2e-5 * abs(ref_loss)
    _compare(named, ref, 1e-4, ref32)
    fx = golden("white_v065_c1")
    loss, named, _ = _gpu_grads(fx, dev)
    _, ref64 = _oracle_grads(fx, torch.float64)
    _, ref32 = _oracle_grads(fx, torch.float32)
    _compare(named, ref64, 1e-4, ref32)


def test_no_outside_samples_and_second_step(golden):
    """n_outside = 0 (no background pass: NeRF gets no gradient) and two consecutive steps on one engine."""
    import oracle.neus_oracle as orc
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    B = 8
    st = synth.make_all_states(31, wdepth=False)
    rend = factory.build_renderer(device=dev, states=st, n_outside=0)
    nets = orc.nets_from_numpy(st, dtype=torch.float64, requires_grad=True)
    cams = synth.make_cameras(31)
    px = np.floor(synth.uniform(31, "no/x", (B,)) * 300) + 250
    py = np.floor(synth.uniform(31, "no/y", (B,)) * 300) + 250
    o, d = synth.pixel_rays(cams[1], px, py)
    near, far = synth.near_far_from_sphere(o, d)
    target = synth.target_colors(o, d)
    for step in range(2):
        out = rend.render(g(o, dev), g(d, dev), g(near, dev), g(far, dev), perturb_overwrite=0, cos_anneal_ratio=1.0)
        z = None
        loss = (out["color_fine"] - g(target, dev)).abs().sum() / B + 0.1 * out["gradient_error"]
        for m in (rend.sdf_network, rend.color_network, rend.deviation_network, rend.nerf):
            m.zero_grad(set_to_none=True)
        loss.backward()
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in rend.nerf.parameters())
    t64 = lambda x: torch.tensor(x, dtype=torch.float64)
    conf = orc.RendererConf(n_outside=0)
    # same z as the GPU used: take the GPU's inside z through its sampler
    zg, _ = rend._sample(g(o, dev), g(d, dev), g(near, dev).reshape(-1), g(far, dev).reshape(-1), 0.0, None, None, None)
    ref = orc.render(nets, t64(o), t64(d), t64(near), t64(far), conf, perturb_overwrite=0, cos_anneal_ratio=1.0,
                     z_vals_inject=zg.cpu().double())
    rl = (ref["color_fine"] - t64(target)).abs().sum() / B + 0.1 * ref["gradient_error"]
    named = [(n, p) for n, p in orc.all_params(nets) if not n.startswith("nerf.")]
    gs = torch.autograd.grad(rl, [p for _, p in named])
    refd = {n: gr for (n, _), gr in zip(named, gs)}
    got = [("sdf." + n, p) for n, p in rend.sdf_network.named_parameters()] + [("variance", rend.deviation_network.variance)] + \
          [("color." + n, p) for n, p in rend.color_network.named_parameters()]
    _compare(got, refd, 1e-4)


def test_param_grads_larger_batch():
    """192 rays: single ReLU flips are diluted, so the plain criterion max|dg| / max|g| < max(1e-4, 3x fp32 floor)
    is applied to every parameter tensor, against the fp64 oracle."""
    import oracle.neus_oracle as orc
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    B, seed = 192, 41
    st = synth.make_all_states(seed, wdepth=True)
    rend = factory.build_renderer(wdepth=True, device=dev, states=st)
    cams = synth.make_cameras(seed)
    px = np.floor(synth.uniform(seed, "lb/x", (B,)) * 520) + 140
    py = np.floor(synth.uniform(seed, "lb/y", (B,)) * 520) + 140
    o, d = synth.pixel_rays(cams[3], px, py)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, 0, B)
    target = synth.target_colors(o, d)
    gtf = synth.uniform(seed, "lb/f", (B, 96)).astype(np.float32)
    with torch.no_grad():
        z, _ = rend._sample(g(o, dev), g(d, dev), g(near, dev).reshape(-1), g(far, dev).reshape(-1), 1.0, g(t1, dev), g(t2, dev), None)
    out = rend.render(g(o, dev), g(d, dev), g(near, dev), g(far, dev), background_rgb=torch.ones(1, 3, device=dev),
                      cos_anneal_ratio=0.4, t_rand=g(t1, dev), t_rand_out=g(t2, dev), z_vals_inject=z)
    loss = _loss(out, g(target, dev), g(gtf, dev), True)
    loss.backward()
    named = []
    for key, mod in (("nerf", rend.nerf), ("sdf", rend.sdf_network), ("variance", rend.deviation_network),
                     ("color", rend.color_network), ("vdn", rend.depth_network)):
        named += [(key + "." + n if key != "variance" else "variance", p) for n, p in mod.named_parameters()]
    refs = {}
    for dt in (torch.float64, torch.float32):
        nets = orc.nets_from_numpy(st, dtype=dt, requires_grad=True)
        tt = lambda x: torch.tensor(np.asarray(x), dtype=dt)
        ro = orc.render(nets, tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3, dtype=dt), cos_anneal_ratio=0.4,
                        t_rand=tt(t1), t_rand_out=tt(t2), z_vals_inject=z.cpu().to(dt))
        rl = _loss(ro, tt(target), tt(gtf), True)
        nm = orc.all_params(nets)
        gs = torch.autograd.grad(rl, [p for _, p in nm])
        refs[dt] = ({n: gr for (n, _), gr in zip(nm, gs)}, rl.item())
    assert abs(loss.item() - refs[torch.float64][1]) < 2e-5 * abs(refs[torch.float64][1])
    bad = []
    for n, p in named:
        b = refs[torch.float64][0][n].double().numpy()
        scale = np.abs(b).max()
        err = np.abs(p.grad.cpu().double().numpy() - b).max() / scale
        floor = np.abs(refs[torch.float32][0][n].double().numpy() - b).max() / scale
        if err >= max(1e-4, 3 * floor):
            bad.append((n, err, floor))
    assert not bad, bad


def test_trainer_three_adam_steps_match_reference(golden):
    """The fast path (no autograd: engine.forward -> loss kernel -> engine.backward -> fused Adam) reproduces the
    REFERENCE's 3-step trajectory (losses and parameters after torch.optim.Adam, tests/golden/adam3.npz)."""
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    fx = golden("adam3")
    B, seed = int(fx["B"]), int(fx["seed"])
    st = synth.make_all_states(seed, wdepth=False, variance=0.3)
    rend = factory.build_renderer(device=dev, states=st)
    tr = Trainer(rend, B, dev)
    o, d, near, far, rgb = (g(fx[k], dev) for k in ("rays_o", "rays_d", "near", "far", "true_rgb"))
    tr.iter_step = 100                                    # fixture: lr / cos-anneal evaluated at iter_step = it + 100
    tr._adam_step_offset = 100
    for it in range(int(fx["steps"])):
        t1, t2 = synth.jitter(seed, it, B)
        sc = tr.train_step(o, d, near, far, rgb, t_rand=g(t1, dev), t_rand_out=g(t2, dev))
        assert abs(sc[0].item() - fx["losses"][it]) < 5e-5 * abs(fx["losses"][it]), (it, sc[0].item(), fx["losses"][it])
    tr.join()          # the background / colour networks' half of the last step runs on the side stream (Trainer docstring)
    named = [("nerf." + n, p) for n, p in rend.nerf.named_parameters()] + [("sdf." + n, p) for n, p in rend.sdf_network.named_parameters()] + \
            [("variance", rend.deviation_network.variance)] + [("color." + n, p) for n, p in rend.color_network.named_parameters()]
    for n, p in named:
        got = p.detach().cpu().reshape(-1)[torch.as_tensor(fx["p_idx/" + n])].numpy()
        # Adam's first steps move every weight by ~lr whatever the gradient size; sign decisions on ~0 gradients can differ
        assert np.abs(got - fx["p_val/" + n]).max() < 2e-5, n
    # checkpoint round trip in the reference's key schema
    ck = tr.state_dict()
    assert set(ck) == {"nerf", "sdf_network_fine", "variance_network_fine", "color_network_fine", "depth_network_fine", "optimizer", "iter_step"}
    assert ck["iter_step"] == 103 and ck["depth_network_fine"] is None and len(ck["optimizer"]["state"]) == len(tr.params)
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros_like(p)) for p in tr.params], lr=1.0)
    opt.load_state_dict({"state": ck["optimizer"]["state"], "param_groups": ck["optimizer"]["param_groups"]})   # torch accepts it
    before = tr.param_flat.clone()
    tr.param_flat.zero_()
    tr.load_checkpoint(ck)
    assert torch.equal(tr.param_flat, before) and tr.iter_step == 103


def test_trainer_wdepth_mask_step_equals_autograd_path():
    """Runner.train's loss with a real mask, mask_weight > 0 and the VDN depth term (dpt_runner.py:207-243): the fused loss kernel
    + engine backward of the Trainer give the same scalars and the same gradient as render() + the reference's torch loss +
    loss.backward() through the custom autograd node."""
    import torch.nn.functional as F
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 128, 47
    st = synth.make_all_states(seed, wdepth=True)
    cams = synth.make_cameras(seed)
    o, d = synth.random_pixel_batch(seed, 0, 2, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, 0, B)
    o, d, near, far, t1, t2 = (g(x, dev) for x in (o, d, near, far, t1, t2))
    rgb = g(synth.uniform(seed, "tm/rgb", (B, 3)), dev)
    gtf = g(synth.uniform(seed, "tm/f", (B, 96)), dev)
    mask = (g(synth.uniform(seed, "tm/m", (B, 1)), dev) > 0.35).float()
    conf = dict(extract_depth=True, depth_start_iter=10, mask_weight=0.1, anneal_end=100)
    tr = Trainer(factory.build_renderer(wdepth=True, device=dev, states=st), B, dev, conf=conf)
    tr.iter_step, tr._adam_step_offset, tr.depth_iter = 40, 40, 1200
    w_depth, cos = tr.depth_iter_weight(), tr.cos_anneal_ratio()
    sc = tr.train_step(o, d, near, far, rgb, gt_feats=gtf, mask=mask, t_rand=t1, t_rand_out=t2).cpu().numpy()
    got = tr.engine.param_grads()

    rend = factory.build_renderer(wdepth=True, device=dev, states=st)
    out = rend.render(o, d, near, far, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=cos, t_rand=t1, t_rand_out=t2)
    mask_sum = mask.sum() + 1e-5
    color_loss = ((out["color_fine"] - rgb) * mask).abs().sum() / mask_sum
    psnr = 20.0 * torch.log10(1.0 / (((out["color_fine"] - rgb) ** 2 * mask).sum() / (mask_sum * 3.0)).sqrt())
    mask_loss = F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
    depth_loss = ((out["render_feats"] - gtf) * mask).abs().sum() / mask_sum
    loss = color_loss + out["gradient_error"] * 0.1 + mask_loss * 0.1 + depth_loss * w_depth
    loss.backward()
    want_sc = [x.item() for x in (loss, color_loss, psnr, out["gradient_error"], depth_loss, mask_loss)]
    np.testing.assert_allclose(sc, want_sc, rtol=2e-5)
    params = rend._all_parameters()
    assert len(params) == len(got)
    for i, (p, gr) in enumerate(zip(params, got)):
        scale = p.grad.abs().max().item()
        assert scale > 0 and (p.grad - gr).abs().max().item() <= 1e-5 * scale, (i, tuple(p.shape))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_trainer_work_lists_do_not_change_the_step(monkeypatch, precision):
    """The Trainer skips foreground samples beyond the relaxed sphere and background samples inside the unit sphere (they
    enter the loss only through exact zeros, renderer.py:284-299): same scalars bit for bit, same gradient up to the
    summation order of the weight-gradient GEMM, as evaluating every sample."""
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 256, 53
    st = synth.make_all_states(seed, wdepth=True)
    cams = synth.make_cameras(seed)
    o, d = synth.random_pixel_batch(seed, 0, 4, B, cams=cams)            # full frame: many samples outside both spheres
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, 0, B)
    o, d, near, far, t1, t2 = (g(x, dev) for x in (o, d, near, far, t1, t2))
    rgb = g(synth.uniform(seed, "wl/rgb", (B, 3)), dev)
    gtf = g(synth.uniform(seed, "wl/f", (B, 96)), dev)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("VDN_FG_COMPACT", mode)
        monkeypatch.setenv("VDN_BG_COMPACT", mode)
        tr = Trainer(factory.build_renderer(wdepth=True, device=dev, states=st, precision=precision), B, dev,
                     conf=dict(extract_depth=True, depth_start_iter=-1))
        tr.iter_step = 10
        sc = tr.train_step(o, d, near, far, rgb, gt_feats=gtf, t_rand=t1, t_rand_out=t2).cpu().numpy().copy()
        eng = tr.engine
        res[mode] = (sc, [x.clone() for x in eng.param_grads()], int(eng.w["fg_active"][1]), int(eng.w["bg_active"][1]))
    sc1, g1, nfg1, nbg1 = res["1"]
    sc0, g0, nfg0, nbg0 = res["0"]
    assert nfg0 == B * 128 and nbg0 == B * 160 and 0 < nfg1 < nfg0 and 32 * B <= nbg1 < nbg0
    assert np.array_equal(sc1, sc0)
    for a, b in zip(g1, g0):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-12


def test_trainer_without_background_evaluates_every_sample():
    """n_outside = 0: render_core has no inside_sphere blend (renderer.py:289), so the Trainer must not skip far samples;
    its step equals the autograd path's loss and gradients."""
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 96, 59
    st = synth.make_all_states(seed, wdepth=False)
    cams = synth.make_cameras(seed)
    o, d = synth.random_pixel_batch(seed, 0, 1, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    t1, _ = synth.jitter(seed, 0, B)
    o, d, near, far, t1 = (g(x, dev) for x in (o, d, near, far, t1))
    rgb = g(synth.uniform(seed, "nb/rgb", (B, 3)), dev)
    tr = Trainer(factory.build_renderer(device=dev, states=st, n_outside=0), B, dev)
    tr.iter_step = 10
    sc = tr.train_step(o, d, near, far, rgb, t_rand=t1).cpu().numpy()
    assert int(tr.engine.w["fg_active"][1]) == B * 128
    got = tr.engine.param_grads()
    rend = factory.build_renderer(device=dev, states=st, n_outside=0)
    out = rend.render(o, d, near, far, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=10 / 50000, t_rand=t1)     # the Trainer's value at iter_step 10
    loss = (out["color_fine"] - rgb).abs().sum() / (B + 1e-5) + 0.1 * out["gradient_error"]
    loss.backward()
    assert abs(sc[0] - loss.item()) <= 2e-6 * abs(loss.item())
    grads = {id(p): gr for p, gr in zip(tr.params, got)}
    for p_ref, p_tr in zip(rend._all_parameters(), tr.params):
        if p_ref.grad is None:
            continue
        gr = grads[id(p_tr)]
        assert (p_ref.grad - gr).abs().max().item() <= 1e-5 * p_ref.grad.abs().max().item() + 1e-12


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("B", [1, 33, 130])
def test_trainer_ragged_batches(B, precision):
    """Batch sizes that are not multiples of the 32-point wave tile (partial tiles, padded bf16 planes, short work lists):
    the Trainer's step equals render() + loss.backward() through the autograd node on the same rays."""
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    seed = 61
    st = synth.make_all_states(seed, wdepth=True)
    cams = synth.make_cameras(seed)
    o, d = synth.random_pixel_batch(seed, B, 5, B, cams=cams, crop=600)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, B, B)
    o, d, near, far, t1, t2 = (g(x, dev) for x in (o, d, near, far, t1, t2))
    rgb, gtf = g(synth.uniform(seed, "rg/rgb%d" % B, (B, 3)), dev), g(synth.uniform(seed, "rg/f%d" % B, (B, 96)), dev)
    tr = Trainer(factory.build_renderer(wdepth=True, device=dev, states=st, precision=precision), B, dev,
                 conf=dict(extract_depth=True, depth_start_iter=-1))
    tr.iter_step, tr.depth_iter = 10, 2500
    wd, cos = tr.depth_iter_weight(), tr.cos_anneal_ratio()
    sc = tr.train_step(o, d, near, far, rgb, gt_feats=gtf, t_rand=t1, t_rand_out=t2).cpu().numpy()
    got = tr.engine.param_grads()
    rend = factory.build_renderer(wdepth=True, device=dev, states=st, precision=precision)
    out = rend.render(o, d, near, far, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=cos, t_rand=t1, t_rand_out=t2)
    ms = B + 1e-5
    loss = (out["color_fine"] - rgb).abs().sum() / ms + 0.1 * out["gradient_error"] + (out["render_feats"] - gtf).abs().sum() / ms * wd
    loss.backward()
    assert np.isfinite(sc).all() and abs(sc[0] - loss.item()) <= 3e-6 * abs(loss.item())
    tol = 2e-5 if precision == "fp32" else 2e-3          # bf16: the work lists change which rows share a dW split
    for p, gr in zip(rend._all_parameters(), got):
        assert (p.grad - gr).abs().max().item() <= tol * p.grad.abs().max().item() + 1e-12


def test_gradients_cdf_and_s_val_outputs_are_attached(golden):
    """renderer.py:426-439 returns `gradients`, `cdf_fine` and `s_val` with their graph; a loss built on them must
    differentiate like the reference's (the runner's own loss does not use them)."""
    import oracle.neus_oracle as orc
    from vdn_train import synth, factory
    fx = golden("white_v03_c05_det")
    dev = torch.device("cuda:0")
    B = int(fx["B"])
    w1 = synth.uniform(5, "aux/w1", (B, 128, 3)).astype(np.float32) - 0.5
    w2 = synth.uniform(5, "aux/w2", (B, 128)).astype(np.float32) - 0.5

    def extra(out, tt):
        # (s_val, weight_sum and weight_max are outputs of the autograd node: their adjoints are folded in by its backward)
        return ((out["gradients"] * tt(w1)).sum() * 0.01 + (out["cdf_fine"] * tt(w2)).sum() * 0.05 + out["s_val"].sum() * 3.0
                + out["weight_max"].sum() * 0.3 + (out["weight_sum"] ** 2).sum() * 0.2)

    st = synth.make_all_states(int(fx["seed"]), wdepth=False, variance=float(fx["variance"]))
    rend = factory.build_renderer(wdepth=False, device=dev, states=st)
    out = rend.render(g(fx["rays_o"], dev), g(fx["rays_d"], dev), g(fx["near"], dev), g(fx["far"], dev), perturb_overwrite=0,
                      background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=float(fx["cos_anneal"]),
                      z_vals_inject=g(fx["z_vals_inside"], dev))
    assert out["gradients"].requires_grad and out["cdf_fine"].requires_grad and out["s_val"].requires_grad
    loss = _loss(out, g(fx["true_rgb"], dev), None, False) + extra(out, lambda x: g(x, dev))
    loss.backward()
    named = [(k + "." + n if k != "variance" else "variance", p)
             for k, m in (("nerf", rend.nerf), ("sdf", rend.sdf_network), ("variance", rend.deviation_network), ("color", rend.color_network))
             for n, p in m.named_parameters()]
    refs = []
    for dtype in (torch.float64, torch.float32):
        nets = orc.nets_from_numpy(st, dtype=dtype, requires_grad=True)
        tt = lambda x: torch.tensor(np.asarray(x), dtype=dtype)
        oo = orc.render(nets, tt(fx["rays_o"]), tt(fx["rays_d"]), tt(fx["near"]), tt(fx["far"]), orc.RendererConf(n_importance=64),
                        perturb_overwrite=0, background_rgb=torch.ones(1, 3, dtype=dtype), cos_anneal_ratio=float(fx["cos_anneal"]),
                        z_vals_inject=tt(fx["z_vals_inside"]))
        lo = _loss(oo, tt(fx["true_rgb"]), None, False) + extra(oo, tt)
        pn = orc.all_params(nets)
        gs = torch.autograd.grad(lo, [p for _, p in pn], allow_unused=True)
        refs.append((lo.item(), {n: (torch.zeros_like(p) if gr is None else gr).detach() for (n, p), gr in zip(pn, gs)}))
    assert abs(loss.item() - refs[0][0]) < 5e-5 * abs(refs[0][0])
    _compare(named, refs[0][1], 1e-4, refs[1][1])


@pytest.mark.parametrize("B,skip_far", [(512, True), (96, True), (37, False)])
def test_one_launch_sdf_backward_equals_rbar_then_fbar(monkeypatch, B, skip_far):
    """bf16 path: vdn_sdf_bwd_split_bf16 (both adjoint chains of the SDF network in one feature-split launch, the second-order
    term ex_l kept in the wave that produced it; csrc/k_sdf_bwd_split.h) against vdn_sdf_bwd_rbar_bf16 + vdn_sdf_bwd_fbar_bf16 on
    the same forward state: the planes the weight-gradient GEMM reads (UB, AB) and every parameter gradient, bit for bit; full
    batch with work lists, a ragged batch, a batch with every sample evaluated."""
    from vdn_train import synth, factory
    from vdn_hip.train import TrainEngine
    from vdn_hip import layout
    dev = torch.device("cuda:0")
    seed = 33
    st = synth.make_all_states(seed, wdepth=False)
    cams = synth.make_cameras(seed)
    o, d = synth.random_pixel_batch(seed, 0, 3, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, 0, B)
    o, d, near, far, t1, t2 = (g(x, dev) for x in (o, d, near, far, t1, t2))
    g_color = g(synth.uniform(seed, "split/gc", (B, 3)) - 0.5, dev)
    g_eik = torch.tensor([0.1], device=dev)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VDN_SDF_BWD_SPLIT", mode)
        rend = factory.build_renderer(device=dev, states=st, precision="bf16")
        eng = TrainEngine(rend, B, dev)
        with torch.no_grad():
            z, z_out = rend._sample(o, d, near.reshape(B), far.reshape(B), 1.0, t1, t2, None)
        eng.forward(o, d, z.contiguous(), z_out, torch.ones(3, device=dev), 0.3, skip_far=skip_far)
        eng.backward(g_color, None, None, g_eik)
        torch.cuda.synchronize()
        res[mode] = (eng.grad_flat.clone(), {k: eng.w[k].clone() for k in ("UB", "AB")}, int(eng.w["fg_active"][1].item()), eng.Pp)
    (g0, p0, n0, Pp), (g1, p1, n1, _) = res["0"], res["1"]
    assert n0 == n1 and (n0 < B * 128) == skip_far
    off = 0
    for l, cols in enumerate((64, 256, 256, 256, 288, 256, 256, 256, 256)):
        nc = 224 if l == 4 else cols                # (ub_4's h part is 7 tiles; its PE part sits in columns 224..287)
        a, b = (layout.from_pt32(p["UB"][off:off + Pp * cols], Pp, cols)[:n0] for p in (p0, p1))
        assert torch.equal(a, b), ("UB", l, (a.float() - b.float()).abs().max().item())
        off += Pp * cols
    off = 0
    for l, cols in [(8, 288)] + [(k, 256) for k in range(7, -1, -1)]:
        nc = 224 if l == 3 else cols
        a, b = (layout.from_pt32(p["AB"][off:off + Pp * cols], Pp, cols)[:n0, :nc] for p in (p0, p1))
        assert torch.equal(a, b), ("AB", l, (a.float() - b.float()).abs().max().item())
        off += Pp * cols
    assert torch.equal(g0, g1) and torch.isfinite(g0).all()
