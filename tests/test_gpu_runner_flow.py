"""The reference runner's own flow over the drop-in boundary (dpt_runner.py:117-144, 197-257, 744): networks built from
the conf's kwargs under torch.set_default_tensor_type('torch.cuda.FloatTensor'), .to(device), torch.optim.Adam over
.parameters(), then render -> the runner's torch loss -> zero_grad -> backward -> optimizer.step (in-place parameter
updates, which the weight images must notice). Checked against the REFERENCE's 3-step Adam trajectory (adam3.npz)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# confs/womsk_white.conf:41-90, as pyhocon hands them to the constructors (dpt_runner.py:117-142)
CONF = {
    "nerf": dict(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], rgb_dims=3, use_viewdirs=True),
    "sdf_network": dict(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                        geometric_init=True, weight_norm=True),
    "variance_network": dict(init_val=0.3),
    "rendering_network": dict(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=4, weight_norm=True,
                              multires_view=4, squeeze_out=True),
    "neus_renderer": dict(n_samples=64, n_importance=64, n_outside=32, up_sample_steps=4, perturb=1.0),
}


def test_runner_flow_three_adam_steps_match_reference(golden):
    from vdn_train import synth
    fx = golden("adam3")
    B, seed = int(fx["B"]), int(fx["seed"])
    torch.set_default_tensor_type("torch.cuda.FloatTensor")          # dpt_runner.py:744
    try:
        from dpt_models.fields import RenderingNetwork, SDFNetwork, SingleVarianceNetwork, NeRF      # dpt_runner.py:18-19
        from dpt_models.renderer import NeuSRenderer
        device = torch.device("cuda")
        nerf_outside = NeRF(**CONF["nerf"]).to(device)
        sdf_network = SDFNetwork(**CONF["sdf_network"]).to(device)
        deviation_network = SingleVarianceNetwork(**CONF["variance_network"]).to(device)
        color_network = RenderingNetwork(**CONF["rendering_network"]).to(device)
        params_to_train = []
        for m in (nerf_outside, sdf_network, deviation_network, color_network):
            params_to_train += list(m.parameters())
        renderer = NeuSRenderer(nerf_outside, sdf_network, deviation_network, color_network, None, **CONF["neus_renderer"])
        optimizer = torch.optim.Adam(params_to_train, lr=5e-4)
        # the fixture's weights (a checkpoint, in the reference's key schema: dpt_runner.py:350-359)
        st = synth.make_all_states(seed, wdepth=False, variance=0.3)
        tt = lambda d: {k: torch.tensor(v) for k, v in d.items()}
        nerf_outside.load_state_dict(tt(st["nerf"]), strict=False)
        sdf_network.load_state_dict(tt(st["sdf_network_fine"]))
        deviation_network.load_state_dict(tt(st["variance_network_fine"]))
        color_network.load_state_dict(tt(st["color_network_fine"]))

        data = torch.tensor(np.concatenate([fx["rays_o"], fx["rays_d"], np.ones((B, 1), np.float32), fx["true_rgb"]], axis=1))
        near, far = torch.tensor(fx["near"]), torch.tensor(fx["far"])
        losses = []
        for it in range(int(fx["steps"])):
            for gq in optimizer.param_groups:
                gq["lr"] = 5e-4 * (it + 100) / 5000.0                       # dpt_runner.py:311-312 at iter_step = it + 100
            rays_o, rays_d, mask, true_rgb = data[:, :3], data[:, 3:6], data[:, 6:7], data[:, 7:10]    # dpt_runner.py:201 (views)
            background_rgb = torch.ones([1, 3])
            mask = torch.ones_like(mask)
            mask_sum = mask.sum() + 1e-5
            t1, t2 = synth.jitter(seed, it, B)
            render_out = renderer.render(rays_o, rays_d, near, far, background_rgb=background_rgb,
                                         cos_anneal_ratio=min(1.0, (it + 100) / 50000.0), depth_before_color=False,
                                         t_rand=torch.tensor(t1), t_rand_out=torch.tensor(t2))      # the two torch.rand draws, injected
            color_fine, gradient_error, weight_sum = render_out["color_fine"], render_out["gradient_error"], render_out["weight_sum"]
            color_error = (color_fine - true_rgb) * mask
            color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / mask_sum
            mask_loss = F.binary_cross_entropy(weight_sum.clip(1e-3, 1.0 - 1e-3), mask)
            loss = color_fine_loss + gradient_error * 0.1 + mask_loss * 0.0
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            losses.append(loss.item())
            del render_out
        for it, (got, want) in enumerate(zip(losses, fx["losses"])):
            assert abs(got - want) < 5e-5 * abs(want), (it, got, want)
        named = [("nerf." + n, p) for n, p in nerf_outside.named_parameters()] + [("sdf." + n, p) for n, p in sdf_network.named_parameters()] + \
                [("variance", deviation_network.variance)] + [("color." + n, p) for n, p in color_network.named_parameters()]
        for n, p in named:
            got = p.detach().cpu().reshape(-1)[torch.as_tensor(fx["p_idx/" + n]).cpu()].numpy()
            assert np.abs(got - fx["p_val/" + n]).max() < 2e-5, n
    finally:
        torch.set_default_tensor_type(torch.FloatTensor)


def test_adam_step_groups_match_torch_adam_across_depth_start():
    """torch.optim.Adam keeps a step per parameter and skips parameters without a gradient: the VDN head and the background
    network's dpt_linear have none until the depth-feature loss switches on (dpt_runner.py:239-243). The Trainer's fused Adam
    must give them their own step count (bias correction) from that moment: compared with torch.optim.Adam fed the
    Trainer's own gradients, 3 steps before and 3 after the switch."""
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 64, 11
    st = synth.make_all_states(seed, wdepth=True)
    rend = factory.build_renderer(wdepth=True, device=dev, states=st)
    tr = Trainer(rend, B, dev, conf=dict(extract_depth=True, depth_start_iter=2, warm_up_end=10))
    shadow = [torch.nn.Parameter(p.detach().clone()) for p in tr.params]
    opt = torch.optim.Adam(shadow, lr=1.0)
    cams = synth.make_cameras(seed)
    g = lambda x: torch.tensor(x).to(dev)
    feats = g(synth.uniform(seed, "adam/feats", (B, 96)).astype(np.float32))
    for it in range(6):
        o, d = synth.random_pixel_batch(seed, it, 0, B, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, it, B)
        lr = tr.learning_rate()
        depth_on = tr.iter_step > 2
        tr.train_step(g(o), g(d), g(near), g(far), g(synth.target_colors(o, d)), gt_feats=feats, t_rand=g(t1), t_rand_out=g(t2))
        grads = tr.engine.param_grads(clone=True)
        for gq in opt.param_groups:
            gq["lr"] = lr
        for i, (sp, gr) in enumerate(zip(shadow, grads)):
            sp.grad = None if (i in tr._depth_idx and not depth_on) else gr.view_as(sp).clone()
        opt.step()
        for i, (sp, p) in enumerate(zip(shadow, tr.params)):
            err = (sp.detach() - p.detach()).abs().max().item()
            assert err < 2e-6, (it, i, err)
    assert tr._depth_adam_steps == 3 and len(tr._depth_idx) > 0
    ck = tr.state_dict()
    steps = {int(v["step"]) for v in ck["optimizer"]["state"].values()}
    assert steps == {6, 3}
    tr2 = Trainer(factory.build_renderer(wdepth=True, device=dev, states=st), B, dev, conf=dict(extract_depth=True, depth_start_iter=2, warm_up_end=10))
    tr2.load_checkpoint(ck)
    assert tr2._depth_adam_steps == 3 and tr2.iter_step - tr2._step0() == 6
    assert torch.equal(tr2.exp_avg, tr.exp_avg) and torch.equal(tr2.param_flat, tr.param_flat)


def test_adam_groups_with_depth_before_color_and_a_step_without_depth_targets():
    """render(depth_before_color=True) feeds the VDN head's output to the colour network (renderer.py:247-248), so in the reference
    the head has a colour-loss gradient - and Adam steps - from iteration 0; only the background network's dpt_linear waits for
    the depth-feature loss (dpt_runner.py:239-243). And a step whose loss has no depth term (here: no gt_feats) leaves those
    parameters' .grad None, which torch.optim.Adam skips - no step on zero gradients, no step-count advance. Compared with
    torch.optim.Adam fed the Trainer's own gradients."""
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 64, 13
    st = synth.make_all_states(seed, wdepth=True, depth_before_color=True)
    rend = factory.build_renderer(wdepth=True, device=dev, states=st, depth_before_color=True)
    tr = Trainer(rend, B, dev, conf=dict(extract_depth=True, depth_start_iter=1, warm_up_end=10))
    dpt_ids = {id(p) for p in rend.nerf.dpt_linear.parameters()}
    assert {id(tr.params[i]) for i in tr._depth_idx} == dpt_ids            # the VDN head is NOT in the late group
    vdn_ids = {id(p) for p in rend.depth_network.parameters()}
    shadow = [torch.nn.Parameter(p.detach().clone()) for p in tr.params]
    start = [p.detach().clone() for p in tr.params]
    opt = torch.optim.Adam(shadow, lr=1.0)
    cams = synth.make_cameras(seed)
    g = lambda x: torch.tensor(x).to(dev)
    feats = g(synth.uniform(seed, "adam/feats", (B, 96)).astype(np.float32))
    for it in range(6):
        o, d = synth.random_pixel_batch(seed, it, 0, B, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, it, B)
        lr = tr.learning_rate()
        gt = None if it == 4 else feats                       # one step of the depth phase without depth targets
        depth_on = tr.iter_step > 1 and gt is not None
        tr.train_step(g(o), g(d), g(near), g(far), g(synth.target_colors(o, d)), gt_feats=gt, t_rand=g(t1), t_rand_out=g(t2))
        grads = tr.engine.param_grads(clone=True)
        for gq in opt.param_groups:
            gq["lr"] = lr
        for i, (sp, gr) in enumerate(zip(shadow, grads)):
            sp.grad = None if (i in tr._depth_idx and not depth_on) else gr.view_as(sp).clone()
        opt.step()
        for i, (sp, p) in enumerate(zip(shadow, tr.params)):
            err = (sp.detach() - p.detach()).abs().max().item()
            assert err < 2e-6, (it, i, err)
        if it == 1:      # the head moves from the first step with a non-zero learning rate (warm-up: lr = 0 at iteration 0), long
            # before the depth loss: its gradient comes through the colour network
            moved = [(p.detach() - s0).abs().max().item() for p, s0 in zip(tr.params, start) if id(p) in vdn_ids]
            assert min(moved) > 0.0
    assert tr._depth_adam_steps == 3          # iterations 2, 3, 5
    steps = {int(v["step"]) for v in tr.state_dict()["optimizer"]["state"].values()}
    assert steps == {6, 3}


@pytest.mark.parametrize("precision,wdepth", [("fp32", False), ("bf16", False), ("bf16", True)])
def test_render_under_grad_on_the_work_list_equals_every_sample(precision, wdepth):
    """render() under grad runs the training launches on the foreground work list and an inference launch of the SDF network on
    the list's complement (dpt_models/renderer.py::_RenderCoreFn): EVERY output - `gradients` and `cdf_fine` of the skipped
    samples included - bit for bit what the evaluation of every sample returns (VDN_RENDER_FG_COMPACT=0); parameter gradients of
    the reference's loss differ by the summation order of the weight-gradient GEMM only; and a loss that does reach the skipped
    samples' saves (on `gradients` / `cdf_fine`) gets the same gradients through the re-run forward."""
    import os
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    B, seed = 192, 2
    cams = synth.make_cameras(seed)
    o, d = synth.random_pixel_batch(seed, 1, 3, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, 0, B)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    true_rgb = tt(synth.target_colors(o, d))
    gt = tt(synth.uniform(seed, "rf/gt", (B, 96)))
    w1 = tt(synth.uniform(seed, "rf/w1", (B, 128, 3)) - 0.5)
    w2 = tt(synth.uniform(seed, "rf/w2", (B, 128)) - 0.5)

    def run(compact, aux_loss):
        os.environ["VDN_RENDER_FG_COMPACT"] = "1" if compact else "0"
        try:
            rend = factory.build_renderer(wdepth=wdepth, device=dev, states=synth.make_all_states(seed, wdepth=wdepth, variance=0.4),
                                          precision=precision)
            out = rend.render(tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=0.7,
                              t_rand=tt(t1), t_rand_out=tt(t2))
            loss = (out["color_fine"] - true_rgb).abs().sum() / B + 0.1 * out["gradient_error"]
            loss = loss + F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), torch.ones(B, 1, device=dev)) * 0.1
            if wdepth:
                loss = loss + 0.3 * (out["render_feats"] - gt).abs().sum() / B
            if aux_loss:
                loss = loss + (out["gradients"] * w1).sum() * 0.01 + (out["cdf_fine"] * w2).sum() * 0.05
            loss.backward()
            grads = torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1) for p in rend._all_parameters()])
            eng = next(iter(rend.__dict__["_engines"].values()))
            n_listed = int(eng.w["fg_active"][1])
            return {k: v.detach().clone() for k, v in out.items() if v is not None}, grads, n_listed, float(loss)
        finally:
            del os.environ["VDN_RENDER_FG_COMPACT"]

    for aux_loss in (False, True):
        ref, g_ref, n_ref, l_ref = run(False, aux_loss)
        out, g, n, l = run(True, aux_loss)
        assert n_ref == B * 128
        if aux_loss:
            assert n == B * 128                       # the backward re-ran the forward on every sample
        else:
            assert 0 < n < B * 128                    # the scene does skip samples
        assert set(out) == set(ref)
        for k in ref:
            assert torch.equal(out[k], ref[k]), k
        assert l == l_ref
        tol = 2e-5 if precision == "fp32" else 2e-3
        assert float((g - g_ref).abs().max()) <= tol * float(g_ref.abs().max()), (aux_loss, float((g - g_ref).abs().max()), float(g_ref.abs().max()))
        assert float(g_ref.abs().max()) > 0


def test_engines_of_alternating_batch_sizes_stay_alive():
    """The runner alternates between its training batch and the (ragged) chunks of its image loops: the training engines of the
    three most recently used batch sizes are kept, a fourth size evicts the least recently used one, and results do not depend on
    what was rendered in between."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(1, variance=0.4), precision="bf16")
    cams = synth.make_cameras(1)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)

    def run(B):
        o, d = synth.random_pixel_batch(1, 0, 2, B, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(1, 0, B)
        out = rend.render(tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=1.0,
                          t_rand=tt(t1), t_rand_out=tt(t2))
        return out["color_fine"].detach().clone()

    first = run(64)
    eng64 = rend.__dict__["_engines"][next(k for k in rend.__dict__["_engines"] if k[0] == 64)]
    run(37)
    again = run(64)
    engines = rend.__dict__["_engines"]
    assert sorted(k[0] for k in engines) == [37, 64] and any(e is eng64 for e in engines.values())
    assert torch.equal(first, again)
    run(20)
    run(11)                                                   # a fourth size: 37, the least recently used, goes
    assert sorted(k[0] for k in rend.__dict__["_engines"]) == [11, 20, 64]


@pytest.mark.parametrize("precision,wdepth", [("bf16", False), ("fp32", False), ("bf16", True)])
def test_graph_replayed_steps_equal_eager_steps(precision, wdepth, monkeypatch):
    """render() under grad from its second call on is a HIP-graph replay (dpt_models/renderer.py::_TrainPlan: sampler + forward,
    and the backward per adjoint pattern). Five optimizer steps with a changing cos_anneal_ratio, a changing background colour
    and injected jitter: every output of every step and every parameter after the last step are BIT-identical to the same steps
    with VDN_RENDER_GRAPHS=0 - and the replayed steps really were replays."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    cams = synth.make_cameras(5)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    B = 40

    def run(graphs):
        monkeypatch.setenv("VDN_RENDER_GRAPHS", "1" if graphs else "0")
        rend = factory.build_renderer(wdepth=wdepth, device=dev, states=synth.make_all_states(5, wdepth=wdepth, variance=0.45), precision=precision)
        params = rend._all_parameters()
        opt = torch.optim.Adam(params, lr=1e-3)
        outs = []
        for it in range(5):
            o, d = synth.random_pixel_batch(5, it, it % 3, B, cams=cams)
            near, far = synth.near_far_from_sphere(o, d)
            t1, t2 = synth.jitter(5, it, B)
            bg = torch.full((1, 3), 1.0 - 0.1 * it, device=dev)
            out = rend.render(tt(o), tt(d), tt(near), tt(far), background_rgb=bg, cos_anneal_ratio=0.2 * it,
                              t_rand=tt(t1), t_rand_out=tt(t2))
            mask = torch.ones(B, 1, device=dev)
            loss = (out["color_fine"] - tt(synth.target_colors(o, d))).abs().sum() / B + 0.1 * out["gradient_error"] \
                + 0.0 * F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
            if wdepth:
                loss = loss + 0.5 * out["render_feats"].abs().sum() / B
            opt.zero_grad()
            loss.backward()
            opt.step()
            outs.append({k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)})
        eng = next(iter(rend.__dict__["_engines"].values()))
        plans = eng.__dict__.get("_plans", {})
        return outs, [p.detach().clone() for p in params], plans

    ga, pa, plans = run(True)
    assert len(plans) == 1 and len(next(iter(plans.values())).bwd) == 1       # one forward graph, one backward graph (one adjoint pattern)
    gb, pb, none = run(False)
    assert not none
    for it, (a, b) in enumerate(zip(ga, gb)):
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), (it, k, float((a[k] - b[k]).abs().max()))
    assert all(torch.equal(x, y) for x, y in zip(pa, pb))


def test_graph_plan_draws_fresh_jitter_and_follows_forward_only_loops():
    """Without injected jitter the plan's sampler draws inside the graph: consecutive replays on the same rays differ (perturb = 1)
    and are equal with perturb_overwrite = 0; render() under grad without a backward in between (the runner's image loops,
    dpt_runner.py:439-445) replays as well; a loss that reads `gradients` falls back to the eager re-run of the forward."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(6, variance=0.4), precision="bf16")
    cams = synth.make_cameras(6)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    B = 48
    o, d = synth.random_pixel_batch(6, 0, 1, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    args = (tt(o), tt(d), tt(near), tt(far))
    bg = torch.ones(1, 3, device=dev)
    cols = [rend.render(*args, background_rgb=bg, cos_anneal_ratio=1.0)["color_fine"].detach().clone() for _ in range(4)]
    assert not torch.equal(cols[2], cols[3]) and float((cols[2] - cols[3]).abs().max()) < 0.2
    fix = [rend.render(*args, background_rgb=bg, cos_anneal_ratio=1.0, perturb_overwrite=0)["color_fine"].detach().clone() for _ in range(3)]
    assert torch.equal(fix[0], fix[1]) and torch.equal(fix[1], fix[2])          # eager, then two replays
    eng = next(iter(rend.__dict__["_engines"].values()))
    assert len(eng.__dict__["_plans"]) == 2
    # a loss on `gradients`: the adjoint reaches the samples the work list skipped - eager forward over every sample, eager backward
    out = rend.render(*args, background_rgb=bg, cos_anneal_ratio=1.0, perturb_overwrite=0)
    (out["gradients"].square().sum() + out["color_fine"].sum()).backward()
    g1 = [p.grad.clone() for p in rend._all_parameters()]
    os_env = __import__("os").environ
    os_env["VDN_RENDER_GRAPHS"] = "0"
    try:
        for p in rend._all_parameters():
            p.grad = None
        out = rend.render(*args, background_rgb=bg, cos_anneal_ratio=1.0, perturb_overwrite=0)
        (out["gradients"].square().sum() + out["color_fine"].sum()).backward()
    finally:
        del os_env["VDN_RENDER_GRAPHS"]
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, rend._all_parameters()))


def test_a_failed_graph_capture_falls_back_to_the_eager_launches(monkeypatch):
    """The graph plan is an optimisation: if capturing it fails, render() warns once for that configuration and keeps issuing the
    launches itself - same results."""
    import warnings
    from dpt_models import renderer as R
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(8, variance=0.4), precision="bf16")
    cams = synth.make_cameras(8)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    B = 32
    o, d = synth.random_pixel_batch(8, 0, 1, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    args = (tt(o), tt(d), tt(near), tt(far))
    bg = torch.ones(1, 3, device=dev)

    def boom(self, *a, **k):
        raise RuntimeError("capture refused (test)")
    monkeypatch.setattr(R._TrainPlan, "__init__", boom)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        cols = []
        for _ in range(4):
            out = rend.render(*args, background_rgb=bg, cos_anneal_ratio=1.0, perturb_overwrite=0)
            out["color_fine"].sum().backward()
            cols.append(out["color_fine"].detach().clone())
    assert sum("capturing render() as a HIP graph failed" in str(x.message) for x in w) == 1
    assert all(torch.equal(cols[0], c) for c in cols[1:])
    eng = next(iter(rend.__dict__["_engines"].values()))
    assert not eng.__dict__.get("_plans")


def test_outputs_of_render_under_grad_allow_inplace_ops():
    """render() under grad hands out plain tensors (they were views of one arena, and autograd refuses in-place ops
    on the views of a multi-output node): color_fine.clamp_() and weight_sum.clip_() work, backward() still reaches every
    parameter, and gradients equal those of the out-of-place forms. A saved output (`weights`) modified before backward() is
    still caught by autograd's version check."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(3, variance=0.4), precision="fp32")
    cams = synth.make_cameras(3)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    B = 24
    o, d = synth.random_pixel_batch(3, 0, 1, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(3, 0, B)
    params = rend._all_parameters()

    def grads(inplace):
        for p in params:
            p.grad = None
        out = rend.render(tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=1.0,
                          t_rand=tt(t1), t_rand_out=tt(t2))
        c, ws = out["color_fine"], out["weight_sum"]
        assert not c._is_view() and not ws._is_view() and not out["gradients"]._is_view()
        if inplace:
            c.mul_(0.5).clamp_(0.05, 0.45)
            ws.clip_(1e-3, 1.0 - 1e-3)
        else:
            c, ws = (c * 0.5).clamp(0.05, 0.45), ws.clip(1e-3, 1.0 - 1e-3)
        (c.sum() + ws.sum() + out["gradient_error"]).backward()
        return [p.grad.clone() for p in params]
    ga, gb = grads(True), grads(False)
    assert all(torch.equal(a, b) for a, b in zip(ga, gb))
    assert sum(float(g.abs().sum()) for g in ga) > 0
    out = rend.render(tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=1.0)
    out["weights"].mul_(2.0)                    # saved for the node's backward: must not pass silently
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out["color_fine"].sum().backward()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("kind", ["all_listed", "none_listed", "single_ray"])
def test_render_under_grad_with_an_empty_list_or_an_empty_complement(kind, precision):
    """The edges of the autograd node's work list: rays through the centre (every inside sample within the relaxed sphere: the
    complement launch has no rows), rays that pass the sphere at a distance (no sample listed: the training launches, the
    backward and the GEMM's SDF entries have no rows) and a batch of one ray - outputs bit for bit those of the evaluation of every
    sample, finite gradients equal to its to the GEMM's summation order."""
    import os
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    B = 1 if kind == "single_ray" else 8
    d = rng.standard_normal((B, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if kind == "none_listed":
        perp = np.cross(d, np.array([0.3, -0.5, 0.8], np.float32))
        perp /= np.linalg.norm(perp, axis=1, keepdims=True)
        o = (perp * 2.5 - d * 3.0).astype(np.float32)          # closest approach 2.5 from the origin
    else:
        o = (-d * 3.0 + rng.standard_normal((B, 3)).astype(np.float32) * (0.01 if kind == "all_listed" else 0.3)).astype(np.float32)
    near, far = synth.near_far_from_sphere(o, d)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    t1, t2 = synth.jitter(3, 0, B)

    def run(compact):
        os.environ["VDN_RENDER_FG_COMPACT"] = compact
        try:
            rend = factory.build_renderer(device=dev, states=synth.make_all_states(3, variance=0.4), precision=precision)
            out = rend.render(tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=0.3,
                              t_rand=tt(t1), t_rand_out=tt(t2))
            ((out["color_fine"] - 0.3).abs().sum() / B + 0.1 * out["gradient_error"]).backward()
            eng = next(iter(rend.__dict__["_engines"].values()))
            grads = torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1) for p in rend._all_parameters()])
            return {k: v.detach().clone() for k, v in out.items() if v is not None}, grads, int(eng.w["fg_active"][1])
        finally:
            del os.environ["VDN_RENDER_FG_COMPACT"]

    ref, g_ref, _ = run("0")
    out, g, n = run("1")
    if kind == "all_listed":
        assert n == B * 128
    if kind == "none_listed":
        assert n == 0
    for k in ref:
        assert torch.equal(out[k], ref[k]), k
    assert torch.isfinite(g).all() and torch.isfinite(g_ref).all()
    assert float((g - g_ref).abs().max()) <= (2e-3 if precision == "bf16" else 2e-5) * float(g_ref.abs().max()) + 1e-12
