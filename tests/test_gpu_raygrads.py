"""Differentiable rays (learnable poses, reference poses.py:198-208 + dpt_runner.py:201-217): d loss / d rays_o, rays_d, near,
far through NeuSRenderer.render - the hand-written ray adjoint (vdn_ray_adjoint + the networks' input adjoints) - against
the fp64 oracle's autograd on the same inputs and the reference's own autograd (tests/golden/raygrad.npz). The reference's
graph: the outside depths depend on far (renderer.py:359); the inside depths depend on (near, far) only when there is no
importance sampling - otherwise they leave a no_grad block as constants (367-386) and d loss / d near is zero."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_grads import _loss, g

pytestmark = pytest.mark.gpu


def _oracle(fx, dtype=torch.float64):
    import oracle.neus_oracle as orc
    from vdn_train import synth
    wdepth = bool(fx["wdepth"])
    st = synth.make_all_states(int(fx["seed"]), wdepth=wdepth, variance=float(fx["variance"]))
    nets = orc.nets_from_numpy(st, dtype=dtype, requires_grad=False)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=dtype)
    leaves = [tt(fx[k]).requires_grad_(True) for k in ("rays_o", "rays_d", "near", "far")]
    o, d, near, far = leaves
    conf = orc.RendererConf(n_importance=int(fx["n_importance"]))
    perturb = -1 if fx["perturb"] > 0 else 0
    # with importance sampling the reference's inside depths come out of a no_grad block (renderer.py:367-386): constants
    z_inj = tt(fx["z_vals_inside"]) if fx["n_importance"] > 0 else None
    out = orc.render(nets, o, d, near, far, conf, perturb_overwrite=perturb,
                     background_rgb=torch.ones(1, 3, dtype=dtype) if fx["white"] else None,
                     cos_anneal_ratio=float(fx["cos_anneal"]), t_rand=tt(fx["t_rand"]), t_rand_out=tt(fx["t_rand_out"]), z_vals_inject=z_inj)
    loss = _loss(out, tt(fx["true_rgb"]), tt(fx["gt_feats"]) if wdepth else None, wdepth)
    grads = torch.autograd.grad(loss, leaves, allow_unused=True)
    return loss.item(), [torch.zeros_like(x) if gr is None else gr.detach() for x, gr in zip(leaves, grads)]


def _gpu(fx, dev, precision="fp32", param_grads=True):
    from vdn_train import synth, factory
    wdepth = bool(fx["wdepth"])
    st = synth.make_all_states(int(fx["seed"]), wdepth=wdepth, variance=float(fx["variance"]))
    rend = factory.build_renderer(wdepth=wdepth, device=dev, states=st, n_importance=int(fx["n_importance"]), precision=precision)
    if not param_grads:
        for p in rend._all_parameters():
            p.requires_grad_(False)
    leaves = [g(fx[k], dev).requires_grad_(True) for k in ("rays_o", "rays_d", "near", "far")]
    o, d, near, far = leaves
    out = rend.render(o, d, near, far, perturb_overwrite=(-1 if fx["perturb"] > 0 else 0),
                      background_rgb=torch.ones(1, 3, device=dev) if fx["white"] else None,
                      cos_anneal_ratio=float(fx["cos_anneal"]), t_rand=g(fx["t_rand"], dev), t_rand_out=g(fx["t_rand_out"], dev),
                      z_vals_inject=g(fx["z_vals_inside"], dev) if fx["n_importance"] > 0 else None)
    loss = _loss(out, g(fx["true_rgb"], dev), g(fx["gt_feats"], dev) if wdepth else None, wdepth)
    loss.backward()
    return loss.item(), [(torch.zeros_like(x) if x.grad is None else x.grad).detach().cpu().double() for x in leaves], rend


@pytest.mark.parametrize("name,param_grads", [("white_n64_v03", True), ("white_v03_c05_det", True), ("white_v03_c0", False), ("wdepth_v03_c05", True)])
def test_ray_gradients_vs_oracle_fp64(golden, name, param_grads):
    fx = golden(name)
    dev = torch.device("cuda:0")
    loss_o, ref = _oracle(fx)
    _, ref32 = _oracle(fx, torch.float32)          # what an fp32 autograd reaches: the floor for cancellation-dominated entries
    loss_g, got, _ = _gpu(fx, dev, param_grads=param_grads)
    assert abs(loss_g - loss_o) <= 2e-4 * abs(loss_o)
    report, bad = [], []
    for nm, a, b, b32 in zip(("rays_o", "rays_d", "near", "far"), got, ref, ref32):
        scale = b.abs().max().item()
        err = (a - b).abs().max().item() / (scale + 1e-30)
        floor = (b32.double() - b).abs().max().item() / (scale + 1e-30)
        report.append("%s: rel-to-max err %.2e, fp32-autograd floor %.2e (|g|max %.2e)" % (nm, err, floor, scale))
        if err > max(1e-3, 3 * floor):
            bad.append(nm)
    print("\n".join(report))
    assert not bad, "\n".join(report)


def test_ray_gradients_bf16_and_parameter_gradients_unchanged(golden):
    """The throughput path produces the same ray gradients to bf16 accuracy, and asking for them does not change the
    parameter gradients (same kernels, the adjoint outputs are extra stores)."""
    fx = golden("white_v03_c05_det")
    dev = torch.device("cuda:0")
    _, ref = _oracle(fx)
    _, got, rend = _gpu(fx, dev, precision="bf16")
    rels = {nm: (a - b).norm().item() / (b.norm().item() + 1e-30) for nm, a, b in zip(("rays_o", "rays_d", "near", "far"), got, ref)}
    print(rels)
    # near / far gradients are sums over the coarse depths' adjoints with heavy cancellation (|g| ~ 1e-3 against terms ~ 1e-1)
    assert rels["rays_o"] < 5e-2 and rels["rays_d"] < 5e-2 and rels["near"] < 0.5 and rels["far"] < 0.5, rels
    with_rays = [p.grad.clone() for p in rend._all_parameters()]
    from test_gpu_grads import _gpu_grads
    import os
    # (ray gradients need every inside sample evaluated; without them render() takes the foreground work list, which changes the
    # weight-gradient GEMM's summation order: the like-for-like comparison is the evaluation of every sample)
    os.environ["VDN_RENDER_FG_COMPACT"] = "0"
    try:
        _, named, _ = _gpu_grads(fx, dev, precision="bf16")
    finally:
        del os.environ["VDN_RENDER_FG_COMPACT"]
    for a, (n, p) in zip(with_rays, named):
        assert torch.equal(a, p.grad), n


def test_learnpose_gradients_reach_the_pose_parameters():
    """dpt_runner.py:197-217 with a learnable camera: LearnPose -> rays (poses.py:198-208) -> near_far_from_sphere
    (dataset.py:111-118) -> render -> loss; d loss / d (r, t) against the same chain on the fp64 oracle."""
    import oracle.neus_oracle as orc
    from dpt_models.poses import LearnPose, LearnIntrin
    from dpt_models import lie_group_helper as lg
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    seed, B = 2, 24
    cams = synth.make_cameras(seed)
    c2w0 = torch.tensor(np.asarray(cams[:3], np.float32))
    H, W, focal = synth.H, synth.W_IMG, 1111.0
    px = torch.tensor(np.floor(synth.uniform(seed, "lp/x", (B,)) * 400 + 200).astype(np.float32))
    py = torch.tensor(np.floor(synth.uniform(seed, "lp/y", (B,)) * 400 + 200).astype(np.float32))
    r0 = torch.tensor(synth.uniform(seed, "lp/r", (3, 3)).astype(np.float32) * 0.02 - 0.01)
    t0 = torch.tensor(synth.uniform(seed, "lp/t", (3, 3)).astype(np.float32) * 0.02 - 0.01)
    true_rgb = synth.uniform(seed, "lp/rgb", (B, 3)).astype(np.float32)
    cam = 1

    def rays_from(pose, K, dtype):
        p = torch.stack([px, py, torch.ones_like(py)], dim=-1).to(dtype).to(pose.device)
        p = torch.matmul(torch.inverse(K)[None, :3, :3], p[:, :, None]).squeeze(-1)
        v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
        v = torch.matmul(pose[None, :3, :3], v[:, :, None]).squeeze(-1)
        return pose[None, :3, 3].expand(v.shape), v

    # ---- MI355X: the package's modules
    pose_net = LearnPose(3, True, True, init_c2w=c2w0.clone()).to(dev)
    with torch.no_grad():
        pose_net.r.copy_(r0.to(dev)); pose_net.t.copy_(t0.to(dev))
    intrin = LearnIntrin(H, W, req_grad=False, order=2, init_focal=torch.tensor(focal)).to(dev)
    st = synth.make_all_states(seed, wdepth=False, variance=0.3)
    rend = factory.build_renderer(wdepth=False, device=dev, states=st, n_importance=0)
    ro, rv = rays_from(pose_net(cam), intrin(), torch.float32)
    a = torch.sum(rv ** 2, dim=-1, keepdim=True); b = 2.0 * torch.sum(ro * rv, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    out = rend.render(ro, rv, mid - 1.0, mid + 1.0, perturb_overwrite=0, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=0.5)
    loss = _loss(out, g(true_rgb, dev), None, False)
    loss.backward()
    got = [pose_net.r.grad[cam].cpu().double(), pose_net.t.grad[cam].cpu().double()]
    assert pose_net.r.grad[0].abs().max() == 0 and pose_net.t.grad[2].abs().max() == 0      # other cameras untouched

    # ---- fp64 oracle with the same chain
    dt = torch.float64
    r = r0.to(dt).clone().requires_grad_(True)
    t = t0.to(dt).clone().requires_grad_(True)
    K = intrin().cpu().to(dt)

    def exp64(rv_):
        zero = torch.zeros(1, dtype=dt)
        Kx = torch.stack([torch.cat([zero, -rv_[2:3], rv_[1:2]]), torch.cat([rv_[2:3], zero, -rv_[0:1]]), torch.cat([-rv_[1:2], rv_[0:1], zero])])
        n = rv_.norm() + 1e-15
        return torch.eye(3, dtype=dt) + (torch.sin(n) / n) * Kx + ((1 - torch.cos(n)) / n ** 2) * (Kx @ Kx)
    c2w = torch.cat([torch.cat([exp64(r[cam]), t[cam].unsqueeze(1)], dim=1), torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=dt)], dim=0) @ c2w0[cam].to(dt)
    ro64, rv64 = rays_from(c2w, K, dt)
    near64, far64 = orc.near_far_from_sphere(ro64, rv64)
    nets = orc.nets_from_numpy(st, dtype=dt, requires_grad=False)
    oo = orc.render(nets, ro64, rv64, near64, far64, orc.RendererConf(n_importance=0), perturb_overwrite=0,
                    background_rgb=torch.ones(1, 3, dtype=dt), cos_anneal_ratio=0.5)
    lo = _loss(oo, torch.tensor(true_rgb, dtype=dt), None, False)
    gr, gt = torch.autograd.grad(lo, [r, t])
    assert abs(loss.item() - lo.item()) <= 2e-4 * abs(lo.item())
    for nm, a_, b_ in (("r", got[0], gr[cam]), ("t", got[1], gt[cam])):
        rel = (a_ - b_).abs().max().item() / (b_.abs().max().item() + 1e-30)
        assert rel < 2e-3, (nm, rel, a_, b_)
    # the rotation helper itself against its fp64 restatement
    assert (lg.Exp(r0[1]).double() - exp64(r0[1].to(dt))).abs().max() < 1e-6


@pytest.mark.parametrize("tag", ["n64", "full"])
def test_ray_gradients_vs_reference_autograd(golden, tag):
    """tests/golden/raygrad.npz: the reference's own d loss / d (rays_o, rays_d, near, far) (make_golden.py: raygrad_fixture).
    Both sides are fp32; near / far gradients are cancellation-dominated sums (their fp32 floor is ~5e-3 of the largest entry)."""
    from vdn_train import synth, factory
    fx = {k.split("/", 1)[1]: v for k, v in golden("raygrad").items() if k.startswith(tag + "/")}
    dev = torch.device("cuda:0")
    st = synth.make_all_states(int(fx["seed"]), wdepth=False, variance=0.3)
    rend = factory.build_renderer(wdepth=False, device=dev, states=st, n_importance=int(fx["n_importance"]))
    leaves = [g(fx[k], dev).requires_grad_(True) for k in ("rays_o", "rays_d", "near", "far")]
    out = rend.render(*leaves, perturb_overwrite=-1, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=float(fx["cos_anneal"]),
                      t_rand=g(fx["t_rand"], dev), t_rand_out=g(fx["t_rand_out"], dev),
                      z_vals_inject=g(fx["z_vals_inside"], dev) if fx["n_importance"] > 0 else None)
    assert np.abs(out["color_fine"].detach().cpu().numpy() - fx["color_fine"]).max() < 2e-4
    loss = _loss(out, g(fx["true_rgb"], dev), None, False)
    assert abs(loss.item() - float(fx["loss"])) <= 2e-4 * abs(float(fx["loss"]))
    loss.backward()
    if not int(fx["near_in_graph"]):
        assert leaves[2].grad is None or leaves[2].grad.abs().max().item() == 0.0
    for nm, x, tol in (("rays_o", leaves[0], 1e-3), ("rays_d", leaves[1], 1e-3), ("near", leaves[2], 3e-2), ("far", leaves[3], 3e-2)):
        ref = fx["grad_" + nm]
        got = np.zeros_like(ref) if x.grad is None else x.grad.detach().cpu().numpy()
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= tol * scale + 1e-12, (nm, np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_pose_refinement_recovers_a_perturbed_camera(precision):
    """What the learnable poses are for (the reference's `*_learn_*` runs): with the networks frozen, targets rendered from the
    true camera and a LearnPose that starts at the unperturbed initial pose, Adam on (r, t) through the ray adjoint finds the
    true camera (tests/probes/pose_refine.py prints the trajectory)."""
    from vdn_train import synth, factory
    from dpt_models.poses import LearnPose
    from dpt_models.lie_group_helper import make_c2w
    dev = torch.device("cuda:0")
    seed, B, steps, cam = 0, 512, 200, 2
    st = synth.make_all_states(seed, wdepth=False, variance=0.3)
    rend = factory.build_renderer(device=dev, states=st, precision=precision)
    for p in rend._all_parameters():
        p.requires_grad_(False)
    cams = torch.tensor(np.asarray(synth.make_cameras(seed)[:4], np.float32)).to(dev)
    Kinv = torch.tensor(synth.intrinsics_inv().astype(np.float32)).to(dev)
    pose_net = LearnPose(4, True, True, init_c2w=cams.clone()).to(dev)
    true_r, true_t = torch.tensor([0.02, -0.015, 0.01], device=dev), torch.tensor([0.05, -0.04, 0.03], device=dev)
    true_pose = make_c2w(true_r, true_t) @ cams[cam]

    def rays(pose, px, py):
        p = torch.matmul(Kinv[None], torch.stack([px, py, torch.ones_like(py)], dim=-1)[:, :, None]).squeeze(-1)
        v = torch.matmul(pose[None, :3, :3], (p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True))[:, :, None]).squeeze(-1)
        o = pose[None, :3, 3].expand(v.shape)
        mid = 0.5 * (-(2.0 * (o * v).sum(-1, keepdim=True))) / (v * v).sum(-1, keepdim=True)        # dataset.py:111-118
        return o, v, mid - 1.0, mid + 1.0

    kw = dict(perturb_overwrite=0, background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=1.0)
    opt = torch.optim.Adam([pose_net.r, pose_net.t], lr=2e-3)
    gen = torch.Generator(device="cpu").manual_seed(1)
    err0 = (true_r.norm().item(), true_t.norm().item())
    first = None
    for it in range(steps):
        px = (torch.rand(B, generator=gen) * 500 + 150).floor().to(dev)
        py = (torch.rand(B, generator=gen) * 500 + 150).floor().to(dev)
        with torch.no_grad():
            target = rend.render(*rays(true_pose, px, py), **kw)["color_fine"]
        loss = (rend.render(*rays(pose_net(cam), px, py), **kw)["color_fine"] - target).abs().mean()
        first = loss.item() if first is None else first
        opt.zero_grad()
        loss.backward()
        opt.step()
    er = (pose_net.r[cam].detach() - true_r).norm().item()
    et = (pose_net.t[cam].detach() - true_t).norm().item()
    assert loss.item() < 0.05 * first and er < 0.1 * err0[0] and et < 0.1 * err0[1], (first, loss.item(), er, et, err0)
    assert pose_net.r.grad[0].abs().max() == 0 and pose_net.t.grad[3].abs().max() == 0


def test_learnable_rays_equal_the_fixed_pose_generator_and_carry_the_graph():
    """dpt_models.poses.LearnableRays (the learnable branch of RaysGenerator, poses.py:168-212) with a zero pose delta and the
    same intrinsics produces the fixed-pose generator's rays (vdn_gen_rays) and pixel data, with a graph to the pose."""
    from vdn_train import synth
    from vdn_train.rays import RaysGenerator
    from dpt_models.poses import LearnPose, LearnIntrin, LearnableRays
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(5)
    n, H, W, C = 3, 48, 64, 4
    images, masks, feats = rng.rand(n, H, W, 3).astype(np.float32), (rng.rand(n, H, W, 3) > 0.3).astype(np.float32), rng.rand(n, H, W, C).astype(np.float32)
    cams = np.asarray(synth.make_cameras(5, n=n), np.float32)
    intrin = LearnIntrin(H, W, req_grad=False, order=2, init_focal=torch.tensor(70.0)).to(dev)
    K = intrin().cpu().numpy()
    fixed = RaysGenerator(images, masks, cams, K, depth_feats=feats, device=dev)
    pose_net = LearnPose(n, True, True, init_c2w=torch.tensor(cams)).to(dev)
    lr = LearnableRays(pose_net, intrin, fixed)
    px, py = rng.randint(0, W, 96).astype(np.float32), rng.randint(0, H, 96).astype(np.float32)
    a = lr.gen_random_rays_at(1, 96, pixels=(px, py))
    b = fixed.gen_random_rays_at(1, 96, pixels=(px, py))
    assert a.shape == b.shape == (96, 10 + C) and a.requires_grad
    assert (a[:, :6] - b[:, :6]).abs().max().item() < 2e-6 and torch.equal(a[:, 6:], b[:, 6:])
    a[:, :6].sum().backward()
    assert pose_net.t.grad[1].abs().sum() > 0 and pose_net.r.grad[1].abs().sum() > 0 and pose_net.r.grad[0].abs().sum() == 0
    o, v = lr.gen_rays_at(2, resolution_level=2)
    o2, v2 = fixed.gen_rays_at(2, resolution_level=2)
    assert o.shape == (H // 2, W // 2, 3) and (v - v2).abs().max().item() < 2e-6 and (o - o2).abs().max().item() < 1e-6


@pytest.mark.gpu
def test_learnable_rays_on_the_reference_shipped_cameras():
    """tests/golden/pnf_rays.npz: the reference's learnable ray branch (RaysGenerator.gen_random_rays_at with learnable=True,
    poses.py:189-212) run by the reference itself on the cameras IT SHIPS (pretrained-models/*/*/pnf_300000.pth, loaded into its
    own LearnPose / LearnIntrin), with recorded pixels, and its autograd's d loss / d (r, t) for a fixed linear loss on the rays.
    The same state_dicts in dpt_models.poses + LearnableRays on the device: rays, pixel gathers and pose gradients."""
    from vdn_train.rays import RaysGenerator
    from dpt_models.poses import LearnPose, LearnIntrin, LearnableRays
    fx = load_golden("pnf_rays")
    dev = torch.device("cuda:0")
    H, W = int(fx["H"]), int(fx["W"])
    worst = {"o": 0.0, "d": 0.0, "gr": 0.0, "gt": 0.0}
    for tag in fx["names"]:
        tag = str(tag)
        n = fx[tag + "/r"].shape[0]
        pose = LearnPose(n, True, True, torch.zeros(n, 4, 4))
        pose.load_state_dict({k: torch.tensor(fx["%s/%s" % (tag, k)]) for k in ("init_c2w", "r", "t")})
        intr = LearnIntrin(H, W, req_grad=True)
        intr.load_state_dict({"fx": torch.tensor(fx[tag + "/fx"])})
        pose, intr = pose.to(dev), intr.to(dev)
        # the constructor's image branch (poses.py:117-122: RGBA composited on white) on the three stored images
        bgra = fx[tag + "/bgra"].astype(np.float64) / 255.0
        img, a = bgra[..., :3], bgra[..., 3:]
        img = (img * a + (1 - a)).astype(np.float32)
        fixed = RaysGenerator(img, a.astype(np.float32), fx[tag + "/c2w"][:3], fx[tag + "/intrinsic"], device=dev)
        lr = LearnableRays(pose, intr, fixed)
        for idx in (0, 1, 2):
            k = "%s/cam%d" % (tag, idx)
            want = fx[k + "/data"]
            pose.zero_grad()
            got = lr.gen_random_rays_at(idx, want.shape[0], pixels=(fx[k + "/pixels_x"].astype(np.float32), fx[k + "/pixels_y"].astype(np.float32)))
            assert got.shape == want.shape == (16, 11)
            g = got.detach().cpu().numpy()
            worst["o"] = max(worst["o"], float(np.abs(g[:, :3] - want[:, :3]).max()))
            worst["d"] = max(worst["d"], float(np.abs(g[:, 3:6] - want[:, 3:6]).max()))
            assert np.array_equal(g[:, 6:], want[:, 6:]), (tag, idx)          # mask | colour | the zero feature column
            loss = (got[:, :6] * torch.tensor(fx[k + "/loss_weights"], device=dev)).sum()
            assert abs(float(loss) - float(fx[k + "/loss"])) < 2e-5 * max(1.0, abs(float(fx[k + "/loss"])))
            loss.backward()
            gr, gt = pose.r.grad.cpu().numpy(), pose.t.grad.cpu().numpy()
            worst["gr"] = max(worst["gr"], float(np.abs(gr[idx] - fx[k + "/grad_r"]).max() / (np.abs(fx[k + "/grad_r"]).max() + 1e-30)))
            worst["gt"] = max(worst["gt"], float(np.abs(gt[idx] - fx[k + "/grad_t"]).max() / (np.abs(fx[k + "/grad_t"]).max() + 1e-30)))
            assert all(np.abs(gr[i]).sum() == 0 for i in range(n) if i != idx)
    # origins: the translation column itself; directions: one normalise + two 3x3 products in fp32
    assert worst["o"] < 1e-6 and worst["d"] < 2e-6, worst
    assert worst["gr"] < 2e-5 and worst["gt"] < 2e-5, worst
