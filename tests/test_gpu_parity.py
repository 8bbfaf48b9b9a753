"""GPU parity: the HIP path (through the dpt_models boundary -> C ABI) against the oracle and the
reference's golden vectors. Tolerances: 1e-5 rel single stages, 1e-4 rel per-ray end-to-end and
per-sample with injected z (BASELINE north-star 1e-4 rel fp32)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, relelem, relmax

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def env(dev):
    import oracle.neus_oracle as orc
    from vdn_train import synth, factory
    cache = {}

    def get(seed, wdepth, variance, **kw):
        key = (seed, wdepth, variance, tuple(sorted(kw.items())))
        if key not in cache:
            st = synth.make_all_states(seed, wdepth=wdepth, variance=variance)
            cache[key] = (factory.build_renderer(wdepth=wdepth, device=dev, states=st, **kw), orc.nets_from_numpy(st), st)
        return cache[key]
    return get


def g(x, dev):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev)


def test_library_loaded_is_in_tree():
    from vdn_hip import lib
    lib.load()
    assert "vdn-nerf_amd/vdn_hip/libvdn_render.so" in lib.LIB_PATH.replace("\\", "/")


@pytest.mark.parametrize("P", [1, 31, 96, 129, 1000])
def test_sdf_network_stage(env, dev, golden, P):
    import oracle.neus_oracle as orc
    from vdn_train import synth
    rend, nets, _ = env(3, True, 0.3)
    if P == 96:
        fx = golden("stages")
        pts = torch.tensor(fx["pts"])
    else:
        pts = torch.tensor(((synth.uniform(11, "t/pts%d" % P, (P, 3)) * 2 - 1) * 1.1).astype(np.float32))
    out, grad = orc.sdf_forward(nets.sdf, pts, nets.sdf_conf, with_gradient=True)
    got = rend.sdf_network(pts.to(dev)).cpu()
    assert got.shape == (P, 257)
    assert relmax(got.numpy(), out.numpy()) < 1e-5
    assert relmax(rend.sdf_network.sdf(pts.to(dev)).detach().cpu().numpy(), out[:, :1].numpy()) < 1e-5
    gg = rend.sdf_network.gradient(pts.to(dev)).cpu()
    assert gg.shape == (P, 1, 3)
    assert relmax(gg[:, 0].numpy(), grad.numpy()) < 1e-5
    if P == 96:    # the reference's own vectors
        assert relmax(got.numpy(), fx["sdf_out"]) < 1e-5
        assert relmax(gg[:, 0].numpy(), fx["sdf_grad"]) < 1e-5


def test_rendering_and_nerf_stage(env, dev, golden):
    fx = golden("stages")
    rend, nets, _ = env(3, True, 0.3)
    pts, dirs = g(fx["pts"], dev), g(fx["dirs"], dev)
    feat, grad = g(fx["sdf_out"][:, 1:], dev), g(fx["sdf_grad"], dev)
    col = rend.color_network(pts, grad, dirs, feat).detach().cpu().numpy()
    vdn = rend.depth_network(pts, grad, dirs, feat).detach().cpu().numpy()
    assert col.shape == (96, 3) and vdn.shape == (96, 96)
    assert relmax(col, fx["color"]) < 1e-5
    assert relmax(vdn, fx["vdn"]) < 1e-5
    a, rgb, ft = rend.nerf(g(fx["pts4"], dev), dirs)
    assert a.shape == (96, 1)
    assert relmax(a.detach().cpu().numpy(), fx["nerf_alpha"]) < 1e-5
    assert relmax(rgb.detach().cpu().numpy(), fx["nerf_rgb"]) < 1e-5
    assert relmax(ft.detach().cpu().numpy(), fx["nerf_feat"]) < 1e-5


def test_standalone_embedder_vs_reference_vectors(dev, golden):
    """a1: Embedder.embed as a callable (embedder.py:27-36) against the reference's own encoding vectors, for the three
    (d, L) the shipped configurations use; order [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]."""
    from dpt_models.embedder import get_embedder
    fx = golden("stages")
    pts = fx["pts"]
    for key_out, d, L in (("pe_6_3", 3, 6), ("pe_4_3", 3, 4), ("pe_10_4", 4, 10)):
        x = pts if d == 3 else np.concatenate([pts, pts[:, :1] * 0.5], -1)        # the inputs tests/golden/make_golden.py fed
        fn, out_dim = get_embedder(L, d)
        assert out_dim == d * (1 + 2 * L)
        got = fn(g(x, dev)).cpu().numpy()
        assert got.shape == fx[key_out].shape
        assert np.abs(got - fx[key_out]).max() < 2e-6, key_out
    fn, _ = get_embedder(6, 3)
    x = torch.rand(5, 7, 3, device=dev)
    ref = torch.cat([x] + [f(x * 2.0 ** k) for k in range(6) for f in (torch.sin, torch.cos)], -1)
    assert fn(x).shape == (5, 7, 39) and (fn(x) - ref).abs().max().item() < 1e-5
    with pytest.raises(RuntimeError):
        fn(torch.zeros(2, 3))


def test_sample_pdf_vs_reference_vectors(dev, golden):
    """a10: the inverse-CDF half of vdn_upsample_round on the reference's OWN sample_pdf vectors (renderer.py:44-74, det=True):
    spdf_bins / spdf_w -> spdf_out, including rows whose CDF is flat over several bins (the denom < 1e-5 -> 1 branch of
    renderer.py:70). The kernel accumulates the CDF in double like ATen's CPU cumsum: bit-level agreement is not promised,
    1e-6 on z in [0, 1] is."""
    from vdn_hip import lib
    from dpt_models.fields import _stream
    g = golden("stages")
    bins, w, ref = g["spdf_bins"], g["spdf_w"], g["spdf_out"]
    B, M = bins.shape
    assert w.shape == (B, M - 1) and ref.shape == (B, 16)
    # the fixture must exercise the flat-CDF branch: some row has consecutive (near-)zero weights
    flat_rows = int(((w[:, 1:] + w[:, :-1]) < 1e-6).any(axis=1).sum())
    assert flat_rows >= 1
    zb, wb = torch.tensor(bins).to(dev), torch.tensor(w).to(dev)
    u = torch.linspace(0.5 / 16, 1.0 - 0.5 / 16, 16, device=dev)
    out = torch.full((B, 16), -1.0, device=dev)
    a = lib.VdnUpsampleArgs()
    a.z, a.weights, a.u, a.new_z = zb.data_ptr(), wb.data_ptr(), u.data_ptr(), out.data_ptr()
    a.B, a.M, a.ld, a.w_ld, a.n_imp, a.inv_s = B, M, M, M - 1, 16, 64.0
    lib.call("vdn_upsample_round", a, _stream())
    got = out.cpu().numpy()
    err = np.abs(got - ref)
    assert err.max() < 1e-6, (err.max(), np.argwhere(err > 1e-6)[:5])
    # argument checking of the new form
    a.w_ld = M - 2
    with pytest.raises(lib.VdnError):
        lib.call("vdn_upsample_round", a, _stream())


@pytest.mark.parametrize("name", ["white_v03_c0", "white_v065_c1", "wdepth_v03_c05"])
def test_upsample_round_by_round_vs_reference(env, dev, golden, name):
    """a11: one up-sampling round at a time on the reference's own intermediate vectors: the reference's z after round i-1
    plus the SDF of those z (evaluated by the fp32 kernel, itself pinned to 1e-5 by test_sdf_network_stage) -> the 16 new z
    -> merged -> against the reference's z after round i (renderer.py:147-207). The inverse CDF is ill-conditioned where
    the CDF is flat: those entries are COUNTED, not skipped - at most 2 % of the samples may sit beyond
    1e-5 / 1e-4 against the reference's own fp32-vs-fp64 floor, and none may be off by more than one bin width."""
    from vdn_hip import lib
    from dpt_models.fields import _stream
    g = golden(name)
    rend, _, _ = env(int(g["seed"]), bool(g["wdepth"]), float(g["variance"]))       # fp32 kernels (the parity path)
    o, d = torch.tensor(g["rays_o"]).to(dev), torch.tensor(g["rays_d"]).to(dev)
    B = o.shape[0]
    u = torch.linspace(0.5 / 16, 1.0 - 0.5 / 16, 16, device=dev)
    n_bad = n_bad4 = n_all = 0
    for i in range(4):
        M = 64 + 16 * i
        if i > 0:
            z_prev = torch.tensor(g["z_round%d" % (i - 1)]).to(dev)
        else:       # the reference's coarse grid + its one jitter per ray (renderer.py:335-336, 347-349)
            near, far = torch.tensor(g["near"]).to(dev), torch.tensor(g["far"]).to(dev)
            z_prev = near + (far - near) * torch.linspace(0.0, 1.0, 64, device=dev)[None, :]
            z_prev = z_prev + (torch.tensor(g["t_rand"]).to(dev).view(B, 1) - 0.5) * (2.0 / 64)
        z_prev = z_prev.contiguous()
        if i == 0:      # the reference's own SDF values at the coarse z: this round's inputs are bit-identical to the reference's
            sdf = torch.tensor(g["coarse_sdf"]).to(dev).contiguous()
        else:
            with torch.no_grad():
                sdf = rend.sdf_network._run(0, rays=(o, d, z_prev)).view(B, M)
        new = torch.empty(B, 16, device=dev)
        a = lib.VdnUpsampleArgs()
        a.rays_o, a.rays_d, a.z, a.sdf, a.u, a.new_z = o.data_ptr(), d.data_ptr(), z_prev.data_ptr(), sdf.data_ptr(), u.data_ptr(), new.data_ptr()
        a.B, a.M, a.ld, a.n_imp, a.inv_s = B, M, M, 16, 64.0 * 2 ** i
        lib.call("vdn_upsample_round", a, _stream())
        ref = g["z_round%d" % i]
        # the reference's 16 new z of this round = its merged row minus the previous row (old values are carried unchanged)
        zp = z_prev.cpu().numpy()
        got = np.sort(new.cpu().numpy(), axis=-1)
        for r in range(B):
            keep = np.ones(M + 16, bool)
            j = 0
            for k in range(M + 16):             # remove one occurrence of every old value (both rows are sorted)
                if j < M and ref[r, k] == zp[r, j]:
                    keep[k] = False
                    j += 1
            assert j == M and keep.sum() == 16, (i, r, j)
            ref_new = ref[r, keep]
            err = np.abs(got[r] - ref_new)
            width = np.diff(ref[r]).max()
            assert err.max() <= width + 1e-6, (i, r, err.max(), width)     # never further off than one bin
            n_bad += int((err > 1e-5).sum())
            n_bad4 += int((err > 1e-4).sum())
            n_all += 16
        if i == 0:
            # identical inputs: what remains is the kernel's arithmetic against ATen's (sigmoid, fp64 scans, the flat-CDF branch)
            print("%s round 0 on the reference's sdf: %d of %d new z beyond 1e-5, %d beyond 1e-4" % (name, n_bad, n_all, n_bad4))
            assert n_bad <= 1 and n_bad4 == 0, (n_bad, n_bad4, n_all)
    # ill-conditioned entries (flat CDF) are counted, not skipped. The yardstick is the reference against
    # itself (fp32 vs fp64, SURVEY.md 4 [probe]): 13 % of the samples beyond 1e-5 and 0.8 % beyond 1e-4; a single round on
    # identical z (rounds 1-3 differ in the SDF values' last bits: the kernel's own fp32 SDF) must stay at that level.
    print("%s: %d of %d new z beyond 1e-5, %d beyond 1e-4" % (name, n_bad, n_all, n_bad4))
    assert n_bad <= 0.13 * n_all and n_bad4 <= 0.015 * n_all, (n_bad, n_bad4, n_all)


def test_merge_sorted_kernel_equals_stable_sort_with_ties(dev, golden, env):
    """a12: cat_z_vals' sort + permuted sdf (renderer.py:197-205): the merge kernel equals torch.sort(stable) on ragged /
    tied inputs, bit for bit."""
    from vdn_hip import lib
    from dpt_models.fields import _stream
    rng = np.random.RandomState(0)
    for (B, M, K) in ((1, 64, 16), (7, 112, 16), (5, 128, 32), (3, 1, 1)):
        z = np.sort(rng.rand(B, M).astype(np.float32), -1)
        nz = np.sort(rng.rand(B, K).astype(np.float32), -1)
        if M > 4:
            nz[0, 0] = z[0, 3]                 # a tie between an old and a new sample
            z[0, 5] = z[0, 4]                  # a tie inside the old samples
        sdf, nsdf = rng.randn(B, M).astype(np.float32), rng.randn(B, K).astype(np.float32)
        ld = M + K + 3
        zb, sb = torch.zeros(B, ld), torch.zeros(B, ld)
        zb[:, :M], sb[:, :M] = torch.tensor(z), torch.tensor(sdf)
        zb, sb, nzd, nsd = zb.to(dev), sb.to(dev), torch.tensor(nz).to(dev), torch.tensor(nsdf).to(dev)
        m = lib.VdnMergeArgs()
        m.z, m.sdf, m.new_z, m.new_sdf, m.z_out, m.sdf_out = (t.data_ptr() for t in (zb, sb, nzd, nsd, zb, sb))
        m.B, m.M, m.K, m.ld, m.ld_out = B, M, K, ld, ld
        lib.call("vdn_merge_sorted", m, _stream())
        zc, idx = torch.sort(torch.cat([torch.tensor(z), torch.tensor(nz)], -1), dim=-1, stable=True)
        sc = torch.gather(torch.cat([torch.tensor(sdf), torch.tensor(nsdf)], -1), 1, idx)
        assert np.array_equal(zb.detach().cpu().numpy()[:, :M + K], zc.numpy())
        assert np.array_equal(sb.detach().cpu().numpy()[:, :M + K], sc.numpy())


CASES = ["white_v03_c0", "white_v03_c05_det", "white_v065_c1", "wdepth_v03_c05", "wdepth_v065_c1",
         "white_n64_v03", "black_v03"]


def _render(rend, fx, dev, inject):
    kw = {}
    if inject and fx["n_importance"] > 0:
        kw["z_vals_inject"] = g(fx["z_vals_inside"], dev)
    return rend.render(g(fx["rays_o"], dev), g(fx["rays_d"], dev), g(fx["near"], dev), g(fx["far"], dev),
                       perturb_overwrite=(-1 if fx["perturb"] > 0 else 0),
                       background_rgb=torch.ones(1, 3, device=dev) if fx["white"] else None,
                       cos_anneal_ratio=float(fx["cos_anneal"]), t_rand=g(fx["t_rand"], dev),
                       t_rand_out=g(fx["t_rand_out"], dev), **kw)


@pytest.mark.parametrize("grad_mode", ["inference", "training"])
@pytest.mark.parametrize("name", CASES)
def test_render_vs_reference_golden(env, dev, golden, name, grad_mode):
    """End to end against the REFERENCE's outputs: per-ray tensors at 1e-4; with the reference's z injected
    also the per-sample tensors."""
    fx = golden(name)
    rend, _, _ = env(int(fx["seed"]), bool(fx["wdepth"]), float(fx["variance"]), n_importance=int(fx["n_importance"]))
    # 'inference' = torch.no_grad() path (per-call buffers); 'training' = the autograd node with kept activations
    torch.set_grad_enabled(grad_mode == "training")
    try:
        _check_golden(rend, fx, dev)
    finally:
        torch.set_grad_enabled(True)


def _check_golden(rend, fx, dev):
    out = _render(rend, fx, dev, inject=False)
    assert out["color_fine"].requires_grad == torch.is_grad_enabled()
    keys = ["color_fine", "weight_sum", "s_val", "z_vals", "gradient_error", "inside_sphere"] + (["render_feats"] if fx["wdepth"] else [])
    for k in keys:
        assert tuple(out[k].shape) == fx["out_" + k].shape, k
        assert relmax(out[k].detach().cpu().numpy(), fx["out_" + k]) < 1e-4, k
    assert set(out.keys()) == {"render_feats", "color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "gradients",
                               "weights", "z_vals", "gradient_error", "inside_sphere"}
    if not fx["wdepth"]:
        assert out["render_feats"] is None
    out = _render(rend, fx, dev, inject=True)
    # Per-sample alpha/weights see sdf * inv_s: at variance 0.65 (inv_s = 665) one fp32 ulp of an sdf ~0.5
    # (6e-8) already moves the sigmoid argument by 4e-5, and the oracle itself differs from the reference by
    # 5.6e-5 there (tests/golden/make_golden.py output). Per-sample tolerance is therefore 1e-4 at inv_s = 20
    # and 3e-4 at inv_s = 665; the per-ray outputs above stay at 1e-4 in both regimes.
    tol = 1e-4 if float(fx["variance"]) < 0.5 else 3e-4
    for k in ("weights", "cdf_fine", "gradients", "weight_max", "color_fine", "weight_sum"):
        assert tuple(out[k].shape) == fx["out_" + k].shape, k
        assert relmax(out[k].detach().cpu().numpy(), fx["out_" + k]) < (1e-4 if k in ("color_fine", "weight_sum", "gradients") else tol), k
    # the per-ray outputs element by element: |delta| <= 1e-4 |ref| + 1e-6 max|ref| (north-star "1e-4 rel"; relmax alone only
    # bounds the error against the tensor's largest entry)
    for k in ["color_fine", "weight_sum", "gradient_error"] + (["render_feats"] if fx["wdepth"] else []):
        assert relelem(out[k].detach().cpu().numpy(), fx["out_" + k]) <= 1.0, (k, relelem(out[k].detach().cpu().numpy(), fx["out_" + k]))


def test_sampler_rounds_vs_oracle(env, dev, golden):
    """Hierarchical z (4 rounds) against the oracle's z; the inverse CDF is ill-conditioned where the CDF is flat
    (SURVEY.md 4), so: sorted, same count, and all but a small fraction of samples within 1e-4."""
    import oracle.neus_oracle as orc
    fx = golden("white_v03_c0")
    rend, nets, _ = env(int(fx["seed"]), False, 0.3)
    o, d, near, far = (g(fx[k], dev) for k in ("rays_o", "rays_d", "near", "far"))
    z, z_out = rend._sample(o, d, near.reshape(-1), far.reshape(-1), 1.0, g(fx["t_rand"], dev), g(fx["t_rand_out"], dev), None)
    zz = z.detach().cpu().numpy()
    assert zz.shape == (int(fx["B"]), 128) and np.all(np.diff(zz, axis=1) >= 0)
    ref = fx["z_vals_inside"]
    frac_bad = np.mean(np.abs(zz - ref) > 1e-4)
    assert frac_bad < 0.03, frac_bad
    tt = torch.tensor
    zc, zo = orc.coarse_and_outside_z(tt(fx["near"]), tt(fx["far"]), orc.RendererConf(), 1.0, tt(fx["t_rand"]), tt(fx["t_rand_out"]))
    assert relmax(z_out.detach().cpu().numpy(), zo.numpy()) < 1e-6


@pytest.mark.parametrize("B", [1, 3, 130])
def test_render_ragged_batches_vs_oracle(env, dev, B):
    import oracle.neus_oracle as orc
    from vdn_train import synth
    rend, nets, _ = env(21, True, 0.3)
    cams = synth.make_cameras(21)
    px = np.floor(synth.uniform(21, "rag/x%d" % B, (B,)) * 500) + 150
    py = np.floor(synth.uniform(21, "rag/y%d" % B, (B,)) * 500) + 150
    o, d = synth.pixel_rays(cams[2], px, py)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(21, 0, B)
    tt = torch.tensor
    ref = orc.render(nets, tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3), cos_anneal_ratio=0.7,
                     t_rand=tt(t1), t_rand_out=tt(t2))
    out = rend.render(g(o, dev), g(d, dev), g(near, dev), g(far, dev), background_rgb=torch.ones(1, 3, device=dev),
                      cos_anneal_ratio=0.7, t_rand=g(t1, dev), t_rand_out=g(t2, dev))
    for k in ("color_fine", "weight_sum", "render_feats", "gradient_error"):
        assert relmax(out[k].detach().cpu().numpy(), ref[k].detach().numpy()) < 1e-4, k


def test_lattice_vs_reference(env, dev, golden):
    fx = golden("stages")
    rend, _, _ = env(3, True, 0.3)
    u = rend.extract_fields(torch.tensor([-0.8, -0.7, -0.6]), torch.tensor([0.7, 0.8, 0.9]), 20)
    assert relmax(u, fx["lattice"]) < 1e-5


def test_determinism(env, dev, golden):
    fx = golden("white_v03_c0")
    rend, _, _ = env(int(fx["seed"]), False, 0.3)
    a = _render(rend, fx, dev, inject=False)
    b = _render(rend, fx, dev, inject=False)
    for k in ("color_fine", "weights", "gradients"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("B,row0,crop", [(512, 32768, None), (512, 16384, None), (300, 4096, None), (64, 0, 420), (512, 32768, 420)])
def test_sdf_tail_kernel_equals_the_large_kernel_bit_for_bit(monkeypatch, B, row0, crop):
    """bf16 training forward: the rows behind `row0` of the foreground work list evaluated by the 32-row feature-split kernel
    (vdn_sdf_fwd_tail_bf16, csrc/k_sdf_fwd1_split.h) instead of by the 128-row kernel - sdf, normals, the feature plane and every
    saved plane (H, V, PE) of every listed row, bit for bit; full batches, a ragged one, a list that ends in front of row0 (no tail),
    a crop whose list is the whole batch (the tail would be longer than its limit: the large kernel keeps every row)."""
    from vdn_train import synth, factory
    from vdn_hip.train import TrainEngine
    from vdn_hip import layout
    dev = torch.device("cuda:0")
    seed = 21
    st = synth.make_all_states(seed)
    cams = synth.make_cameras(seed)
    o, d = synth.random_pixel_batch(seed, 0, 5, B, cams=cams, crop=crop)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, 0, B)
    o, d, near, far, t1, t2 = (torch.tensor(x).to(dev) for x in (o, d, near, far, t1, t2))
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VDN_SDF_TAIL", mode)
        monkeypatch.setenv("VDN_SDF_TAIL_ROW0", str(row0))
        monkeypatch.setenv("VDN_SDF_TAIL_MAX", "40000" if row0 != 32768 else "8192")
        rend = factory.build_renderer(device=dev, states=st, precision="bf16")
        eng = TrainEngine(rend, B, dev)
        with torch.no_grad():
            z, z_out = rend._sample(o, d, near.reshape(B), far.reshape(B), 1.0, t1, t2, None)
        for k in ("H", "V", "PE", "feat"):
            eng.w[k].zero_()
        eng.forward(o, d, z.contiguous(), z_out, torch.ones(3, device=dev), 0.3, skip_far=True)
        torch.cuda.synchronize()
        n = int(eng.w["fg_active"][1].item())
        idx = eng.w["fg_active"][0][:n].long()
        planes = {"sdf": eng.w["sdf"][idx].clone(), "normals": eng.w["normals"][idx].clone(),
                  "feat": layout.from_pt32(eng.w["feat"], eng.Pp, 256)[:n], "PE": layout.from_pt32(eng.w["PE"], eng.Pp, 64)[:n]}
        for l in range(8):
            planes["H%d" % l] = layout.from_pt32(eng.w["H"][l], eng.Pp, 256)[:n]
            planes["V%d" % l] = layout.from_pt32(eng.w["V"][l], eng.Pp, 256)[:n]
        res[mode] = (n, planes, {k: eng.w[k].clone() for k in ("color", "weights")})
    (n0, p0, o0), (n1, p1, o1) = res["0"], res["1"]
    assert n0 == n1 and n0 > 0
    for k in p0:
        assert torch.equal(p0[k], p1[k]), (k, n0, (p0[k].float() - p1[k].float()).abs().max().item())
        assert torch.isfinite(p0[k]).all()
    for k in o0:
        assert torch.equal(o0[k], o1[k]), k


def test_on_device_ray_generator(dev):
    """vdn_train.rays.RaysGenerator against the formulas of poses.py:168-212 / dataset.py:111-118 (numpy, float64)."""
    from vdn_train import synth
    from vdn_train.rays import RaysGenerator
    rng = np.random.RandomState(3)
    n, H, W, C = 3, 40, 56, 5
    images, masks, feats = rng.rand(n, H, W, 3).astype(np.float32), (rng.rand(n, H, W, 3) > 0.3).astype(np.float32), rng.rand(n, H, W, C).astype(np.float32)
    cams = synth.make_cameras(3, n=n)
    K = np.eye(4)
    K[:3, :3] = np.linalg.inv(synth.intrinsics_inv(focal=60.0, h=H, w=W))
    gen = RaysGenerator(images, masks, cams, K, depth_feats=feats, device=dev)
    px, py = rng.randint(0, W, 64).astype(np.float32), rng.randint(0, H, 64).astype(np.float32)
    out, near, far = gen.gen_random_rays_at(1, 64, pixels=(px, py), return_near_far=True)
    out = out.cpu().numpy()
    assert out.shape == (64, 10 + C)
    p = np.stack([px, py, np.ones_like(px)], -1).astype(np.float64) @ np.linalg.inv(K[:3, :3]).T
    v = p / np.linalg.norm(p, axis=-1, keepdims=True)
    d = v @ cams[1][:3, :3].T
    assert np.abs(out[:, 3:6] - d).max() < 1e-5 and np.abs(out[:, 0:3] - cams[1][:3, 3]).max() < 1e-6
    yi, xi = py.astype(int), px.astype(int)
    assert np.array_equal(out[:, 7:10], images[1][yi, xi]) and np.array_equal(out[:, 6], masks[1][yi, xi, 0])
    assert np.array_equal(out[:, 10:], feats[1][yi, xi])
    o64 = np.broadcast_to(cams[1][:3, 3], d.shape)
    mid = -(o64 * d).sum(-1) / (d * d).sum(-1)
    assert np.abs(near.cpu().numpy()[:, 0] - (mid - 1)).max() < 1e-5 and np.abs(far.cpu().numpy()[:, 0] - (mid + 1)).max() < 1e-5
    rnd = gen.gen_random_rays_at(0, 128)
    assert rnd.shape == (128, 10 + C) and float(rnd[:, 3:6].norm(dim=-1).sub(1).abs().max()) < 1e-5
    o, v = gen.gen_rays_at(2, resolution_level=2)
    assert o.shape == (H // 2, W // 2, 3) and v.shape == (H // 2, W // 2, 3)
    tx, ty = np.linspace(0, W - 1, W // 2), np.linspace(0, H - 1, H // 2)
    pp = np.array([tx[5], ty[7], 1.0]) @ np.linalg.inv(K[:3, :3]).T
    dd = (pp / np.linalg.norm(pp)) @ cams[2][:3, :3].T
    assert np.abs(v[7, 5].cpu().numpy() - dd).max() < 1e-5
    # a render fed by the generator runs end to end
    rend_rays = gen.gen_random_rays_at(0, 8)
    assert torch.isfinite(rend_rays).all()
    # poses.py:214-252: interpolated camera; the end points are the two cameras themselves, in between the centre moves on the
    # world-to-camera-side interpolation the reference uses and the directions stay unit vectors of a proper rotation
    for ratio, idx in ((0.0, 0), (1.0, 2)):
        ob, vb = gen.gen_rays_between(ratio, 0, 2, resolution_level=2)
        oa, va = gen.gen_rays_at(idx, resolution_level=2)
        assert (ob - oa).abs().max().item() < 2e-5 and (vb - va).abs().max().item() < 2e-5
    ob, vb = gen.gen_rays_between(0.3, 0, 2, resolution_level=2)
    assert ob.shape == (H // 2, W // 2, 3) and float(vb.norm(dim=-1).sub(1).abs().max()) < 1e-5
    w0, w1 = np.linalg.inv(cams[0]), np.linalg.inv(cams[2])
    from scipy.spatial.transform import Rotation as Rot, Slerp
    w = np.eye(4)
    w[:3, :3] = Slerp([0, 1], Rot.from_matrix(np.stack([w0[:3, :3], w1[:3, :3]])))(0.3).as_matrix()
    w[:3, 3] = (0.7 * w0 + 0.3 * w1)[:3, 3]
    c2w = np.linalg.inv(w)
    assert np.abs(ob[0, 0].cpu().numpy() - c2w[:3, 3]).max() < 2e-5
    pp = np.array([tx[5], ty[7], 1.0]) @ np.linalg.inv(K[:3, :3]).T
    assert np.abs(vb[7, 5].cpu().numpy() - (pp / np.linalg.norm(pp)) @ c2w[:3, :3].T).max() < 2e-5


def test_on_device_ray_generator_vs_reference_vectors(dev):
    """vdn_train.rays.RaysGenerator against the REFERENCE's own RaysGenerator / near_far_from_sphere outputs (tests/golden/rays.npz,
    written by make_golden.py::rays_fixture from poses.py:96-252 and dataset.py:111-118 run on CPU): random-pixel batches with the
    reference's recorded torch.randint draws injected (both constructor branches, with and without VDN target features), whole-image
    rays at three resolution levels, interpolated cameras."""
    from vdn_train.rays import RaysGenerator
    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rays.npz")))
    for tag, with_depth in (("rgba", True), ("rgbmask", False)):
        gen = RaysGenerator(fx[tag + "__images"], fx[tag + "__masks"], fx["pose_all"], fx["intrinsics_all"],
                            depth_feats=fx["rgba__depth_feats"] if with_depth else None, device=dev)
        for idx in (1, 2):
            k = "%s__rand_%d" % (tag, idx)
            ref, px, py = fx[k + "__data"], fx[k + "__pixels_x"], fx[k + "__pixels_y"]
            assert int(fx[k + "__img_idx"]) == idx
            out, near, far = gen.gen_random_rays_at(idx, len(px), pixels=(px, py), return_near_far=True)
            out = out.cpu().numpy()
            assert out.shape == ref.shape                                         # [B, 10 + C]; C = 1 zero column without features
            assert np.array_equal(out[:, 0:3], ref[:, 0:3])                       # rays_o: the pose's translation
            assert np.abs(out[:, 3:6] - ref[:, 3:6]).max() < 2e-6                 # rays_d
            assert np.array_equal(out[:, 6:], ref[:, 6:])                         # mask | rgb | feats: gathers
            assert np.abs(near.cpu().numpy() - fx[k + "__near"]).max() < 1e-5 and np.abs(far.cpu().numpy() - fx[k + "__far"]).max() < 1e-5
    for idx, l in ((0, 1), (2, 2), (1, 4)):
        o, v = gen.gen_rays_at(idx, resolution_level=l)
        ro, rv = fx["at_%d_l%d__rays_o" % (idx, l)], fx["at_%d_l%d__rays_v" % (idx, l)]
        assert tuple(o.shape) == ro.shape and tuple(v.shape) == rv.shape
        assert np.array_equal(o.cpu().numpy(), ro) and np.abs(v.cpu().numpy() - rv).max() < 2e-6
    for k in sorted(k[:-7] for k in fx if k.startswith("between_") and k.endswith("__ratio")):
        _, i0, i1, _, lv = k.split("_")
        o, v = gen.gen_rays_between(float(fx[k + "__ratio"]), int(i0), int(i1), resolution_level=int(lv[1:]))
        assert tuple(o.shape) == fx[k + "__rays_o"].shape
        assert np.abs(o.cpu().numpy() - fx[k + "__rays_o"]).max() < 1e-5 and np.abs(v.cpu().numpy() - fx[k + "__rays_v"]).max() < 1e-5


def test_val_img_over_a_scene_directory(tmp_path):
    """Scene files -> SceneData -> on-device rays -> batched render of one camera (Runner.val_img, dpt_runner.py:417-474),
    incl. the depth_from_sdf writer; the image equals direct render() calls on the same rays and jitter."""
    from PIL import Image
    from vdn_train import synth, factory, dataset, validate
    dev = torch.device("cuda:0")
    H, W, n = 24, 32, 2
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "image", "mask"))
    cams = synth.make_cameras(2)[:n]
    K4 = np.eye(4)
    K4[:3, :3] = [[30.0, 0, (W - 1) / 2.0], [0, 30.0, (H - 1) / 2.0], [0, 0, 1]]
    names = ["%03d" % i for i in range(n)]
    dataset.write_cameras_npz(os.path.join(root, "cameras_sphere.npz"), names, [K4 @ np.linalg.inv(c) for c in cams], [np.eye(4)] * n)
    rng = np.random.default_rng(1)
    for nm in names:
        Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8), "RGB").save(os.path.join(root, "image", nm + ".png"))
        Image.fromarray(np.full((H, W, 3), 255, np.uint8), "RGB").save(os.path.join(root, "image", "mask", nm + ".png"))
    scene = dataset.SceneData(root)
    gen = scene.rays_generator(dev)
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(2))
    torch.manual_seed(7)
    l1, psnr, eik, img = validate.val_img(rend, scene, gen, 1, resolution_level=1, batch_size=512, cos_anneal_ratio=0.8,
                                          gen_depth_for_finetune=True)
    assert img.shape == (H, W, 3) and np.isfinite(img).all() and eik.shape == (2,)
    want_l1, want_psnr = validate.image_metrics(img, scene.images[1])
    assert abs(l1 - want_l1) < 1e-6 and abs(psnr - want_psnr) < 1e-5
    depth = np.load(scene.depth_from_sdf_path(1))
    assert depth.shape == (H, W, 1) and (depth > 0).all()
    # the same rays / jitter through render() directly
    torch.manual_seed(7)
    o, d = gen.gen_rays_at(1)
    o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
    with torch.no_grad():
        for s in (0, 512):
            near, far = gen.near_far_from_sphere(o[s:s + 512], d[s:s + 512])
            out = rend.render(o[s:s + 512], d[s:s + 512], near, far, cos_anneal_ratio=0.8, background_rgb=torch.ones(1, 3, device=dev))
            assert np.array_equal(out["color_fine"].cpu().numpy(), img.reshape(-1, 3)[s:s + 512])
            w = out["weights"][:, :128] * out["inside_sphere"]
            zd = out["z_vals"].gather(1, w.argmax(-1, keepdim=True)).cpu().numpy()
            assert np.array_equal(zd, depth.reshape(-1, 1)[s:s + 512])
    # Runner.validate_image (dpt_runner.py:520-587): colour + normal image, written as the runner names them
    torch.manual_seed(7)
    img255, nimg = validate.validate_image(rend, gen, 1, resolution_level=1, batch_size=512, cos_anneal_ratio=0.8, out_dir=root, iter_step=12)
    assert np.allclose(img255, (img * 255).clip(0, 255)) and nimg.shape == (H, W, 3) and nimg.min() >= 0 and nimg.max() <= 255
    torch.manual_seed(7)
    with torch.no_grad():
        near, far = gen.near_far_from_sphere(o[:512], d[:512])
        out = rend.render(o[:512], d[:512], near, far, cos_anneal_ratio=0.8, background_rgb=torch.ones(1, 3, device=dev))
        nv = (out["gradients"] * out["weights"][:, :128, None] * out["inside_sphere"][..., None]).sum(1).cpu().numpy()
    want = (nv @ np.linalg.inv(cams[1][:3, :3]).T * 128 + 128).clip(0, 255)
    assert np.abs(nimg.reshape(-1, 3)[:512] - want).max() < 1e-3
    assert os.path.exists(os.path.join(root, "normals", "00000012_0_1.png"))
    # the colour file as cv.imwrite leaves it (dpt_runner.py:575-581): render stacked over the ground truth, the BGR arrays
    # stored so that the file shows true colours - its lower half is the input PNG itself
    disk = np.asarray(Image.open(os.path.join(root, "validations_fine", "00000012_0_1.png")))
    assert disk.shape == (2 * H, W, 3)
    assert np.array_equal(disk[H:], np.asarray(Image.open(os.path.join(root, "image", "001.png")).convert("RGB")))
    assert np.array_equal(disk[:H], np.rint(img255).astype(np.uint8)[..., ::-1])
    # Runner.render_novel_image (589-616): at ratio 0 the interpolated view is camera idx_0 itself
    torch.manual_seed(7)
    nov = validate.render_novel_image(rend, gen, 1, 0, 0.0, resolution_level=1, cos_anneal_ratio=0.8)
    assert nov.dtype == np.uint8 and nov.shape == (H, W, 3)
    assert np.abs(nov.astype(np.float64) - (img * 256).clip(0, 255).astype(np.uint8)).max() <= 1


@pytest.mark.gpu
def test_image_loops_vs_the_reference_runner(tmp_path):
    """tests/golden/runner.npz: the reference RUNNER's own Runner.val_img(gen_depth_for_finetune=True) and Runner.validate_image
    (dpt_runner.py:417-491, 520-587), run by the reference on CPU (make_golden.py::runner_fixture: dpt_runner.py loaded by file
    path, cv2 / pyhocon / trimesh / tensorboard stubbed, jitter injected per batch). vdn_train.validate on the fp32 kernels, same
    images, cameras, weights and jitter: the L1 / PSNR it reports, the depth_from_sdf array it saves, the weight_max picture, the
    colour file (render stacked over the ground truth) and the normal image as cv.imwrite received them."""
    from PIL import Image
    from vdn_train import synth, factory, validate
    from vdn_train.rays import RaysGenerator
    fx = load_golden("runner")
    dev = torch.device("cuda:0")
    seed, idx, H, W, BS = int(fx["seed"]), int(fx["idx"]), int(fx["H"]), int(fx["W"]), int(fx["batch_size"])
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(seed, wdepth=False, variance=float(fx["variance"])))
    # the constructor's RGBA branch (poses.py:117-122) is pinned by rays.npz; here its arrays come from the fixture
    gen = RaysGenerator(fx["images"], fx["masks"], fx["pose_all"], fx["intrinsic"], device=dev)
    car = min(1.0, float(fx["iter_step"]) / float(fx["anneal_end"]))                 # dpt_runner.py:304-308
    nb = (H * W + BS - 1) // BS
    jit = [(torch.tensor(fx["jitter/%d/t_rand" % b]).to(dev), torch.tensor(fx["jitter/%d/t_rand_out" % b]).to(dev)) for b in range(nb)]
    out = str(tmp_path)
    l1, psnr, eik, img = validate.val_img(rend, None, gen, idx, resolution_level=1, batch_size=BS, cos_anneal_ratio=car, gen_depth_for_finetune=True,
                                          jitter=jit, out_dir=out, iter_step=int(fx["iter_step"]))
    assert abs(l1 - float(fx["val_img/color_fine_loss"])) < 1e-4 * float(fx["val_img/color_fine_loss"])
    assert abs(psnr - float(fx["val_img/psnr"])) < 1e-3
    assert np.abs(eik.reshape(-1) - fx["val_img/gradient_error"].reshape(-1)).max() < 1e-4 * np.abs(fx["val_img/gradient_error"]).max()
    res = validate.render_image(rend, gen, idx, 1, BS, car, True, True, jitter=jit)
    want_d = fx["val_img/weight_depth"]
    assert res["weight_depth"].shape == want_d.shape == (H, W, 1)
    # the argmax of a ray's weights: fp32 noise can move it between near-equal neighbours on a few rays
    close = np.abs(res["weight_depth"] - want_d) < 1e-4 * np.abs(want_d).max()
    assert close.mean() >= 0.97, close.mean()
    wm = np.asarray(Image.open(os.path.join(out, str(fx["val_img/weight_max_name"]))))
    want_wm = np.rint(fx["val_img/weight_max_png"][..., 0]).clip(0, 255).astype(np.uint8)       # cv.imwrite: saturate_cast<uchar>
    assert wm.shape == want_wm.shape and (np.abs(wm.astype(int) - want_wm.astype(int)) <= 1).mean() >= 0.95
    img255, nimg = validate.validate_image(rend, gen, idx, resolution_level=1, batch_size=BS, cos_anneal_ratio=car, out_dir=out,
                                           iter_step=int(fx["iter_step"]), jitter=jit)
    want_val, want_n = fx["validate_image/validations_fine"], fx["validate_image/normals"]
    assert want_val.shape == (2 * H, W, 3) and want_n.shape == (H, W, 3)
    assert np.abs(img255 - want_val[:H]).max() < 0.03                                 # 1e-4 of the 0..255 range
    assert np.abs(gen.image_at(idx, 1) - want_val[H:]).max() < 1e-4                   # the ground-truth half: image_at (poses.py:254-256)
    assert np.abs(nimg - want_n).max() < 0.05
    for key, arr in (("validations_fine", want_val), ("normals", want_n)):
        disk = np.asarray(Image.open(os.path.join(out, str(fx["validate_image/%s_name" % key]))))
        ref8 = np.rint(arr).clip(0, 255).astype(np.uint8)[..., ::-1]               # cv.imwrite stores the BGR array; PIL reads RGB
        assert disk.shape == ref8.shape and (np.abs(disk.astype(int) - ref8.astype(int)) <= 1).all(), key


@pytest.mark.parametrize("wdepth", [False, True], ids=["womsk_white", "womsk_white_wdepth"])
def test_full_size_batch_vs_oracle(dev, wdepth):
    """BASELINE.json's full size directly against the oracle: 512 rays x (64 + 64 + 32) samples, fp32 kernels, both shipped
    configurations (renderer.py:332-439). End to end (own sampler) the per-ray outputs hold 1e-4 of the largest entry; with the
    oracle's z injected - the inverse-CDF sampler is ill-conditioned where the CDF is flat, which is sampling
    noise, not arithmetic - they hold it element by element: |delta| <= 1e-4 |ref| + 1e-6 max|ref|."""
    import oracle.neus_oracle as orc
    from vdn_train import synth, factory
    B, seed = 512, 5
    st = synth.make_all_states(seed, wdepth=wdepth)
    rend = factory.build_renderer(wdepth=wdepth, device=dev, states=st)
    nets = orc.nets_from_numpy(st)
    cams = synth.make_cameras(seed)
    o, d = synth.random_pixel_batch(seed, 0, 0, B, cams=cams, crop=420)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, 0, B)
    tt = torch.tensor
    rec = {}
    with torch.no_grad():
        ref = orc.render(nets, tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3), cos_anneal_ratio=0.5,
                         t_rand=tt(t1), t_rand_out=tt(t2), record=rec)
        args = [g(x, dev) for x in (o, d, near, far)]
        kw = dict(background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=0.5, t_rand=g(t1, dev), t_rand_out=g(t2, dev))
        out = rend.render(*args, **kw)
        inj = rend.render(*args, z_vals_inject=rec["z_vals_inside"].to(dev).contiguous(), **kw)
    keys = ["color_fine", "weight_sum", "gradient_error"] + (["render_feats"] if wdepth else [])
    for k in keys:
        assert tuple(out[k].shape) == tuple(ref[k].shape), k
        assert relmax(out[k].cpu().numpy(), ref[k].numpy()) < 1e-4, (k, "end to end")
        e = relelem(inj[k].cpu().numpy(), ref[k].numpy())
        assert e <= 1.0, (k, "injected z, element-wise", e)
    for k in ("weights", "gradients", "cdf_fine"):
        assert relmax(inj[k].cpu().numpy(), ref[k].numpy()) < 1e-4, k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_full_size_batch_properties(dev, precision):
    """BASELINE.json's full size (512 rays x 128 + 32 samples): size-independent properties (the direct comparison with the
    oracle is test_full_size_batch_vs_oracle).
    Rays are independent, so rendering the batch in two halves or in a permuted order gives the same per-ray outputs;
    weights are a sub-probability distribution per ray; z is sorted."""
    from vdn_train import synth, factory
    B = 512
    rend = factory.build_renderer(wdepth=True, device=dev, states=synth.make_all_states(0, wdepth=True), precision=precision)
    cams = synth.make_cameras(0)
    o, d = synth.random_pixel_batch(0, 0, 0, B, cams=cams, crop=420)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(0, 0, B)
    o, d, near, far, t1, t2 = (g(x, dev) for x in (o, d, near, far, t1, t2))
    bg = torch.ones(1, 3, device=dev)
    run = lambda idx: rend.render(o[idx], d[idx], near[idx], far[idx], background_rgb=bg, cos_anneal_ratio=0.5,
                                  t_rand=t1[idx], t_rand_out=t2[idx])
    keys = ("color_fine", "render_feats", "weights", "weight_sum", "weight_max", "gradients", "z_vals", "cdf_fine", "inside_sphere")
    with torch.no_grad():
        full = run(torch.arange(B, device=dev))
        lo, hi = run(torch.arange(0, 256, device=dev)), run(torch.arange(256, B, device=dev))
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(dev)
        pf = run(perm)
    for k in keys:
        assert torch.equal(full[k], torch.cat([lo[k], hi[k]])), k
        assert torch.equal(full[k][perm], pf[k]), k
    w = full["weights"]
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-4).all()      # (1 - alpha + 1e-7) factors: up to 160e-7 above 1
    assert torch.allclose(full["weight_sum"][:, 0], w.sum(-1), atol=1e-6)
    assert torch.equal(full["weight_max"][:, 0], w.max(-1).values)
    assert (full["z_vals"][:, 1:] >= full["z_vals"][:, :-1]).all()
    col = full["color_fine"]
    assert torch.isfinite(col).all()          # not range-bound: the NeRF++ background colour has no sigmoid (fields.py:349)
    assert torch.isfinite(full["gradients"]).all() and float(full["gradient_error"]) >= 0


def test_render_on_a_non_default_stream(env, dev, golden):
    """The C ABI takes the hipStream_t as a 64-bit handle: launching from a side stream gives the same bits as the default one."""
    fx = golden("white_v03_c0")
    rend, _, _ = env(int(fx["seed"]), False, 0.3)
    a = _render(rend, fx, dev, inject=False)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        b = _render(rend, fx, dev, inject=False)
    s.synchronize()
    for k in ("color_fine", "weights", "gradients"):
        assert torch.equal(a[k], b[k]), k


def _mesh_stats(V, F):
    from collections import Counter
    d = Counter()
    for a, b in ((0, 1), (1, 2), (2, 0)):
        for e in zip(F[:, a].tolist(), F[:, b].tolist()):
            d[e] += 1
    und = Counter()
    for (a, b), c in d.items():
        und[(min(a, b), max(a, b))] += c
    return max(d.values()), set(und.values()), len(V) - len(und) + len(F)


def test_marching_tets_kernel_equals_oracle(dev):
    """vdn_mesh_count / vdn_mesh_emit against the numpy restatement on a small smooth field: the same triangles, in the same
    order, bit for bit (positions and edge keys)."""
    from oracle import marching_tets as omt
    from vdn_hip import lib, mesh
    R = 13
    rng = np.random.default_rng(4)
    gx = np.linspace(-1, 1, R)
    X, Y, Z = np.meshgrid(gx, gx, gx, indexing="ij")
    u = (0.55 - np.sqrt(X * X + 0.8 * Y * Y + 1.3 * Z * Z) + 0.15 * np.sin(3 * X) * np.cos(2 * Y) + 0.02 * rng.standard_normal(X.shape)).astype(np.float32)
    want_pos, want_key = omt.marching_tets(u, 0.03)
    ud = torch.tensor(u, device=dev)
    n = (R - 1) ** 3
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    a = lib.VdnMeshArgs()
    a.u, a.threshold, a.R, a.counts = ud.data_ptr(), 0.03, R, counts.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    lib.call("vdn_mesh_count", a, st)
    incl = torch.cumsum(counts, 0, dtype=torch.int64)
    assert int(incl[-1]) == len(want_pos) > 100
    off = (incl - counts).contiguous()
    pos = torch.empty(len(want_pos), 3, 3, device=dev)
    key = torch.empty(len(want_pos), 3, dtype=torch.int64, device=dev)
    a.offsets, a.tri_pos, a.tri_key = off.data_ptr(), pos.data_ptr(), key.data_ptr()
    lib.call("vdn_mesh_emit", a, st)
    assert np.array_equal(key.cpu().numpy(), want_key)
    np.testing.assert_allclose(pos.cpu().numpy(), want_pos, rtol=0, atol=2e-6)     # fma contraction: within 1 ulp of the lattice scale
    V, F = mesh.marching_tets(ud, 0.03)
    Vo, Fo = omt.weld(want_pos, want_key)
    assert np.array_equal(F.cpu().numpy(), Fo)
    np.testing.assert_allclose(V.cpu().numpy(), Vo, rtol=0, atol=2e-6)


@pytest.mark.parametrize("shape,chi", [("sphere", 2), ("torus", 0)])
def test_marching_tets_surface_properties(dev, shape, chi):
    """Size-independent properties at a larger lattice: closed (every edge in exactly two triangles), consistently oriented
    (every directed edge once), right Euler characteristic, vertices on the level set, normals pointing outwards."""
    from vdn_hip import mesh
    R = 72
    g1 = torch.linspace(-1, 1, R, device=dev)
    X, Y, Z = torch.meshgrid(g1, g1, g1, indexing="ij")
    if shape == "sphere":
        sdf = torch.sqrt(X * X + Y * Y + Z * Z) - 0.7
    else:
        sdf = torch.sqrt((torch.sqrt(X * X + Y * Y) - 0.6) ** 2 + Z * Z) - 0.22
    V, F = mesh.marching_tets(-sdf, 0.0)
    V, F = V.cpu().numpy().astype(np.float64), F.cpu().numpy()
    dmax, uses, euler = _mesh_stats(V, F)
    assert dmax == 1 and uses == {2} and euler == chi
    h = 2.0 / (R - 1)
    P = V * h - 1.0
    if shape == "sphere":
        r = np.linalg.norm(P, axis=1)
        assert np.abs(r - 0.7).max() < 0.6 * h * h / 0.7 + 1e-6            # linear interpolation of a curved field: O(h^2)
        nrm = np.cross(P[F[:, 1]] - P[F[:, 0]], P[F[:, 2]] - P[F[:, 0]])
        assert (np.sum(nrm * P[F].mean(1), axis=1) > 0).all()
        area = 0.5 * np.linalg.norm(nrm, axis=1).sum()
        assert abs(area - 4 * np.pi * 0.49) < 0.01 * 4 * np.pi * 0.49


@pytest.mark.parametrize("field", ["blob", "noise", "cut_by_the_box", "flat_ties", "torus"])
def test_marching_cubes_kernel_equals_the_pymcubes_restatement(dev, field):
    """vdn_mesh_mc_count / vdn_mesh_mc_emit against oracle/marching_cubes.py (the published algorithm of the PyMCubes the reference
    calls at renderer.py:36): the SAME vertex array and the SAME triangle array - counts, order, float64 bits. Fields: a smooth blob
    (the stages.npz kind of level set), white noise (all 256 cases, every ambiguous face), a surface that leaves the lattice through
    all six faces (the cells that create vertices on shared boundary edges), values that tie with the level and with each other
    (`<=` side of the case test, the f1 == f2 midpoint), and a torus."""
    from oracle import marching_cubes as omc
    from vdn_hip import mesh
    rng = np.random.default_rng(11)
    R = 14
    gx = np.linspace(-1, 1, R)
    X, Y, Z = np.meshgrid(gx, gx, gx, indexing="ij")
    iso = 0.03
    if field == "blob":
        u = 0.55 - np.sqrt(X * X + 0.8 * Y * Y + 1.3 * Z * Z) + 0.15 * np.sin(3 * X) * np.cos(2 * Y) + 0.02 * rng.standard_normal(X.shape)
    elif field == "noise":
        u, iso = rng.standard_normal(X.shape), 0.0
    elif field == "cut_by_the_box":
        u = 1.25 - np.sqrt(X * X + Y * Y + Z * Z) + 0.1 * np.sin(5 * X + 1) * np.sin(4 * Y) * np.cos(3 * Z)
    elif field == "flat_ties":
        u, iso = np.round(2.0 * (0.7 - np.sqrt(X * X + Y * Y + Z * Z))) / 2.0, 0.0          # multiples of 0.5: many values equal the level
    else:
        u, iso = 0.25 - np.sqrt((np.sqrt(X * X + Y * Y) - 0.6) ** 2 + Z * Z), 0.0
    u = u.astype(np.float32)
    Vo, Fo = omc.marching_cubes(u, iso)
    V, F = mesh.marching_cubes(torch.tensor(u, device=dev), iso)
    assert V.dtype == torch.float64 and F.dtype == torch.int64
    V, F = V.cpu().numpy(), F.cpu().numpy()
    assert V.shape == Vo.shape and F.shape == Fo.shape and len(Fo) > 50, (V.shape, Vo.shape, F.shape, Fo.shape)
    assert np.array_equal(F, Fo)
    assert np.array_equal(V, Vo)            # float64, bit for bit: one division per vertex, evaluated as the library writes it


def test_marching_cubes_surface_properties_at_full_size(dev):
    """The reference's default lattice (resolution 64 blocks; here 128^3): closed, consistently oriented, Euler characteristic of the
    shape, vertices on the level set, one vertex per cut lattice edge."""
    from vdn_hip import mesh
    R = 128
    g1 = torch.linspace(-1, 1, R, device=dev)
    X, Y, Z = torch.meshgrid(g1, g1, g1, indexing="ij")
    sdf = torch.sqrt((torch.sqrt(X * X + Y * Y) - 0.6) ** 2 + Z * Z) - 0.22
    V, F = mesh.marching_cubes(-sdf, 0.0)
    V, F = V.cpu().numpy(), F.cpu().numpy()
    dmax, uses, euler = _mesh_stats(V, F)
    assert dmax == 1 and uses == {2} and euler == 0
    assert len(np.unique(V, axis=0)) == len(V)
    P = V * (2.0 / (R - 1)) - 1.0
    d = np.sqrt((np.sqrt(P[:, 0] ** 2 + P[:, 1] ** 2) - 0.6) ** 2 + P[:, 2] ** 2) - 0.22
    assert np.abs(d).max() < 3e-4


def test_extract_geometry_matches_the_restatement_on_the_networks_lattice(env, dev):
    """NeuSRenderer.extract_geometry (renderer.py:441-446 -> 33-41) end to end on the SDF network: the lattice u = -sdf the kernels
    fill, triangulated on the device, equals the restatement applied to that same lattice - vertex and face counts, arrays bit for
    bit after the reference's own world-coordinate transform."""
    from oracle import marching_cubes as omc
    from dpt_models import renderer as rmod
    rend, _, _ = env(3, True, 0.3)
    lo, hi, res = torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0]), 24
    V, F = rend.extract_geometry(lo, hi, resolution=res, threshold=0.0)
    u = rmod.extract_fields(lo, hi, res, lambda pts: -rend.sdf_network.sdf(pts))
    Vo, Fo = omc.marching_cubes(u, 0.0)
    Vo = Vo / (res - 1.0) * 2.0 + (-1.0)                                    # renderer.py:39
    assert V.shape == Vo.shape and F.shape == Fo.shape and len(F) > 100
    assert np.array_equal(F, Fo) and np.array_equal(V, Vo)
    Vt, Ft = rend.extract_geometry(lo, hi, resolution=res, threshold=0.0, method="tets")      # the tetrahedra stay available
    assert len(Ft) > len(F)


def test_extract_geometry_without_pymcubes(env, dev):
    rend, _, _ = env(3, True, 0.3)
    V, F = rend.extract_geometry(torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0]), resolution=40, threshold=0.0)
    assert V.shape[1] == 3 and F.shape[1] == 3 and len(F) > 100 and F.max() == len(V) - 1
    sdf = rend.sdf_network.sdf(torch.tensor(V, dtype=torch.float32, device=dev)).cpu().numpy()
    assert np.abs(sdf).max() < 2e-2                                        # vertices sit on the network's zero level set
    dmax, uses, _ = _mesh_stats(V, F)
    assert dmax == 1 and uses <= {1, 2}                                    # closed except where the surface leaves the box


def test_smoke_entry_and_library_loaded_before_torch():
    """The driver's smoke() hook, and a fresh process that loads the kernel library before it ever imports torch itself
    (the library must still bind to torch's HIP runtime)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as g
    g.smoke()
    code = ("import sys; sys.path.insert(0, %r); from vdn_hip import lib; lib.load(); import torch; "
            "from vdn_train import synth, factory; r = factory.build_renderer(device=torch.device('cuda:0'), states=synth.make_all_states(0)); "
            "print(tuple(r.sdf_network.sdf(torch.zeros(4, 3, device='cuda:0')).shape))" % os.path.join(root, "vdn-nerf_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "(4, 1)" in out.stdout, out.stderr[-2000:]


def test_background_active_list_and_bitwise_equivalence(env, dev, golden, monkeypatch):
    """The NeRF++ background network skips the samples render_core multiplies by (1 - inside_sphere) = 0
    (renderer.py:284-299). (1) The device list is exactly {inside samples outside the unit sphere} + {outside samples}, in
    ascending order. (2) Rendering with and without the skip gives bit-identical outputs and parameter gradients."""
    from dpt_models.renderer import background_active
    fx = golden("wdepth_v03_c05")
    rend, _, _ = env(int(fx["seed"]), True, 0.3)
    o, d, near, far = (g(fx[k], dev) for k in ("rays_o", "rays_d", "near", "far"))
    kw = dict(background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=0.5, t_rand=g(fx["t_rand"], dev), t_rand_out=g(fx["t_rand_out"], dev))

    def run():
        for p in rend._all_parameters():
            p.grad = None
        out = rend.render(o, d, near, far, **kw)
        (out["color_fine"].sum() + out["render_feats"].sum() + 0.1 * out["gradient_error"]).backward()
        return out, [p.grad.clone() for p in rend._all_parameters()]

    out1, g1 = run()
    monkeypatch.setenv("VDN_BG_COMPACT", "0")
    out0, g0 = run()
    for k in ("color_fine", "render_feats", "weights", "weight_sum", "gradient_error"):
        assert torch.equal(out1[k], out0[k]), k
    # dW sums run over fewer rows in a different split partition: equal up to fp32 summation order
    for a, b in zip(g1, g0):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-12
    # the list itself
    B, N, T = o.shape[0], 128, 160
    inside = out1["inside_sphere"].cpu().numpy()
    with torch.no_grad():
        z, _ = rend._sample(o, d, near.reshape(-1), far.reshape(-1), rend.perturb, kw["t_rand"], kw["t_rand_out"], None)
        _, mid_z = rend._sections(z.contiguous(), N, 2.0 / rend.n_samples)
        idx, n = background_active(o, d, mid_z, T)
    want = [r * T + s for r in range(B) for s in range(T) if s >= N or inside[r, s] == 0.0]
    assert int(n) == len(want) and idx[:int(n)].cpu().tolist() == want
    assert 32 * B <= len(want) < B * T          # something was skipped, the outside samples never are


@pytest.mark.parametrize("sorted_old", [True, False])
def test_merge_sorted_kernel_vs_stable_sort(dev, sorted_old):
    """vdn_merge_sorted = cat + sort + permuted sdf (renderer.py:197-205), with ties and - through the counting path -
    unsorted old rows: equals a stable sort of the concatenation, for z and for the sdf that travels with it."""
    from vdn_hip import lib
    gen = torch.Generator().manual_seed(11)
    B, M, K, ld = 37, 80, 16, 128
    old = torch.round(torch.rand(B, M, generator=gen) * 50) / 50            # many exact ties
    if sorted_old:
        old = torch.sort(old, -1)[0]
    new = torch.sort(torch.round(torch.rand(B, K, generator=gen) * 50) / 50, -1)[0]
    sdf_old, sdf_new = torch.rand(B, M, generator=gen), torch.rand(B, K, generator=gen)
    z = torch.zeros(B, ld)
    s = torch.zeros(B, ld)
    z[:, :M], s[:, :M] = old, sdf_old
    z, s, new_d, sdf_new_d = z.to(dev), s.to(dev), new.to(dev).contiguous(), sdf_new.to(dev).contiguous()
    m = lib.VdnMergeArgs()
    m.z, m.new_z, m.z_out = z.data_ptr(), new_d.data_ptr(), z.data_ptr()
    m.sdf, m.new_sdf, m.sdf_out = s.data_ptr(), sdf_new_d.data_ptr(), s.data_ptr()
    m.B, m.M, m.K, m.ld, m.ld_out = B, M, K, ld, ld
    lib.call("vdn_merge_sorted", m, torch.cuda.current_stream().cuda_stream)
    cat_z, cat_s = torch.cat([old, new], -1), torch.cat([sdf_old, sdf_new], -1)
    want_z, idx = torch.sort(cat_z, dim=-1, stable=True)
    assert torch.equal(z[:, :M + K].cpu(), want_z)
    assert torch.equal(s[:, :M + K].cpu(), torch.gather(cat_s, 1, idx))


def test_train_prep_equals_the_separate_launches(dev):
    """vdn_train_prep (sections of both depth sets + both work lists in two launches) against vdn_sections x 2,
    vdn_foreground_active and vdn_background_active: bit for bit, with and without the foreground list."""
    from vdn_hip import lib
    from vdn_train import synth
    B, N, O = 37, 128, 32
    T = N + O
    o, d = synth.random_pixel_batch(4, 0, 3, B, crop=520)
    near, far = synth.near_far_from_sphere(o, d)
    rng = np.random.RandomState(4)
    z = np.sort(near + (far - near) * rng.rand(B, N).astype(np.float32), axis=1).astype(np.float32)
    z_out = np.sort(far + 0.02 + 3.0 * rng.rand(B, O).astype(np.float32), axis=1).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    zt, zf = g(z, dev), g(np.concatenate([z, z_out], 1), dev)
    ro, rd = g(o, dev), g(d, dev)
    f = lambda *s: torch.full(s, -7.0, dtype=torch.float32, device=dev)
    i32 = lambda *s: torch.full(s, -7, dtype=torch.int32, device=dev)
    ref = dict(dists=f(B, N), mid=f(B, N), bdists=f(B, T), bmid=f(B, T), fg=(i32(B * N), i32(1), i32(B)), bg=(i32(B * T), i32(1), i32(B)))
    for zz, dd, mm, n in ((zt, ref["dists"], ref["mid"], N), (zf, ref["bdists"], ref["bmid"], T)):
        a = lib.VdnSectionArgs()
        a.z, a.dists, a.mid_z, a.sample_dist, a.B, a.n, a.ld = zz.data_ptr(), dd.data_ptr(), mm.data_ptr(), 2.0 / 64, B, n, n
        lib.call("vdn_sections", a, st)
    fa = lib.VdnForegroundActiveArgs()
    fa.rays_o, fa.rays_d, fa.mid_z, fa.B, fa.N, fa.radius = ro.data_ptr(), rd.data_ptr(), ref["mid"].data_ptr(), B, N, 1.2
    fa.active_idx, fa.n_active, fa.ray_counts = (t.data_ptr() for t in ref["fg"])
    lib.call("vdn_foreground_active", fa, st)
    ba = lib.VdnBackgroundActiveArgs()
    ba.rays_o, ba.rays_d, ba.mid_z, ba.B, ba.N, ba.T = ro.data_ptr(), rd.data_ptr(), ref["mid"].data_ptr(), B, N, T
    ba.active_idx, ba.n_active, ba.ray_counts = (t.data_ptr() for t in ref["bg"])
    lib.call("vdn_background_active", ba, st)
    for with_fg in (True, False):
        got = dict(dists=f(B, N), mid=f(B, N), bdists=f(B, T), bmid=f(B, T), fg=(i32(B * N), i32(1), i32(B)), bg=(i32(B * T), i32(1), i32(B)))
        tp = lib.VdnTrainPrepArgs()
        zo_t, zfeed = g(z_out, dev), f(B, T)
        tp.rays_o, tp.rays_d, tp.z, tp.z_out, tp.z_feed = ro.data_ptr(), rd.data_ptr(), zt.data_ptr(), zo_t.data_ptr(), zfeed.data_ptr()
        tp.B, tp.N, tp.T, tp.z_ld, tp.sample_dist, tp.fg_radius = B, N, T, N, 2.0 / 64, 1.2
        tp.dists, tp.mid_z, tp.bg_dists, tp.bg_mid = (got[k].data_ptr() for k in ("dists", "mid", "bdists", "bmid"))
        if with_fg:
            tp.fg_active_idx, tp.fg_n_active, tp.fg_ray_counts = (t.data_ptr() for t in got["fg"])
        tp.bg_active_idx, tp.bg_n_active, tp.bg_ray_counts = (t.data_ptr() for t in got["bg"])
        lib.call("vdn_train_prep", tp, st)
        assert torch.equal(zfeed, zf)
        for k in ("dists", "mid", "bdists", "bmid"):
            assert torch.equal(got[k], ref[k]), k
        for k in ("fg", "bg") if with_fg else ("bg",):
            n = int(ref[k][1].item())
            assert int(got[k][1].item()) == n and 0 < n < ref[k][0].numel()
            assert torch.equal(got[k][0][:n], ref[k][0][:n]) and torch.equal(got[k][2], ref[k][2]), k
        if not with_fg:
            assert int(got["fg"][1].item()) == -7


@pytest.mark.gpu
@pytest.mark.parametrize("wdepth", [False, True], ids=["womsk_white", "womsk_white_wdepth"])
def test_render_plan_replays_equal_render_calls(wdepth):
    """NeuSRenderer.plan(): render() captured once as a HIP graph and replayed on new
    rays is bit-identical to plain render() calls; a parameter changed in place is picked up by the next replay; with the
    jitter on, every replay draws its own."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(wdepth=wdepth, device=dev, states=synth.make_all_states(3, wdepth=wdepth), precision="bf16")
    cams = synth.make_cameras(3)
    bg = torch.ones(1, 3, device=dev)
    kw = dict(background_rgb=bg, cos_anneal_ratio=0.7, perturb_overwrite=0, depth_before_color=False)

    def batch(step):
        o, d = synth.random_pixel_batch(3, step, step % len(cams), 512, rank=0, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        return tuple(torch.tensor(x).to(dev) for x in (o, d, near, far))
    with torch.no_grad():
        plan = rend.plan(512, **kw)
        for step in range(3):
            if step == 2:
                for p in rend.sdf_network.parameters():
                    p.mul_(1.01)
                rend.color_network.parameters().__next__().add_(0.003)
            b = batch(step)
            got = {k: v.clone() for k, v in plan(*b).items() if v is not None}
            want = rend.render(*b, **kw)
            for k, v in got.items():
                assert torch.equal(v, want[k]), (step, k)
        with pytest.raises(ValueError):
            plan(*[t[:100] for t in batch(0)])
        jit = rend.plan(512, background_rgb=bg, cos_anneal_ratio=0.7, perturb_overwrite=1)
        b = batch(0)
        z0 = jit(*b)["z_vals"][:, :128].clone()
        z1 = jit(*b)["z_vals"][:, :128].clone()
        assert not torch.equal(z0, z1) and (z0 - z1).abs().max() < 0.5
    with pytest.raises(RuntimeError):
        for p in rend.sdf_network.parameters():
            p.requires_grad_(True)
        rend.plan(512, **kw)


@pytest.mark.gpu
def test_small_sdf_passes_equal_the_large_kernel_bit_for_bit():
    """vdn_sdf_mlp_fwd_bf16(mode 0) routes launches of <= 8192 points (the sampler's up-sampling passes, renderer.py:352-372)
    to the feature-split kernel (csrc/k_sdf_fwd0_split.h) and larger ones to the 128-point kernel: same values, bit for bit,
    for point input and for the sampler's ray form (column slices of wider buffers), full and ragged sizes."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(5), precision="bf16")
    net = rend.sdf_network
    g = torch.Generator(device=dev).manual_seed(11)
    with torch.no_grad():
        for P in (8192, 8191, 37, 1):
            pts = (torch.rand(P, 3, device=dev, generator=g) * 2 - 1) * 1.2
            small = net._run(0, pts=pts)
            reps = (16384 + P - 1) // P + 1                    # the same points inside a launch the large kernel takes
            large = net._run(0, pts=pts.repeat(reps, 1))
            assert large.shape[0] > 8192 and torch.equal(small, large[:P]), P
            assert torch.isfinite(small).all()
        B = 512
        o = torch.rand(B, 3, device=dev, generator=g) - 0.5
        d = torch.nn.functional.normalize(torch.rand(B, 3, device=dev, generator=g) - 0.5, dim=1)
        z = torch.rand(B, 128, device=dev, generator=g) * 2
        s16, s64 = torch.zeros(B, 128, device=dev), torch.zeros(B, 128, device=dev)
        net._run(0, rays=(o, d, z[:, 64:80]), sdf_out=s16[:, 64:80])       # 8192 points: split kernel
        net._run(0, rays=(o, d, z[:, 32:96]), sdf_out=s64[:, 32:96])       # 32768 points: large kernel
        assert torch.equal(s16[:, 64:80], s64[:, 64:80]) and (s16[:, :64] == 0).all() and (s16[:, 80:] == 0).all()


@pytest.mark.gpu
def test_fused_sampler_rounds_equal_the_two_launch_rounds(monkeypatch):
    """vdn_sdf_merge_upsample_bf16 (the SDF pass of an up-sampling round, renderer.py:201, and cat_z_vals + the next up_sample,
    renderer.py:372-386, in one launch) against the two launches it replaces: every output of render() bit for bit, jitter on."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(4), precision="bf16")
    cams = synth.make_cameras(4)
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("VDN_FUSE_SDF_ROUNDS", fused)
        res = []
        for B in (512, 77):                     # 77 rays: the last workgroup holds one ray
            o, d = synth.random_pixel_batch(4, 1, 1, B, rank=0, cams=cams)
            near, far = synth.near_far_from_sphere(o, d)
            torch.manual_seed(3)
            with torch.no_grad():
                res.append(rend.render(*(torch.tensor(x).to(dev) for x in (o, d, near, far)), background_rgb=torch.ones(1, 3, device=dev),
                                       cos_anneal_ratio=0.6))
        outs[fused] = res
    for a, b in zip(outs["1"], outs["0"]):
        for k, v in a.items():
            if v is not None:
                assert torch.equal(v, b[k]), k
    assert torch.isfinite(outs["1"][0]["z_vals"]).all()


@pytest.mark.parametrize("B,N,radius", [(1, 128, 1.2), (37, 128, 1.2), (512, 128, 1.2), (65, 64, 0.9), (300, 130, 5.0), (300, 130, 1e-3)])
def test_foreground_list_and_its_complement_partition_the_samples(dev, B, N, radius):
    """vdn_foreground_active and its `complement` form (include/vdn_render.h; render() under grad evaluates the SDF network with
    saves on the first list and without on the second): dense ids ascending in both, disjoint, together every sample of every
    ray, and the listed ones are exactly those whose mid-point lies within `radius` by the compositor's own float expression -
    ragged sizes, the empty list (radius 5: everything inside; 1e-3: nothing) and NaN depths included (a NaN norm fails `<`: it
    belongs to the complement)."""
    from vdn_hip import lib
    rng = np.random.default_rng(B * 1000 + N)
    o = rng.standard_normal((B, 3)).astype(np.float32) * 1.5
    d = rng.standard_normal((B, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    mid = np.sort(rng.uniform(0.0, 4.0, (B, N)).astype(np.float32), axis=1)
    if B > 1:
        mid[B // 2, N // 3] = np.nan
    to, td, tm = (torch.tensor(x).to(dev) for x in (o, d, mid))
    lists = []
    for comp in (0, 1):
        idx = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
        n = torch.zeros(1, dtype=torch.int32, device=dev)
        cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        a = lib.VdnForegroundActiveArgs()
        a.rays_o, a.rays_d, a.mid_z, a.B, a.N, a.radius, a.complement = to.data_ptr(), td.data_ptr(), tm.data_ptr(), B, N, radius, comp
        a.active_idx, a.n_active, a.ray_counts = idx.data_ptr(), n.data_ptr(), cnt.data_ptr()
        lib.call("vdn_foreground_active", a, torch.cuda.current_stream().cuda_stream)
        k = int(n.item())
        got = idx.cpu().numpy()
        assert (got[k:] == -1).all()                           # nothing written behind the count
        lists.append(got[:k])
    fg, rest = lists
    assert len(fg) + len(rest) == B * N
    assert (np.diff(fg) > 0).all() and (np.diff(rest) > 0).all()
    assert np.array_equal(np.sort(np.concatenate([fg, rest])), np.arange(B * N))
    # the compositor's expression, operation for operation in float32 (no contraction)
    x = o[:, None, 0] + d[:, None, 0] * mid
    y = o[:, None, 1] + d[:, None, 1] * mid
    w = o[:, None, 2] + d[:, None, 2] * mid
    pn = np.sqrt(((x * x).astype(np.float32) + (y * y).astype(np.float32)).astype(np.float32) + (w * w).astype(np.float32), dtype=np.float32)
    want = np.flatnonzero((pn < np.float32(radius)).reshape(-1))
    assert np.array_equal(fg, want.astype(np.int32))
