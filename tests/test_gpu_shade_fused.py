"""The north-star kernel of the inference path: vdn_shade_fused_bf16 (csrc/k_sdf_fwd2.h MODE 2) = renderer.py:239-315 in ONE
launch - SDF network + gradient sweep, colour head on the feature vector in registers, the ray's NeuS alpha / background blend /
transmittance scan / weighted sums from LDS, the eikonal sums by the last ray - against the separate launches of the same
bf16 path (SDF kernel, colour head, compositor, eikonal reduce) and against the reference's golden vectors.

What must be equal bit for bit: everything that does not pass through the colour head - the SDF kernel's arithmetic is the same
code (sdf, `gradients`), so `cdf_fine`, `inside_sphere`, `weights`, `weight_sum`, `weight_max`, `s_val`, `z_vals` and
`gradient_error` (same partial sums, reduced in the same order by the last ray as by eikonal_reduce_kernel). The colour differs by
the rounding of ONE input of the colour head's first layer: the normal's z component enters as an f32 term in the fused kernel
and as a bf16 operand in rendernet_fwd_kernel."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def g(x, dev):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev)


def _render(rend, batch, fused, **kw):
    os.environ["VDN_SHADE_FUSED"] = "1" if fused else "0"
    try:
        with torch.no_grad():
            return rend.render(*batch, background_rgb=torch.ones(1, 3, device=batch[0].device), cos_anneal_ratio=0.5, **kw)
    finally:
        del os.environ["VDN_SHADE_FUSED"]


def _batch(B, dev, seed=0, crop=None):
    from vdn_train import synth
    cams = synth.make_cameras(seed)
    o, d = synth.random_pixel_batch(seed, 3, 2, B, cams=cams, crop=crop)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, 3, B)
    return (g(o, dev), g(d, dev), g(near, dev), g(far, dev)), dict(t_rand=g(t1, dev), t_rand_out=g(t2, dev))


EXACT = ("cdf_fine", "inside_sphere", "weights", "weight_sum", "weight_max", "s_val", "z_vals", "gradients", "gradient_error")


@pytest.mark.parametrize("B,crop", [(1, None), (3, 420), (37, None), (512, None), (512, 420), (1000, 420)])
def test_fused_shading_equals_the_separate_launches(B, crop):
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(0, variance=0.4), precision="bf16")
    assert rend._fused_shading(128) and rend.shade_launches() == 1
    batch, kw = _batch(B, dev, crop=crop)
    ref = _render(rend, batch, False, **kw)
    out = _render(rend, batch, True, **kw)
    again = _render(rend, batch, True, **kw)             # the arrival counter was left at zero
    assert set(out) == set(ref)
    for k in EXACT:
        assert torch.equal(out[k], ref[k]), k
        assert torch.equal(out[k], again[k]), k
    assert out["render_feats"] is None
    # the colour: one bf16 rounding of one input apart (measured ~1e-4; the bf16 path's own error vs fp32 is ~1e-3)
    assert float((out["color_fine"] - ref["color_fine"]).abs().max()) < 2e-3
    assert torch.equal(out["color_fine"], again["color_fine"])
    assert torch.isfinite(out["color_fine"]).all()


@pytest.mark.parametrize("name,min_psnr", [("white_v03_c0", 55.0), ("white_v065_c1", 42.0), ("black_v03", 55.0)])
def test_fused_shading_vs_reference_golden(golden, name, min_psnr):
    """The reference's own outputs (tests/golden, fp32) at the bf16 path's bounds (tests/test_gpu_bf16.py), through the one launch."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    fx = golden(name)
    st = synth.make_all_states(int(fx["seed"]), wdepth=False, variance=float(fx["variance"]))
    rend = factory.build_renderer(device=dev, states=st, precision="bf16")
    os.environ["VDN_SHADE_FUSED"] = "1"
    try:
        with torch.no_grad():
            bg = torch.ones(1, 3, device=dev) if name.startswith("white") else None
            out = rend.render(g(fx["rays_o"], dev), g(fx["rays_d"], dev), g(fx["near"], dev), g(fx["far"], dev), background_rgb=bg,
                              cos_anneal_ratio=float(fx["cos_anneal"]), t_rand=g(fx["t_rand"], dev), t_rand_out=g(fx["t_rand_out"], dev),
                              z_vals_inject=g(fx["z_vals_inside"], dev))
    finally:
        del os.environ["VDN_SHADE_FUSED"]
    mse = float(((out["color_fine"].cpu().numpy() - fx["out_color_fine"]) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-20)) > min_psnr
    assert np.abs(out["weight_sum"].cpu().numpy() - fx["out_weight_sum"]).max() < 2e-2
    assert abs(out["gradient_error"].item() - float(fx["out_gradient_error"])) < 5e-2 * max(float(fx["out_gradient_error"]), 1e-2)


@pytest.mark.parametrize("B", [5, 512])
def test_fused_shading_with_a_vdn_head(B):
    """womsk_white_wdepth (BASELINE.json configs[4]): the fused launch also writes the feature plane, the VDN head and the weighted
    sums of its 96 channels follow as two launches (3 instead of 6). `render_feats` depends on the colour head only through
    nothing at all - weights and VDN features are the same bits as the separate launches'."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(wdepth=True, device=dev, states=synth.make_all_states(3, wdepth=True, variance=0.4), precision="bf16")
    assert rend._fused_shading(128) and rend.shade_launches() == 3
    batch, kw = _batch(B, dev, seed=3)
    ref = _render(rend, batch, False, **kw)
    out = _render(rend, batch, True, **kw)
    for k in EXACT + ("render_feats",):
        assert torch.equal(out[k], ref[k]), k
    assert out["render_feats"].shape == (B, 96) and float(out["render_feats"].abs().max()) > 0
    assert float((out["color_fine"] - ref["color_fine"]).abs().max()) < 2e-3


def test_fused_shading_declines_what_it_does_not_cover():
    """depth_before_color, other sample counts, mixed precisions: the separate launches run (and the entry point itself answers -10 for N != 128)."""
    from vdn_train import synth, factory
    from vdn_hip import lib
    dev = torch.device("cuda:0")
    r = factory.build_renderer(wdepth=True, device=dev, states=synth.make_all_states(0, wdepth=True, depth_before_color=True),
                               precision="bf16", depth_before_color=True)
    assert not r._fused_shading(128, depth_before_color=True)       # the colour head would read the VDN head's output
    r = factory.build_renderer(device=dev, states=synth.make_all_states(0), precision="fp32")
    assert r._fused_shading(128)                         # (the fp32 path has its own one-launch kernel since round 5: vdn_shade_fused_f32)
    r.color_network.precision = "bf16"
    assert not r._fused_shading(128)                     # one precision per launch
    r = factory.build_renderer(device=dev, states=synth.make_all_states(0), precision="bf16", n_importance=0)
    assert not r._fused_shading(64)
    batch, kw = _batch(8, dev)
    out = _render(r, batch, True, **kw)                  # 64 samples per ray: the separate launches
    assert torch.isfinite(out["color_fine"]).all()
    sa, cm = lib.VdnSdfArgs(), lib.VdnCompositeArgs()
    z = torch.zeros(8, 64, device=dev)
    sa.blob = r.sdf_network._images().blobs["full"].data_ptr()
    sa.rays_o, sa.rays_d, sa.z, sa.sdf, sa.normals = batch[0].data_ptr(), batch[1].data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr()
    sa.n_per_ray, sa.z_ld, sa.sdf_ld, sa.P, sa.scale = 64, 64, 64, 8 * 64, 1.0
    cm.B, cm.N, cm.T = 8, 64, 96
    t = torch.zeros(1, dtype=torch.int32, device=dev)
    assert not lib.try_call("vdn_shade_fused_bf16", sa, lib.ptr(r.color_network._images().blobs["c2"]), 1, cm, lib.ptr(t),
                            torch.cuda.current_stream().cuda_stream)


def test_persistent_background_outputs_and_jitter_blocks():
    """render() keeps the background network's outputs in one buffer per stream (rows off the work list hold zeros or an earlier
    batch's finite values: render_core multiplies them by zero) and draws its jitter 16 batches at a time. Neither may show:
    a batch rendered behind a DIFFERENT batch equals the same batch on a fresh renderer bit for bit; the same seed gives the
    same image; consecutive batches get different jitter."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    st = synth.make_all_states(0, variance=0.4)
    r1 = factory.build_renderer(device=dev, states=st, precision="bf16")
    r2 = factory.build_renderer(device=dev, states=st, precision="bf16")
    a, kwa = _batch(512, dev, seed=0)
    b, kwb = _batch(512, dev, seed=4, crop=420)
    _render(r1, a, True, **kwa)
    out1 = _render(r1, b, True, **kwb)                   # behind batch a: its scratch buffer holds a's outputs
    out2 = _render(r2, b, True, **kwb)                   # fresh buffer: zeros
    for k in ("color_fine", "weights", "weight_sum", "gradient_error"):
        assert torch.equal(out1[k], out2[k]), k
    assert "_scratch" in r1.nerf.__dict__ and len(r1.nerf.__dict__["_scratch"]) == 1
    # jitter: seeded draws repeat, consecutive draws differ
    torch.manual_seed(11)
    z1 = _render(r1, a, True)["z_vals"].clone()
    z1b = _render(r1, a, True)["z_vals"].clone()
    torch.manual_seed(11)
    z2 = _render(r1, a, True)["z_vals"]
    assert torch.equal(z1, z2) and not torch.equal(z1, z1b)


@pytest.mark.parametrize("B,wdepth", [(1, False), (37, False), (512, False), (96, True)])
def test_fp32_fused_shading_equals_the_separate_launches_bit_for_bit(B, wdepth):
    """vdn_shade_fused_f32 (csrc/k_shade_f32.h): renderer.py:239-315 in ONE launch on the exact-fp32 kernels - the SDF network's body,
    the colour head's body and the compositor's body back to back in the workgroup that owns the ray, the eikonal term by the ray
    that finishes last. The same device code as vdn_sdf_mlp_fwd_f32 + vdn_rendernet_fwd_f32 + vdn_alpha_composite_fwd + the eikonal
    reduce: EVERY output of render() bit for bit, twice in a row (the arrival counter is left at zero), with a VDN head (its two
    launches follow) and without. The reference's golden cases run through this launch by default
    (tests/test_gpu_parity.py::test_render_vs_reference_golden: the 1e-4 bar on the kernel north_star names)."""
    from vdn_train import synth, factory
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(wdepth=wdepth, device=dev, states=synth.make_all_states(0, wdepth=wdepth, variance=0.4), precision="fp32")
    assert rend._fused_shading(128) and rend.shade_launches() == (3 if wdepth else 1)
    batch, kw = _batch(B, dev)
    ref = _render(rend, batch, False, **kw)
    out = _render(rend, batch, True, **kw)
    again = _render(rend, batch, True, **kw)
    assert set(out) == set(ref)
    for k, v in ref.items():
        if v is None:
            assert out[k] is None
            continue
        assert torch.equal(out[k], v), k
        assert torch.equal(again[k], v), k
    assert torch.isfinite(out["color_fine"]).all() and float(out["weight_sum"].max()) > 0.5
