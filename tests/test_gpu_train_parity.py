"""Training parity on the MI355X: the Trainer's loss / PSNR curves over identical steps against the CPU oracle's
autograd + torch.optim.Adam loop (= the reference's Runner.train arithmetic)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))


@pytest.mark.parametrize("precision,loss_tol,psnr_tol", [("fp32", 1e-3, 0.02), ("bf16", 3e-2, 0.3)])
def test_training_curves_follow_the_oracle(precision, loss_tol, psnr_tol):
    import train_parity
    ref, got = train_parity.run(steps=24, B=32, precision=precision)
    assert np.isfinite(got).all()
    rel = np.abs(got[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
    assert rel.max() < loss_tol, (precision, rel.max(), rel.argmax())
    assert np.abs(got[:, 1] - ref[:, 1]).max() < psnr_tol, (precision, np.abs(got[:, 1] - ref[:, 1]).max())
    assert ref[-1, 0] < ref[0, 0]                      # and the loss is actually going down


def test_training_is_bitwise_reproducible():
    import torch
    """Two trainers in lockstep on identical inputs must produce identical gradient bits at every step. The MLP kernels count
    their own LDS-DMA / store completions (s_waitcnt vmcnt(N) with compile-time N): a count that is too large releases a wait
    while a weight chunk is still in flight, which shows up as run-to-run differences in a few percent of the steps long
    before it shows up in a tolerance test (it did, once: tests/probes/determinism2.py)."""
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 512, 0
    cams = synth.make_cameras(seed)
    gg = lambda x: torch.tensor(x).to(dev)
    for wdepth in (False, True):
        trs = []
        for _ in range(2):
            torch.manual_seed(0)
            rend = factory.build_renderer(wdepth=wdepth, device=dev, precision="bf16")
            trs.append(Trainer(rend, B, dev, conf=dict(warm_up_end=50, end_iter=300, anneal_end=75, extract_depth=wdepth, depth_start_iter=-1)))
        feats = gg(synth.uniform(seed, "repro/feats", (B, 96)).astype(np.float32)) if wdepth else None
        for it in range(60 if not wdepth else 30):
            o, d = synth.random_pixel_batch(seed, it, it % 40, B, cams=cams, crop=420)
            near, far = synth.near_far_from_sphere(o, d)
            t1, t2 = synth.jitter(seed, it, B)
            args = [gg(o), gg(d), gg(near), gg(far), gg(synth.target_colors(o, d, 0.5))]
            for tr in trs:
                tr.train_step(*args, gt_feats=feats, t_rand=gg(t1), t_rand_out=gg(t2))
            assert torch.equal(trs[0].engine.grad_flat, trs[1].engine.grad_flat), (wdepth, it)
        assert torch.equal(trs[0].param_flat, trs[1].param_flat)


def test_overlapped_schedule_is_bit_identical_to_the_in_order_one():
    """Trainer(overlap=True) - the default: the colour / VDN / background networks' weight-gradient GEMM, their Adam step and
    their weight images on the side stream, beside the next step's sampler - launches the same kernels on the same data as the
    in-order schedule: losses, gradients, parameters and Adam moments must be identical bit for bit at every step (a missing
    stream dependency shows up here as a difference)."""
    import torch
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 512, 3
    cams = synth.make_cameras(seed)
    gg = lambda x: torch.tensor(x).to(dev)
    conf = dict(warm_up_end=20, end_iter=300, anneal_end=40, extract_depth=True, depth_start_iter=5)
    trs = []
    for ov in (True, False):
        torch.manual_seed(0)
        trs.append(Trainer(factory.build_renderer(wdepth=True, device=dev, precision="bf16"), B, dev, conf=conf, overlap=ov))
    assert trs[0].overlap and not trs[1].overlap
    feats = gg(synth.uniform(seed, "ovl/feats", (B, 96)).astype(np.float32))
    for it in range(24):
        o, d = synth.random_pixel_batch(seed, it, it % 40, B, cams=cams, crop=420)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, it, B)
        args = [gg(o), gg(d), gg(near), gg(far), gg(synth.target_colors(o, d, 0.5))]
        sc = [tr.train_step(*args, gt_feats=feats, t_rand=gg(t1), t_rand_out=gg(t2)).clone() for tr in trs]
        assert torch.equal(sc[0], sc[1]), it
        if it % 4 == 3:       # (reading the buffers joins the streams: not at every step, so that steps do overlap)
            assert torch.equal(trs[0].engine.grad_flat, trs[1].engine.grad_flat), it
    assert torch.equal(trs[0].param_flat, trs[1].param_flat)
    assert torch.equal(trs[0].exp_avg, trs[1].exp_avg) and torch.equal(trs[0].exp_avg_sq, trs[1].exp_avg_sq)
    assert trs[0]._depth_adam_steps == trs[1]._depth_adam_steps > 0


@pytest.mark.parametrize("crop", [None, 420])
def test_fused_compositor_launch_is_bit_identical_to_the_four_launches(monkeypatch, crop):
    """Plain configuration (womsk_white: no mask loss, no VDN head, one rank): vdn_composite_train - the compositor, the colour
    term's gradient and the compositor's adjoint of one ray in one launch, the eikonal denominator taken from the foreground work
    list's length, the loss scalars reduced on a stream of their own - against vdn_alpha_composite_fwd + eikonal reduce +
    vdn_loss_fwd_bwd + vdn_alpha_composite_bwd (VDN_FUSED_COMPOSITE=0): the reported scalars, the gradients, the parameters and
    the Adam moments of every step, bit for bit (full-frame pixels: the list is a strict subset; object-centric crop: all rows)."""
    import torch
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 512, 5
    cams = synth.make_cameras(seed)
    gg = lambda x: torch.tensor(x).to(dev)
    conf = dict(warm_up_end=10, end_iter=300, anneal_end=40)
    trs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("VDN_FUSED_COMPOSITE", fused)
        torch.manual_seed(0)
        tr = Trainer(factory.build_renderer(device=dev, precision="bf16"), B, dev, conf=conf)
        trs.append(tr)
    for it in range(16):
        o, d = synth.random_pixel_batch(seed, it, it % 40, B, cams=cams, crop=crop)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, it, B)
        args = [gg(o), gg(d), gg(near), gg(far), gg(synth.target_colors(o, d, 0.5))]
        sc = []
        for fused, tr in zip(("1", "0"), trs):
            monkeypatch.setenv("VDN_FUSED_COMPOSITE", fused)
            sc.append(tr.train_step(*args, t_rand=gg(t1), t_rand_out=gg(t2)).clone())
            assert bool(tr.engine._fused_keep is not None) if fused == "1" else True
        assert torch.equal(sc[0], sc[1]), (it, sc[0].tolist(), sc[1].tolist())
        if it % 4 == 3:
            assert torch.equal(trs[0].engine.grad_flat, trs[1].engine.grad_flat), it
            assert torch.equal(trs[0].g_color, trs[1].g_color)
    assert torch.equal(trs[0].param_flat, trs[1].param_flat)
    assert torch.equal(trs[0].exp_avg, trs[1].exp_avg) and torch.equal(trs[0].exp_avg_sq, trs[1].exp_avg_sq)
    assert torch.isfinite(trs[0].param_flat).all()


def test_loss_gradients_inside_the_compositor_launches_with_the_vdn_head(monkeypatch):
    """womsk_white_wdepth, one rank, no mask: vdn_composite_fwd_train (the features' weighted sums write d loss / d render_feats on
    the spot) + vdn_composite_bwd_train (the colour term's gradient made inside the compositor's adjoint, the eikonal denominator
    from the foreground list's length), the eikonal reduce and the loss kernel on the logging stream - against
    vdn_alpha_composite_fwd + vdn_loss_fwd_bwd + vdn_alpha_composite_bwd on the critical path (VDN_FUSED_COMPOSITE=0): scalars,
    gradients, parameters and Adam moments of every step, bit for bit; the depth term switches on at step 4."""
    import torch
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 512, 7
    cams = synth.make_cameras(seed)
    gg = lambda x: torch.tensor(x).to(dev)
    conf = dict(warm_up_end=10, end_iter=300, anneal_end=40, extract_depth=True, depth_start_iter=3)
    trs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("VDN_FUSED_COMPOSITE", fused)
        torch.manual_seed(0)
        trs.append(Trainer(factory.build_renderer(wdepth=True, device=dev, precision="bf16"), B, dev, conf=conf))
    feats = gg(synth.uniform(seed, "fusedwd/feats", (B, 96)).astype(np.float32))
    used = 0
    for it in range(16):
        o, d = synth.random_pixel_batch(seed, it, it % 40, B, cams=cams, crop=None if it % 2 else 420)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, it, B)
        args = [gg(o), gg(d), gg(near), gg(far), gg(synth.target_colors(o, d, 0.5))]
        sc = []
        for fused, tr in zip(("1", "0"), trs):
            monkeypatch.setenv("VDN_FUSED_COMPOSITE", fused)
            sc.append(tr.train_step(*args, gt_feats=feats, t_rand=gg(t1), t_rand_out=gg(t2)).clone())
        used += int(trs[0].__dict__.get("_g_log") is not None)
        assert torch.equal(sc[0], sc[1]), (it, sc[0].tolist(), sc[1].tolist())
        if it % 4 == 3:
            assert torch.equal(trs[0].engine.grad_flat, trs[1].engine.grad_flat), it
            assert torch.equal(trs[0].g_color, trs[1].g_color)
            if it > 4:                                    # (the depth term is in the loss from step 4 on)
                assert torch.equal(trs[0].g_feats, trs[1].g_feats)
    assert used >= 10 and trs[1].__dict__.get("_g_log") is None
    assert torch.equal(trs[0].param_flat, trs[1].param_flat)
    assert torch.equal(trs[0].exp_avg, trs[1].exp_avg) and torch.equal(trs[0].exp_avg_sq, trs[1].exp_avg_sq)
    assert trs[0]._depth_adam_steps == trs[1]._depth_adam_steps > 0
    assert torch.isfinite(trs[0].param_flat).all()


@pytest.mark.parametrize("wdepth", [False, True], ids=["womsk_white", "womsk_white_wdepth"])
def test_colour_head_inside_the_training_forward_launch(monkeypatch, wdepth):
    """vdn_sdf_color_train_bf16 (csrc/k_sdf_fwd2.h MODE 3: the SDF network's training forward and the colour head in ONE launch, the
    feature vector in registers) against vdn_sdf_mlp_fwd_bf16 + vdn_rendernet_fwd_bf16 (VDN_TRAIN_COLOR_FUSED=0). The SDF side is the
    same instruction stream: sdf, normals, the feature plane and the H / V / PE saves bit for bit. The colour head's first layer takes
    the normal's z component in f32 instead of bf16 (the 33rd small input as a rank-1 term): colours within 2e-4, its saved
    activations within a bf16 step or two, its small-input plane bit for bit - and a training run stays together
    (losses 1e-4 over 8 steps; with the VDN head the depth-feature term switches on at step 4)."""
    import torch
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B, seed = 512, 7
    cams = synth.make_cameras(seed)
    gg = lambda x: torch.tensor(x).to(dev)
    conf = dict(warm_up_end=10, end_iter=300, anneal_end=40)
    if wdepth:
        conf.update(extract_depth=True, depth_start_iter=3)
    feats = gg(synth.uniform(seed, "cf/feats", (B, 96)).astype("float32")) if wdepth else None
    trs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("VDN_TRAIN_COLOR_FUSED", fused)
        torch.manual_seed(0)
        trs.append(Trainer(factory.build_renderer(wdepth=wdepth, device=dev, precision="bf16"), B, dev, conf=conf))
    for it in range(8):
        o, d = synth.random_pixel_batch(seed, it, it % 40, B, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, it, B)
        args = [gg(o), gg(d), gg(near), gg(far), gg(synth.target_colors(o, d, 0.5))]
        sc = []
        for fused, tr in zip(("1", "0"), trs):
            monkeypatch.setenv("VDN_TRAIN_COLOR_FUSED", fused)
            sc.append(tr.train_step(*args, gt_feats=feats, t_rand=gg(t1), t_rand_out=gg(t2)).clone())
            assert bool(tr.engine._color_fused) == (fused == "1")
        if it == 0:
            # same parameters: compare what the two forwards left in the workspaces
            wa, wb = trs[0].engine.w, trs[1].engine.w
            torch.cuda.synchronize()
            n = int(wa["fg_active"][1].item())
            assert n == int(wb["fg_active"][1].item()) and 0 < n <= trs[0].engine.P
            rows = n // 32 * 32                 # (whole 32-row blocks of the list: the planes are tile-blocked, the rows behind the list undefined)
            for k in ("sdf", "normals"):
                assert torch.equal(wa[k], wb[k]), k
            for k in ("feat", "PE", "col_small"):
                assert torch.equal(wa[k][:rows], wb[k][:rows]), k
            for k in ("H", "V"):
                # [layer, 32-row block, 32-feature tile, 1024]: layer 3 has 217 outputs = 7 tiles, its eighth tile is never written
                ta, tb = (w_[k][:, :rows].reshape(8, rows // 32, 8, 1024) for w_ in (wa, wb))
                for l in range(8):
                    nt = 7 if l == 3 else 8
                    assert torch.equal(ta[l, :, :nt], tb[l, :, :nt]), (k, l)
            idx = wa["fg_active"][0][:n].long()
            dc = (wa["col_out"][idx] - wb["col_out"][idx]).abs().max().item()
            assert dc < 2e-4, dc
            ha, hb = wa["col_h"][:, :rows].float(), wb["col_h"][:, :rows].float()
            diff = (ha - hb).abs()
            # (every pre-activation of the first layer moves by the rounding of one input: a tenth of the activations land on the
            # neighbouring bf16 value, and the layers behind inherit it)
            assert (diff > 0).float().mean().item() < 0.3 and (diff <= 0.02 * hb.abs() + 2e-3).all()
        a, b = sc[0].cpu().numpy(), sc[1].cpu().numpy()
        assert abs(a[0] - b[0]) < 1e-4 * abs(b[0]) + 1e-6, (it, a.tolist(), b.tolist())
    pa, pb = trs[0].param_flat, trs[1].param_flat
    assert torch.isfinite(pa).all() and (pa - pb).norm().item() < 1e-3 * pb.norm().item()
