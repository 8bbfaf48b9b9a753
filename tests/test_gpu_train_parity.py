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
