"""Training parity (SURVEY.md 8d 'PSNR vs ref' item 2): the same K steps - same weights, rays, jitter, schedules - taken by the
CPU oracle (autograd + torch.optim.Adam, i.e. what the reference's Runner.train does) and by the MI355X Trainer (HIP forward /
hand-derived backward / fused Adam); returns both loss / PSNR curves.

  python tests/probes/train_parity.py [steps] [batch] [precision]     -> gpurun_out/train_parity_<precision>.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "vdn-nerf_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

CONF = dict(warm_up_end=50, end_iter=2000, anneal_end=100)       # short schedules so lr and cos-anneal actually move


def batches(seed, steps, B):
    from vdn_train import synth
    cams = synth.make_cameras(seed)
    out = []
    for it in range(steps):
        o, d = synth.random_pixel_batch(seed, it, it % len(cams), B, cams=cams, crop=420)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, it, B)
        out.append((o, d, near, far, synth.target_colors(o, d, albedo=0.5), t1, t2))
    return out


def run_oracle(st, data, dtype=torch.float32):
    import oracle.neus_oracle as orc
    nets = orc.nets_from_numpy(st, dtype=dtype, requires_grad=True)
    params = [p for _, p in orc.all_params(nets)]
    opt = torch.optim.Adam(params, lr=5e-4)
    tt = lambda x: torch.tensor(np.asarray(x), dtype=dtype)
    curve = []
    for it, (o, d, near, far, rgb, t1, t2) in enumerate(data):
        for gq in opt.param_groups:
            gq["lr"] = 5e-4 * orc.learning_rate_factor(it, CONF["warm_up_end"], CONF["end_iter"])
        out = orc.render(nets, tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3, dtype=dtype),
                         cos_anneal_ratio=orc.cos_anneal_ratio(it, CONF["anneal_end"]), t_rand=tt(t1), t_rand_out=tt(t2))
        lo = orc.loss_from_render(out, tt(rgb))
        opt.zero_grad()
        lo["loss"].backward()
        opt.step()
        curve.append((lo["loss"].item(), lo["psnr"].item()))
    return np.array(curve)


def run_gpu(st, data, precision):
    from vdn_train import factory
    from vdn_train.trainer import Trainer
    dev = torch.device("cuda:0")
    B = data[0][0].shape[0]
    tr = Trainer(factory.build_renderer(device=dev, states=st, precision=precision), B, dev, conf=CONF)
    g = lambda x: torch.tensor(x).to(dev)
    curve = []
    for o, d, near, far, rgb, t1, t2 in data:
        sc = tr.train_step(g(o), g(d), g(near), g(far), g(rgb), t_rand=g(t1), t_rand_out=g(t2))
        curve.append((sc[0].item(), sc[2].item()))
    return np.array(curve)


def run(steps=30, B=32, precision="fp32", seed=9):
    from vdn_train import synth
    st = synth.make_all_states(seed, wdepth=False)
    data = batches(seed, steps, B)
    return run_oracle(st, data), run_gpu(st, data, precision)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    if prec == "cpu_fp64":          # calibration, CPU only: how far the oracle's own fp32 run drifts from its fp64 run
        from vdn_train import synth
        st, data = synth.make_all_states(9, wdepth=False), batches(9, steps, B)
        ref, got = run_oracle(st, data, torch.float64), run_oracle(st, data, torch.float32)
    else:
        ref, got = run(steps, B, prec)
    rel = np.abs(got[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
    res = {"steps": steps, "batch": B, "precision": prec, "loss_rel_diff_max": float(rel.max()), "loss_rel_diff_mean": float(rel.mean()),
           "psnr_abs_diff_max_db": float(np.abs(got[:, 1] - ref[:, 1]).max()),
           "oracle_loss": ref[:, 0].tolist(), "gpu_loss": got[:, 0].tolist(), "oracle_psnr": ref[:, 1].tolist(), "gpu_psnr": got[:, 1].tolist()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "train_parity_%s.json" % prec), "w"))
    print({k: v for k, v in res.items() if not isinstance(v, list)})
