"""Train the shipped womsk_white configuration on the synthetic 800x800 scene and report loss / PSNR over time
(the 'PSNR vs ref' half of BASELINE.json's metric; dpt_runner.py:230 for the formula).

  train_psnr.py [steps] [precision] [--views N] [--checkpoints K] [--cross] [--out FILE]

Paired runs: the same seed, the same initial weights (torch.manual_seed(0): the reference's geometric init), the same
pixel stream (keyed by step) and the same schedule for `precision` = fp32 (the exact-f32 kernels, the path that holds
the 1e-4 parity) and bf16 (the headline path). Validation: N held-out views (never trained on), a 64 x 64 grid of rays
each, perturb off; per checkpoint the mean and the standard deviation over the views. --cross also renders the
checkpoint with the OTHER precision's kernels (same weights): separates what training in bf16 costs from what
rendering in bf16 costs."""
import argparse, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
from vdn_train.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("steps", nargs="?", type=int, default=3000)
ap.add_argument("precision", nargs="?", default="bf16")
ap.add_argument("--views", type=int, default=8)
ap.add_argument("--checkpoints", type=int, default=10)
ap.add_argument("--cross", action="store_true")
ap.add_argument("--out", default=None)
ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--seed", type=int, default=0, help="initial weights (torch.manual_seed) and pixel stream")
ap.add_argument("--quiet", action="store_true", help="only the summary line")
ap.add_argument("--rig", choices=["synthetic", "real"], default="synthetic", help="real: the 33 learned cameras + focal of a scene the "
                "reference ships (pretrained-models/pixiu/womsk_learn_white_colmap/pnf_300000.pth via tests/golden/pnf_rays.npz), "
                "pixels from the same central window; the analytic target scene is the same")
args = ap.parse_args()
steps, prec = args.steps, args.precision
dev = torch.device("cuda:0")
torch.manual_seed(args.seed)
B, seed = 512, args.seed
# geometric init exactly as the reference constructs it (fields.py:45-63); no synthetic perturbation
rend = factory.build_renderer(device=dev, precision=prec)
# shortened schedule so that a few thousand steps cover warm-up and annealing
tr = Trainer(rend, B, dev, conf=dict(warm_up_end=200, end_iter=steps, anneal_end=max(steps // 4, 1)))
other = None
if args.cross:
    oprec = "fp32" if prec == "bf16" else "bf16"
    other = factory.build_renderer(device=dev, precision=oprec)
cams, FOCAL, NCAM = synth.make_cameras(seed), synth.FOCAL, 40
g = lambda x: torch.tensor(x).to(dev)
ALBEDO = 0.5
CROP = 420            # train on the central window where the object covers most pixels (object-centric capture)
HELD = [3, 7, 13, 18, 23, 27, 33, 38][:args.views]          # never trained on
if args.rig == "real":
    pn = np.load(os.path.join(ROOT, "tests", "golden", "pnf_rays.npz"), allow_pickle=False)
    tag = "pixiu.womsk_learn_white_colmap"
    cams, FOCAL = pn[tag + "__c2w"].astype(np.float64), float(pn[tag + "__fx"]) ** 2 * 800.0      # poses.py:80-84: focal = fx^2 * W
    NCAM = len(cams)          # (same central window as on the synthetic rig: on full frames this scene's white background wins and the
                              # level set collapses to empty space in both precisions - 4.00 dB in all 16 runs of a first attempt)
    HELD = [3, 7, 11, 15, 19, 23, 27, 31][:args.views]
G = args.grid
vx, vy = np.meshgrid(np.linspace(190, 610, G), np.linspace(190, 610, G))
val = []
for v in HELD:
    vo, vd = synth.pixel_rays(cams[v], vx.reshape(-1), vy.reshape(-1), focal=FOCAL)
    vn, vf = synth.near_far_from_sphere(vo, vd)
    val.append([g(vo), g(vd), g(vn), g(vf), g(synth.target_colors(vo, vd, ALBEDO))])
white = torch.ones(1, 3, device=dev)


def psnr_views(r):
    """PSNR (dpt_runner.py:230 with mask = 1) of every held-out view rendered by renderer r."""
    res = []
    with torch.no_grad():
        for vo, vd, vn, vf, vt in val:
            cols = [r.render(vo[i:i + 512], vd[i:i + 512], vn[i:i + 512], vf[i:i + 512], perturb_overwrite=0, background_rgb=white,
                             cos_anneal_ratio=tr.cos_anneal_ratio())["color_fine"] for i in range(0, vo.shape[0], 512)]
            c = torch.cat(cols)
            res.append((20.0 * torch.log10(1.0 / ((c - vt) ** 2).mean().sqrt())).item())
    return res


def sync_other():
    """Copy the trained weights into the other-precision renderer (same modules, other kernels)."""
    with torch.no_grad():
        for a, b in ((rend.nerf, other.nerf), (rend.sdf_network, other.sdf_network), (rend.deviation_network, other.deviation_network),
                     (rend.color_network, other.color_network)):
            for pa, pb in zip(a.parameters(), b.parameters()):
                pb.copy_(pa)


log = []
out_f = open(args.out, "w") if args.out else None
t0 = time.time()
order = (synth.uniform(seed, "trainperm", (steps,)) * NCAM).astype(np.int64) % NCAM
free = [i for i in range(NCAM) if i not in HELD]
every = max(steps // args.checkpoints, 1)
for it in range(steps):
    img = free[int(order[it]) % len(free)]
    o, d = synth.random_pixel_batch(seed, it, img, B, cams=cams, crop=CROP, focal=FOCAL)
    near, far = synth.near_far_from_sphere(o, d)
    sc = tr.train_step(g(o), g(d), g(near), g(far), g(synth.target_colors(o, d, ALBEDO)))
    if (it + 1) % every == 0 or it == 0 or it == steps - 1:
        s = sc.cpu().numpy()
        tr.join()                                 # (the networks' own weight-image accessor joins too)
        pv = psnr_views(rend)
        rec = dict(step=it + 1, precision=prec, loss=float(s[0]), train_psnr=float(s[2]), eikonal=float(s[3]),
                   val_psnr_mean=float(np.mean(pv)), val_psnr_sd=float(np.std(pv)), val_psnr_views=[round(x, 3) for x in pv],
                   inv_s=float(torch.exp(rend.deviation_network.variance * 10).item()), wall_s=time.time() - t0)
        if other is not None:
            sync_other()
            po = psnr_views(other)
            rec["val_psnr_mean_on_%s_kernels" % oprec] = float(np.mean(po))
            rec["val_psnr_sd_on_%s_kernels" % oprec] = float(np.std(po))
        log.append(rec)
        line = json.dumps(rec)
        if not args.quiet:
            print(line, flush=True)
        if out_f:
            out_f.write(line + "\n")
            out_f.flush()
last = [r["val_psnr_mean"] for r in log[-3:]]
summary = {"precision": prec, "rig": args.rig, "steps": steps, "seed": args.seed, "views": HELD, "final_val_psnr_mean": log[-1]["val_psnr_mean"], "final_val_psnr_sd": log[-1]["val_psnr_sd"],
           "mean_of_last_3_checkpoints": float(np.mean(last)), "final_train_psnr": log[-1]["train_psnr"], "wall_s": time.time() - t0}
print(json.dumps(summary))
if out_f:
    out_f.write(json.dumps(summary) + "\n")
    out_f.close()
