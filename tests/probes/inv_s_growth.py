"""Does the variance take off within 2 000 steps of the synthetic scene? Per torch seed (the per-ray jitter draws), bf16 and fp32
training from the same geometric init: final inv_s. Separates the sampler-jitter sensitivity of the early training dynamics
from the precision of the kernels (tests/test_gpu_bf16.py seeds its run)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
from vdn_train.trainer import Trainer
dev = torch.device("cuda:0")
B, seed, steps = 512, 0, 2000
cams = synth.make_cameras(seed)
gg = lambda x: torch.tensor(x).to(dev)
order = (synth.uniform(seed, "trainperm", (steps,)) * 40).astype(np.int64) % 40
for prec in sys.argv[1].split(","):
    for ts in range(int(sys.argv[3]) if len(sys.argv) > 3 else 0, int(sys.argv[2])):
        torch.manual_seed(ts)
        rend = factory.build_renderer(device=dev, precision=prec)
        tr = Trainer(rend, B, dev, conf=dict(warm_up_end=200, end_iter=steps, anneal_end=steps // 4))
        trace = []
        for it in range(steps):
            img = int(order[it]) if int(order[it]) != 7 else 8
            o, d = synth.random_pixel_batch(seed, it, img, B, cams=cams, crop=420)
            near, far = synth.near_far_from_sphere(o, d)
            tr.train_step(gg(o), gg(d), gg(near), gg(far), gg(synth.target_colors(o, d, 0.5)))
            if it % 500 == 499:
                trace.append(float(torch.exp(rend.deviation_network.variance * 10).item()))
        print(prec, "torch seed", ts, "inv_s at 500/1000/1500/2000:", " ".join("%.4f" % v for v in trace), flush=True)
