import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
from vdn_train.trainer import Trainer
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, seed = 512, 0
rend = factory.build_renderer(device=dev, precision=prec)
tr = Trainer(rend, B, dev, conf=dict(warm_up_end=200, end_iter=20000, anneal_end=5000))
cams = synth.make_cameras(seed)
g = lambda x: torch.tensor(x).to(dev)
probe = g(np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, -0.5], [0.8, 0, 0], [0.3, 0.3, 0.3]], dtype=np.float32))
for it in range(steps):
    img = it % 40
    o, d = synth.random_pixel_batch(seed, it, img, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    rgb = synth.target_colors(o, d, 0.5)
    sc = tr.train_step(g(o), g(d), g(near), g(far), g(rgb))
    if it % 100 == 0 or it == steps - 1:
        w = tr.engine.w
        hit = torch.tensor((rgb < 0.999).any(-1)).to(dev)
        ws = w["weights"].sum(-1)
        col = w["color"]
        err = (col - g(rgb)).abs().sum(-1)
        with torch.no_grad():
            sd = rend.sdf_network.sdf(probe)[:, 0].cpu().numpy()
        s = sc.cpu().numpy()
        print("it %5d loss %.4f eik %.4f | hit rays %3d: wsum %.3f err %.3f col %.3f | miss: wsum %.3f err %.3f | sdf(0) %.3f sdf(r=.5) %.3f %.3f %.3f sdf(.8) %.3f | inv_s %.1f"
              % (it, s[0], s[3], int(hit.sum()), ws[hit].mean().item(), err[hit].mean().item(), col[hit].mean().item(), ws[~hit].mean().item(), err[~hit].mean().item(),
                 sd[0], sd[1], sd[2], sd[3], sd[4], float(torch.exp(rend.deviation_network.variance * 10))), flush=True)
