"""Run the same N training steps twice (fresh renderer each time) and compare the parameters bit for bit.
usage: determinism.py [steps] ; VDN_SIDE_STREAM=0/1 selects the side stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
from vdn_train.trainer import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
B, seed = 512, 0
cams = synth.make_cameras(seed)
g = lambda x: torch.tensor(x).to(dev)
def run():
    torch.manual_seed(0)
    rend = factory.build_renderer(device=dev, precision="bf16")
    tr = Trainer(rend, B, dev, conf=dict(warm_up_end=50, end_iter=steps, anneal_end=max(steps // 4, 1)))
    hist = []
    for it in range(steps):
        o, d = synth.random_pixel_batch(seed, it, it % 40, B, cams=cams, crop=420)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, it, B)
        sc = tr.train_step(g(o), g(d), g(near), g(far), g(synth.target_colors(o, d, 0.5)), t_rand=g(t1), t_rand_out=g(t2))
        if it % 50 == 0 or it == steps - 1:
            hist.append((it, float(sc[0].item())))
    return tr.param_flat.clone(), hist
a, ha = run()
b, hb = run()
diff = (a != b).sum().item()
print("side stream", os.environ.get("VDN_SIDE_STREAM", "1"), "steps", steps, "params differing:", diff, "of", a.numel(), "max abs", (a - b).abs().max().item())
for (i, x), (_, y) in zip(ha, hb):
    print("  step %4d loss %.8f %.8f %s" % (i, x, y, "" if x == y else "<-- differs"))
