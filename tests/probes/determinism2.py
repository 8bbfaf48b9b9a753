"""Two trainers in lockstep on identical inputs: report the first step at which their gradients differ, per network and per
workspace tensor (which kernel family first produces different bits)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
from vdn_train.trainer import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
B, seed = 512, 0
cams = synth.make_cameras(seed)
g = lambda x: torch.tensor(x).to(dev)
trs = []
for k in range(2):
    torch.manual_seed(0)
    rend = factory.build_renderer(device=dev, precision="bf16")
    trs.append(Trainer(rend, B, dev, conf=dict(warm_up_end=50, end_iter=300, anneal_end=75)))
names = []
for tr in trs[:1]:
    off = 0
    for m, nm in ((tr.r.nerf, "nerf"), (tr.r.sdf_network, "sdf"), (tr.r.deviation_network, "var"), (tr.r.color_network, "color")):
        n = sum(p.numel() for p in m.parameters())
        names.append((nm, off, off + n)); off += n
bad = 0
for it in range(steps):
    o, d = synth.random_pixel_batch(seed, it, it % 40, B, cams=cams, crop=420)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, it, B)
    args = [g(o), g(d), g(near), g(far), g(synth.target_colors(o, d, 0.5))]
    outs = [tr.train_step(*args, t_rand=g(t1), t_rand_out=g(t2)).clone() for tr in trs]
    torch.cuda.synchronize()
    ga, gb = trs[0].engine.grad_flat, trs[1].engine.grad_flat
    if not torch.equal(ga, gb):
        bad += 1
        msg = []
        for nm, a, b in names:
            nd = (ga[a:b] != gb[a:b]).sum().item()
            if nd: msg.append("%s:%d" % (nm, nd))
        wa, wb = trs[0].engine.w, trs[1].engine.w
        wdiff = []
        for k in wa:
            ta, tb = wa[k], wb[k]
            if torch.is_tensor(ta) and ta.shape == tb.shape and not torch.equal(ta, tb):
                wdiff.append(k)
        print("step %d grads differ: %s | workspaces differing: %s" % (it, " ".join(msg), " ".join(wdiff[:30])))
        # resynchronise trainer 1 to trainer 0 so that the NEXT difference is again a first difference
        trs[1].param_flat.copy_(trs[0].param_flat); trs[1].exp_avg.copy_(trs[0].exp_avg); trs[1].exp_avg_sq.copy_(trs[0].exp_avg_sq)
        for net in trs[1].engine.nets.values(): net.img.invalidate()
        from vdn_hip import images
        images.refresh_together([net.img for net in trs[1].engine.nets.values()], torch.cuda.current_stream().cuda_stream, trs[1]._img_cache)
        if bad >= 6: break
print("steps", it + 1, "steps with differing gradients:", bad)
