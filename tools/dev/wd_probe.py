import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch
from vdn_train import synth, factory
from vdn_train.trainer import Trainer
dev = torch.device("cuda:0")
B, seed = 512, 0
cams = synth.make_cameras(seed)
gg = lambda x: torch.tensor(x).to(dev)
conf = dict(warm_up_end=50, end_iter=300, anneal_end=75, extract_depth=True, depth_start_iter=-1)
trs = []
for fused in ("1", "1", "0", "1"):
    os.environ["VDN_FUSED_COMPOSITE"] = fused
    torch.manual_seed(0)
    trs.append((fused, Trainer(factory.build_renderer(wdepth=True, device=dev, precision="bf16"), B, dev, conf=conf)))
feats = gg(synth.uniform(seed, "repro/feats", (B, 96)).astype(np.float32))
for it in range(3):
    o, d = synth.random_pixel_batch(seed, it, it % 40, B, cams=cams, crop=420)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(seed, it, B)
    args = [gg(o), gg(d), gg(near), gg(far), gg(synth.target_colors(o, d, 0.5))]
    outs = []
    for fused, tr in trs:
        os.environ["VDN_FUSED_COMPOSITE"] = fused
        sc = tr.train_step(*args, gt_feats=feats, t_rand=gg(t1), t_rand_out=gg(t2)).clone()
        torch.cuda.synchronize()
        outs.append((sc, tr.engine.grad_flat.clone(), tr.g_feats.clone(), tr.g_color.clone(), tr.engine.w["d_vdn"].clone(), tr.engine.w["d_bg_feat"].clone()))
    ref = outs[2]
    for k, (fused, tr) in enumerate(trs):
        o_ = outs[k]
        print(it, k, fused, "scalars", torch.equal(o_[0], ref[0]), "grad", torch.equal(o_[1], ref[1]), "g_feats", torch.equal(o_[2], ref[2]),
              "g_color", torch.equal(o_[3], ref[3]), "d_vdn", torch.equal(o_[4], ref[4]), "d_bg_feat", torch.equal(o_[5], ref[5]),
              float(o_[2].abs().max()), float(o_[1][:8].abs().max()))
