"""Step time of the data-parallel code path on ONE GPU (a one-rank RCCL group, collectives forced on): what the N > 1 step
launches, minus the wire. usage: dp1_wall.py [tag]   (VDN_DP_FUSED=0: the unfused arm)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
dist.init_process_group("nccl", rank=0, world_size=1)
from vdn_train import synth, factory
from vdn_train.trainer import Trainer
dev = torch.device("cuda:0")
B, seed = 512, 0
cams = synth.make_cameras(seed)
gg = lambda x: torch.tensor(x).to(dev)
torch.manual_seed(0)
OFF = os.environ.get("DP1_OFF") == "1"           # the same step without collectives: the denominator of the ratio
tr = Trainer(factory.build_renderer(device=dev, precision="bf16"), B, dev, collectives=not OFF)
batches = []
for it in range(23):
    o, d = synth.random_pixel_batch(seed, it, it % 40, B, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    batches.append([gg(o), gg(d), gg(near), gg(far), gg(synth.target_colors(o, d, 0.5))])
for i in range(700):
    tr.train_step(*batches[i % 23])
res = []
for r in range(6):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(60):
        tr.train_step(*batches[i % 23])
    torch.cuda.synchronize(); res.append((time.time() - t0) / 60 * 1e6)
tr.coll.timing = not OFF
for i in range(40):
    tr.train_step(*batches[i % 23])
torch.cuda.synchronize()
print({k: round(v["mean_ms"] * 1e3, 1) for k, v in tr.coll.exposed_ms().items()}, "us exposed per call")
print("%-12s one-rank DP step %.1f us (min %.1f)" % (sys.argv[1] if len(sys.argv) > 1 else "dp", float(np.median(res)), min(res)))
dist.destroy_process_group()
