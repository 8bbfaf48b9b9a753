"""What else is cold inside the step? The step's SDF forward launch (work list, training saves, warm-up on), one-stream leg:
back to back / behind 400 MB of STORES (dirty lines in L2 + MALL) / behind 400 MB of LOADS (clean eviction) / behind both.
usage: cold_probe.py"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
os.environ.update(VDN_SIDE_STREAM="0", VDN_OVERLAP="0")
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
leg = bench.Leg(argparse.Namespace(batch=512), dev, 1, 0, "bf16", False, 32)
for i in range(640):
    leg.step(i)
eng = leg.trainer.engine
o, d = leg.batches[0][0], leg.batches[0][1]
launch = lambda: eng._sdf_forward(o, d)
trash = torch.empty(100 << 20, dtype=torch.float32, device=dev)
trash2 = torch.ones(100 << 20, dtype=torch.float32, device=dev)

def timed(pre, n=20):
    ts = []
    for i in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ts]) * 1e3
    return "median %6.1f  min %6.1f  max %6.1f us" % (np.median(t), t.min(), t.max())

print("rows", int(eng.w["fg_active"][1].item()), "(main launch + tail launch, warm-up on)")
print("back to back:                 ", timed(lambda: None))
print("behind 400 MB of stores:      ", timed(lambda: trash.zero_()))
print("behind 400 MB of loads:       ", timed(lambda: trash2.sum()))
# code warmth: behind the 400 MB of loads, the SAME kernel on an 8-ray engine (8 workgroups = one per XCD under round-robin
# placement: they pull the kernel's 137 KB of code - and the weight stream - into every L2) in front of the timed launch
from vdn_hip.train import TrainEngine
small = TrainEngine(leg.rend, 8, dev)
small._fg_compact = False
small.w["mid_z"].copy_(eng.w["mid_z"][:8])
so, sd = o[:8].contiguous(), d[:8].contiguous()
t400 = torch.ones(100 << 20, dtype=torch.float32, device=dev)
print("behind 400 MB of loads + the same kernel on 8 workgroups:", timed(lambda: (t400.sum(), small._sdf_forward(so, sd))))
del t400
for mb in (32, 64, 128, 256, 800):
    t = torch.ones(mb << 18, dtype=torch.float32, device=dev)
    print("behind %4d MB of loads:       " % mb, timed(lambda: t.sum()))
    del t
big = torch.zeros(6 << 28, dtype=torch.float32, device=dev)         # 6 GiB: 3072 pages of 2 MiB
idx = (torch.arange(3072, device=dev) * (2 << 20) // 4)
print("behind one word from each of 3072 2-MiB pages (TLB only):", timed(lambda: big[idx].sum()))
del big
situ = leg.sdf_in_situ(40)
print("in situ: mean %.1f us at %.0f rows" % (situ["kernel_ms"] * 1e3, situ["points"]))
