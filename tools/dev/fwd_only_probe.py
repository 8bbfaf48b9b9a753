"""bench.py's forward_only figure taken apart (dev): render() of the Trainer's renderer after training steps - host enqueue time
vs device time, with / without the Trainer's cold-start flag.  usage: fwd_only_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse, torch, bench
leg = bench.Leg(argparse.Namespace(batch=512), torch.device("cuda:0"), 1, 0, "bf16", False, 64)
for i in range(300):
    leg.step(i)
torch.cuda.synchronize()
bg = torch.ones(1, 3, device="cuda:0")
nb = len(leg.batches)
def run(tag, n=200):
    with torch.no_grad():
        for i in range(5):
            leg.rend.render(*leg.batches[i % nb][:4], background_rgb=bg, cos_anneal_ratio=0.5)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(n):
            leg.rend.render(*leg.batches[i % nb][:4], background_rgb=bg, cos_anneal_ratio=0.5)
        t1 = time.time()
        torch.cuda.synchronize()
        t2 = time.time()
    print("%-28s host %.1f us/call, total %.1f us/call = %.3f M rays/s" % (tag, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, 512 * n / (t2 - t0) / 1e6))
run("trainer's renderer")
run("trainer's renderer (again)")
leg.rend.sdf_network.__dict__["_cold_start"] = False
run("cold_start off")
for m in (leg.rend.nerf, leg.rend.color_network):
    m.__dict__.pop("_stream_join", None)
run("join hooks off too")
