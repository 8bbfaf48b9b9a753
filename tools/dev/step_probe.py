"""N training steps of the bench configuration and nothing else: wall time per step, for comparison with the summed kernel
time of a `rocprofv3 --kernel-trace --stats` run of the same command (GPU idle share = launch gaps).  usage: step_probe.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
args = argparse.Namespace(batch=512)
leg = bench.Leg(args, torch.device("cuda:0"), 1, 0, "bf16", False, 16)
for i in range(20):
    leg.step(i)
torch.cuda.synchronize()
t0 = time.time()
for i in range(n):
    leg.step(20 + i)
t_issue = time.time() - t0
torch.cuda.synchronize()
dt = time.time() - t0
print("steps %d  wall %.1f us/step  (host finished issuing after %.1f us/step)" % (n, 1e6 * dt / n, 1e6 * t_issue / n))
eng = leg.trainer.engine
rows = []
for i in range(16):
    leg.step(i)
    rows.append((int(eng.w["fg_active"][1].item()), int(eng.w["bg_active"][1].item())))
print("work-list rows (foreground of %d, background of %d):" % (eng.P, eng.Q), rows)
print("dW GEMM algorithmic bytes %.0f MB, flops %.1f G" % (eng.dw_bytes() / 1e6, eng.dw_flops() / 1e9))
