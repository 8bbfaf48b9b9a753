"""Does a high-priority caller stream (the side stream keeps the default priority) change the step time? (dev probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import torch
import bench

class A: pass
a = A(); a.batch = 512
dev = torch.device("cuda", 0)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
leg = bench.Leg(a, dev, 1, 0, "bf16", False, 32)
for i in range(700):
    leg.step(i)
torch.cuda.synchronize()

def run(stream, n=400):
    torch.cuda.synchronize()
    t0 = time.time()
    if stream is None:
        for i in range(n):
            leg.step(i)
    else:
        with torch.cuda.stream(stream):
            for i in range(n):
                leg.step(i)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e6
hi = torch.cuda.Stream(device=dev, priority=-1)
for rep in range(6):
    print("default stream %.1f us/step   high-priority stream %.1f us/step" % (run(None), run(hi)))
