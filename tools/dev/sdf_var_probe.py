"""Why does the SAME 65 536-row training-mode launch of the fused SDF kernel take 159 us in one bench leg and 181-187 us in
another? One process, one kernel, everything else varied:
  (a) buffer addresses: three engines built one after the other (each allocates its own planes);
  (b) what ran just before: idle chip (1 s sleep) / a burst of the same launch / 200 training steps (hot chip, other data in
      L2 / MALL);
  (c) position in a long back-to-back run of the same launch (clock settling): launches 1-10 against 491-500.
usage: sdf_var_probe.py      -> one table on stdout"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np
import torch
import bench

dev = torch.device("cuda:0")
args = argparse.Namespace(batch=512)


def launches(fn, n):
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for i in range(n):
        e0[i].record()
        fn()
        e1[i].record()
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in zip(e0, e1)]) * 1e3


def fmt(t):
    return "median %6.1f  min %6.1f  max %6.1f us" % (np.median(t), t.min(), t.max())


legs = []
for k in range(3):
    leg = bench.Leg(args, dev, 1, 0, "bf16", False, 32)
    for i in range(40):
        leg.step(i)
    torch.cuda.synchronize()
    legs.append(leg)
    eng = leg.trainer.engine
    o, d = leg.batches[0][0], leg.batches[0][1]
    eng._fg_compact = False
    full = lambda eng=eng, o=o, d=d: eng._sdf_forward(o, d)
    print("engine %d  H plane at 0x%x (mod 2 MiB: 0x%x)" % (k, eng.w["H"].data_ptr(), eng.w["H"].data_ptr() % (2 << 20)))
    time.sleep(1.0)
    print("   after 1 s idle, 20 launches:          ", fmt(launches(full, 20)))
    t = launches(full, 500)
    print("   500 back to back: 1-10 %6.1f | 11-50 %6.1f | 241-260 %6.1f | 491-500 %6.1f us (medians)" % (
        np.median(t[:10]), np.median(t[10:50]), np.median(t[240:260]), np.median(t[490:])))
    eng._fg_compact = True
    for i in range(200):
        leg.step(40 + i)
    eng._fg_compact = False
    print("   right behind 200 training steps, 20:  ", fmt(launches(full, 20)))
    # the step's own launch (work list) the same three ways
    eng._fg_compact = True
    leg.step(300)
    torch.cuda.synchronize()
    rows = int(eng.w["fg_active"][1].item())
    time.sleep(1.0)
    print("   work-list launch (%d rows), idle:     " % rows, fmt(launches(full, 20)))
    for i in range(200):
        leg.step(301 + i)
    print("   work-list launch, behind 200 steps:   ", fmt(launches(full, 20)))
    situ = leg.sdf_in_situ(60)
    print("   in situ, two streams, 60 steps: mean %.1f median %.1f min %.1f max %.1f us at %.0f rows" % (
        situ["kernel_ms"] * 1e3, situ["kernel_ms_median"] * 1e3, situ["kernel_ms_min"] * 1e3, situ["kernel_ms_max"] * 1e3, situ["points"]))
os.environ.update(VDN_SIDE_STREAM="0", VDN_OVERLAP="0")
leg = bench.Leg(args, dev, 1, 0, "bf16", False, 32)
for i in range(540):
    leg.step(i)
situ = leg.sdf_in_situ(60)
print("one-stream leg, in situ 60 steps: mean %.1f median %.1f min %.1f max %.1f us at %.0f rows" % (
    situ["kernel_ms"] * 1e3, situ["kernel_ms_median"] * 1e3, situ["kernel_ms_min"] * 1e3, situ["kernel_ms_max"] * 1e3, situ["points"]))

# cold L2 / MALL? the step's launch alone, (a) back to back, (b) with 400 MB of unrelated stores between two launches (what the
# background network's training forward leaves in the caches in front of it in the step), (c) with a 3-MB read of the kernel's
# own weight stream by a tiny kernel right before it (torch sum over the blob: one XCD's L2 at best)
eng = leg.trainer.engine
o, d = leg.batches[0][0], leg.batches[0][1]
eng.sdf_probe = None
step_launch = lambda: eng._sdf_forward(o, d)
trash = torch.empty(100 << 20, dtype=torch.float32, device=dev)
blob = eng.nets["sdf"].img.blobs["full"]


def timed(pre, n=20):
    ts = []
    for i in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step_launch()
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in ts]) * 1e3


for i in range(100):
    leg.step(700 + i)
print("one-stream leg, isolated work-list launch (%d rows):" % int(eng.w["fg_active"][1].item()))
print("   back to back:                          ", fmt(timed(lambda: None)))
print("   400 MB of stores in front of each:     ", fmt(timed(lambda: trash.zero_())))
print("   400 MB of stores + blob read in front: ", fmt(timed(lambda: (trash.zero_(), blob.sum()))))
