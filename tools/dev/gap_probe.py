"""Where the streams of a training step wait, without a profiler: a one-thread stamp kernel (tools/dev/stamp.hip, 100-MHz clock)
in front of and behind every library call of a few steady-state steps. Per call: queue (stream handle), time the stamp in front
ran (= the stream's previous work and every wait_event in front of the call were done), duration, and the gap to the end of the
previous call on the same stream.  usage: gap_probe.py [steps=3] [wdepth]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse
import numpy as np
import torch
import bench
from vdn_hip import lib
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wdepth = len(sys.argv) > 2 and sys.argv[2] == "wdepth"
leg = bench.Leg(argparse.Namespace(batch=512), torch.device("cuda:0"), 1, 0, "bf16", wdepth, 64)
for i in range(700):
    leg.step(i)
torch.cuda.synchronize()
st = ctypes.CDLL(os.path.join(ROOT, "tools", "dev", "_build", "libstamp.so"))
st.dev_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
slots = torch.zeros(8192, dtype=torch.int64, device="cuda:0")
log, nxt = [], [0]
orig_call, orig_try = lib.call, lib.try_call
def wrap(orig):
    def f(name, *args, **kw):
        s = args[-1]
        if not isinstance(s, int) or nxt[0] + 2 > slots.numel():
            return orig(name, *args, **kw)
        i = nxt[0]; nxt[0] += 2
        st.dev_stamp(slots.data_ptr() + 8 * i, s)
        r = orig(name, *args, **kw)
        st.dev_stamp(slots.data_ptr() + 8 * (i + 1), s)
        log.append((name, s, i))
        return r
    return f
lib.call, lib.try_call = wrap(orig_call), wrap(orig_try)
import vdn_hip.train, vdn_hip.images, vdn_train.trainer, dpt_models.renderer, dpt_models.fields
marks = []
for i in range(steps):
    marks.append(len(log))
    leg.step(700 + i)
torch.cuda.synchronize()
lib.call, lib.try_call = orig_call, orig_try
t = slots.cpu().numpy()
b = marks[-2] if steps > 1 else 0             # the second-to-last step's calls, plus whatever of the last step overlaps
e = marks[-1] if steps > 1 else len(log)
t0 = t[log[b][2]]
qs, last_end = {}, {}
print("      start      dur      gap  stream  call")
for name, s, i in log[b:e + 12]:
    q = qs.setdefault(s, len(qs))
    a0, a1 = (t[i] - t0) / 100.0, (t[i + 1] - t0) / 100.0
    gap = a0 - last_end.get(s, a0)
    last_end[s] = a1
    print("%10.1f %8.1f %8.1f  q%d %s%s" % (a0, a1 - a0, gap, q, "      " * q, name))
print("step (first call to first call of the next step): %.1f us" % ((t[log[e][2]] - t0) / 100.0))
