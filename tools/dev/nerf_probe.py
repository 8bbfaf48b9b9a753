"""Background-network kernels at the bench's steady-state work list, timed alone (HIP events, back to back) and inside the
one-stream step: forward with saves, forward without (inference), backward. For A/B runs of library variants (VDN_LIB).
usage: nerf_probe.py [tag]"""
import os, sys
os.environ.setdefault("VDN_SIDE_STREAM", "0")
os.environ.setdefault("VDN_OVERLAP", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse
import numpy as np
import torch
import bench
from vdn_hip import lib
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
leg = bench.Leg(argparse.Namespace(batch=512), torch.device("cuda:0"), 1, 0, "bf16", False, 64)
for i in range(700):
    leg.step(i)
torch.cuda.synchronize()
eng, rend = leg.trainer.engine, leg.rend
w = eng.w
st = torch.cuda.current_stream().cuda_stream
rows = int(w["bg_active"][1].item())
o, d = leg.batches[(699) % 64][0], leg.batches[699 % 64][1]

def nerf_args(save):
    n = lib.VdnNerfArgs()
    n.blob = eng.nets["nerf"].img.blobs["fwd"].data_ptr()
    n.rays_o, n.rays_d, n.z, n.n_per_ray, n.P = o.data_ptr(), d.data_ptr(), w["bg_mid"].data_ptr(), eng.T, eng.Q
    n.density, n.rgb = w["bg_density"].data_ptr(), w["bg_rgb"].data_ptr()
    if save:
        n.save_h, n.save_pe, n.save_feature, n.save_vpe, n.save_hv = (w[k].data_ptr() for k in ("nf_h", "nf_pe", "nf_feature", "nf_vpe", "nf_hv"))
    n.active_idx, n.n_active = w["bg_active"][0].data_ptr(), w["bg_active"][1].data_ptr()
    return n
a_save, a_inf = nerf_args(True), nerf_args(False)
nb = lib.VdnNerfBwdArgs()
nb.blob = eng.nets["nerf"].img.blobs["bwd"].data_ptr()
nb.g_density, nb.g_rgb = w["d_bg_density"].data_ptr(), w["d_bg_rgb"].data_ptr()
nb.save_h, nb.save_hv = w["nf_h"].data_ptr(), w["nf_hv"].data_ptr()
nb.delta_o, nb.delta_v, nb.delta_head, nb.delta_h = (w[k].data_ptr() for k in ("nf_do", "nf_dv", "nf_dhead", "nf_dh"))
nb.P = eng.Q
nb.active_idx, nb.n_active = w["bg_active"][0].data_ptr(), w["bg_active"][1].data_ptr()
trash = torch.empty(400 * 1000 * 1000 // 4, dtype=torch.float32, device="cuda:0")

def timed(fn, cold, n=15):
    ts = []
    for i in range(n + 2):
        if cold:
            trash.add_(1.0)          # 400 MB read + 400 MB written in front: the launch finds the caches as a step leaves them
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), min(ts), max(ts)
res = {}
for name, fn in (("fwd_save", lambda: lib.call("vdn_nerf_mlp_fwd_bf16", a_save, st)), ("fwd_inference", lambda: lib.call("vdn_nerf_mlp_fwd_bf16", a_inf, st)),
                 ("bwd", lambda: lib.call("vdn_nerf_mlp_bwd_bf16", nb, st))):
    for cold in (False, True):
        res[name + ("_cold" if cold else "")] = timed(fn, cold)
# the step (one stream) around it
import time
torch.cuda.synchronize()
t0 = time.time()
for i in range(120):
    leg.step(700 + i)
torch.cuda.synchronize()
step_us = (time.time() - t0) / 120 * 1e6
print("%-10s rows %d  " % (tag, rows) + "  ".join("%s %.1f (%.1f-%.1f)" % ((k,) + v) for k, v in res.items()) + "  one-stream step %.1f us" % step_us)
