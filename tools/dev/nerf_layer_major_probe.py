"""Layer-major backward of the background network's pts_linears (reference fields.py:324-353 and its autograd) as a PROTOTYPE on
library GEMMs: for l = 7 .. 1

    dW_l      = delta_l^T  h_{l-1}                 [256 x Q] x [Q x 256]      (f32 out)
    delta_l-1 = (delta_l  W_l) * (h_{l-1} > 0)      [Q x 256] x [256 x 256], ReLU mask

each as its own launch(es) on row-major bf16 planes, so that delta_l and h_{l-1} are read by the dW product right behind the launch
that wrote delta_l (33 MB per plane at the bench's 65.7 K-row work list: inside the 256-MB Infinity Cache). Timed back to back on one
stream with HIP events: per product, per layer, and the whole 7-layer pass; beside it the two launches it would replace, timed
the same way on the same box through the training engine (csrc/k_nerf_bwd.h point-major chain + the background network's group of
the split-K weight-gradient GEMM), at the same row count.
usage: nerf_layer_major_probe.py [rows=65710] [rounds=30]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import numpy as np, torch

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 65710
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
bf = torch.bfloat16


def timeit(fn, rounds=R):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


# planes of the 8 hidden layers (row-major [Q, 256] bf16: what a library GEMM wants), weights [out, in]
h = [torch.randn(Q, 256, device=dev).relu_().to(bf) for _ in range(8)]
W = [(torch.randn(256, 256, device=dev) / 16).to(bf) for _ in range(8)]
delta = [torch.empty(Q, 256, device=dev, dtype=bf) for _ in range(8)]
delta[7].copy_(torch.randn(Q, 256, device=dev))
dW = [torch.empty(256, 256, device=dev, dtype=torch.float32) for _ in range(8)]
tmp = torch.empty(Q, 256, device=dev, dtype=bf)


def dx(l):          # delta_{l-1} = (delta_l W_l) * relu'(h_{l-1})
    torch.mm(delta[l], W[l], out=tmp)
    torch.mul(tmp, h[l - 1] > 0, out=delta[l - 1])


def dx_gemm(l):
    torch.mm(delta[l], W[l], out=tmp)


def dw(l):          # dW_l = delta_l^T h_{l-1}   (bf16 operands; the library accumulates in f32, output cast here)
    dW[l].copy_(torch.mm(delta[l].t(), h[l - 1]))


def dw_gemm(l):
    torch.mm(delta[l].t(), h[l - 1])


def layer_major():
    for l in range(7, 0, -1):
        dw(l)
        dx(l)


print("rows %d, planes %.1f MB each (bf16), device %s" % (Q, Q * 256 * 2 / 1e6, torch.cuda.get_device_name(0)))
t = timeit(lambda: dx_gemm(7)); print("dX product  [Q,256] x [256,256]            median %7.1f us  min %7.1f   (%.0f TFLOP/s)" % (t[0], t[1], 2.0 * Q * 256 * 256 / t[0] / 1e6))
t = timeit(lambda: dx(7));      print("dX product + ReLU mask (2 more launches)   median %7.1f us  min %7.1f" % t)
t = timeit(lambda: dw_gemm(7)); print("dW product  [256,Q] x [Q,256]              median %7.1f us  min %7.1f   (%.0f TFLOP/s)" % (t[0], t[1], 2.0 * Q * 256 * 256 / t[0] / 1e6))
t = timeit(lambda: dw(7));      print("dW product + f32 store                     median %7.1f us  min %7.1f" % t)
t = timeit(lambda: (dw(7), dx(7))); print("one layer (dW then dX on the plane just read) median %7.1f us  min %7.1f" % t)
t_lm = timeit(layer_major)
print("layer-major pass, 7 layers x (dW + dX + mask) on library GEMMs: median %7.1f us  min %7.1f" % t_lm)

# ---- what it would replace: the point-major chain + the background group of the split-K GEMM, through the engine
from vdn_train import synth, factory
from vdn_hip import lib
from vdn_hip.train import TrainEngine
rend = factory.build_renderer(device=dev, states=synth.make_all_states(0), precision="bf16")
B = 512
cams = synth.make_cameras(0)
o, d = synth.random_pixel_batch(0, 0, 0, B, cams=cams)
near, far = synth.near_far_from_sphere(o, d)
g = lambda x: torch.tensor(x).to(dev)
params = rend._all_parameters()
os.environ["VDN_RENDER_GRAPHS"] = "0"
out = rend.render(g(o), g(d), g(near), g(far), background_rgb=torch.ones(1, 3, device=dev), cos_anneal_ratio=0.5)
(out["color_fine"].sum() + out["gradient_error"]).backward()
eng = next(iter(rend.__dict__["_engines"].values()))
w = eng.w
rows_bg = int(w["bg_active"][1].item())
st = lib.stream_handle()


def chain():
    nb = lib.VdnNerfBwdArgs()
    nb.blob = eng.nets["nerf"].img.blobs["bwd"].data_ptr()
    nb.g_density, nb.g_rgb = w["d_bg_density"].data_ptr(), w["d_bg_rgb"].data_ptr()
    nb.save_h, nb.save_hv = w["nf_h"].data_ptr(), w["nf_hv"].data_ptr()
    nb.delta_o, nb.delta_v, nb.delta_head, nb.delta_h = (w[k].data_ptr() for k in ("nf_do", "nf_dv", "nf_dhead", "nf_dh"))
    nb.P = eng.Q
    nb.active_idx, nb.n_active = w["bg_active"][0].data_ptr(), w["bg_active"][1].data_ptr()
    lib.call("vdn_nerf_mlp_bwd_bf16", nb, lib.stream_handle())


def gemm():
    tab, n, wgs = eng.dw_groups["nerf"]
    lib.call("vdn_dw_gemm_bf16", lib.ptr(tab), n, wgs, lib.stream_handle())


print("\nthe shipped pair on this batch's work list (%d of %d background rows; the chain and the GEMM cover ALL 13 matrices of the" % (rows_bg, eng.Q))
print("network - 8 pts_linears, the heads and the view branch - the prototype above only the 7 square pts_linears products):")
t_c = timeit(chain); print("vdn_nerf_mlp_bwd_bf16 (point-major chain: every delta plane written once, h planes read once)  median %7.1f us  min %7.1f" % t_c)
t_g = timeit(gemm);  print("vdn_dw_gemm_bf16, background group (split-K over %d rows)                                     median %7.1f us  min %7.1f" % ((rows_bg,) + t_g))
t_b = timeit(lambda: (chain(), gemm())); print("both back to back                                                                          median %7.1f us  min %7.1f" % t_b)
scale = rows_bg / float(Q)
print("\nlayer-major prototype scaled to the same rows: %.1f us for 7 of the 13 matrices, against %.1f us for all 13 in the shipped pair" % (t_lm[0] * scale, t_b[0]))
