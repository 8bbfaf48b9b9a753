import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vdn_train import synth, factory
dev = torch.device("cuda:0")
for prec in ("fp32", "bf16"):
    rend = factory.build_renderer(device=dev, states=synth.make_all_states(1), precision=prec)
    pts = torch.tensor(synth.uniform(1, "dbg", (256, 3)).astype(np.float32) - 0.5).to(dev)
    s0 = rend.sdf_network.sdf(pts)
    out = rend.sdf_network(pts)
    img = rend.sdf_network._images()
    torch.cuda.synchronize()
    print(prec, "sdf mode0", s0[:4, 0].tolist(), "mode1", out[:4, 0].tolist(), "feat abs mean", out[:, 1:].abs().mean().item())
    for k, b in img.blobs.items():
        print("   blob", k, b.numel(), "nonzero bytes", int((b != 0).sum().item()))
    print("   weff abs mean", img.weff.abs().mean().item())
