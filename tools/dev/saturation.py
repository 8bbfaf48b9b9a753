"""How saturated is Softplus(beta=100) in the SDF network, per value and per the groups a kernel could skip (dev)?
A hidden activation is saturated when |t| > 25 in the kernel's units (t = 100 log2(e) a): then g = max(t, 0) exactly in f32.
Groups: one accumulator register of one wave = 32 consecutive rows x 2 features (f, f+4); one tile = 32 rows x 32 features."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import torch
import bench
from vdn_hip import layout

class A: pass
a = A(); a.batch = 512
dev = torch.device("cuda", 0)
leg = bench.Leg(a, dev, 1, 0, "bf16", False, 8)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    leg.step(i)
torch.cuda.synchronize()
eng = leg.trainer.engine
n = int(eng.w["fg_active"][1].item())
Pp = eng.Pp
H = eng.w["H"].view(8, Pp * 256)
tot = sat = 0
reg_all = reg_sat = tile_all = tile_sat = 0
for l in range(8):
    nc = 224 if l == 3 else 256
    g = layout.from_pt32(H[l], Pp, 256)[: (n // 32) * 32, :nc].float()
    s = (g > 25.0) | (g < 3e-8)
    tot += s.numel(); sat += int(s.sum())
    blk = s.view(-1, 32, nc)                                  # [blocks, 32 rows, features]
    # register groups: features f and f+4 with (f % 8) < 4
    f = torch.arange(nc, device=dev)
    base = f[(f % 8) < 4]
    pair = blk[:, :, base] & blk[:, :, base + 4]
    r = pair.all(dim=1)
    reg_all += r.numel(); reg_sat += int(r.sum())
    t = blk.view(blk.shape[0], 32, nc // 32, 32).all(dim=3).all(dim=1)
    tile_all += t.numel(); tile_sat += int(t.sum())
print("rows %d: saturated values %.1f %%, fully saturated registers (32 rows x 2 features) %.2f %%, fully saturated tiles %.3f %%"
      % (n, 100.0 * sat / tot, 100.0 * reg_sat / reg_all, 100.0 * tile_sat / tile_all))
