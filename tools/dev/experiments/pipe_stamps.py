"""Diagnostic build only (VDN_PIPE_STAMP=1 python -m ... build): per-stage cycle shares of the pipelined SDF backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import argparse, numpy as np, torch, bench
leg = bench.Leg(argparse.Namespace(batch=512), torch.device("cuda:0"), 1, 0, "bf16", False, 32)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    leg.step(i)
torch.cuda.synchronize()
eng = leg.trainer.engine
NL, NS = eng.pipe["lanes"], 17
sy = eng.pipe["sync"].cpu().numpy()
st = sy[2 + NS * NL + 2:].view(np.int64)[:NS * NL * 8].reshape(NS, NL, 8)
print("status", sy[1], "ticket", sy[0])
print("stage  total  vmcnt  barrier  wait  chain+epi  dw   blocks   (median over lanes, kilo-cycles)")
for s in range(NS):
    m = np.median(st[s], axis=0) / 1e3
    print("%5d %6.0f %6.0f %7.0f %6.0f %8.0f %6.0f %6.0f" % (s, m[0], m[1], m[2], m[3], m[4], m[5], np.median(st[s][:, 6])))
