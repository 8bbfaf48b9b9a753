"""The unchanged-runner flow (bench.py::runner_flow) alone, for a rocprofv3 kernel trace: wall per step vs kernel time per step.
usage: runner_probe.py [bf16|fp32] [steps] [wdepth]"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
import torch
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
args = types.SimpleNamespace(batch=512)
print(bench.runner_flow(args, torch.device("cuda:0"), prec, steps, wdepth=len(sys.argv) > 3))
