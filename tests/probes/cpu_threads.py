"""CPU-baseline calibration: the oracle's training step (bench.py's cpu_baseline leg) at different thread counts on this host.
  python tests/probes/cpu_threads.py        -> rays/s per thread count (the bench uses the fastest, 16 on the 2 x EPYC GPU host)"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "vdn-nerf_amd"), ROOT):
    sys.path.insert(0, p)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
for T in (8, 16, 32, 64, 128):
    if T > (os.cpu_count() or 1):
        break
    os.environ["VDN_CPU_THREADS"] = str(T)
    t = time.time()
    r = bench.cpu_baseline(512, 0, budget_s=0.0)
    print(T, "threads: %.1f rays/s" % r["value"], "(%s; %.1f s)" % (r["sample"].split(" of ")[0], time.time() - t), flush=True)
