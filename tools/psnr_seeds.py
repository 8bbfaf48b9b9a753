"""'PSNR vs ref' for the bf16 path as a PAIRED, MULTI-SEED comparison.
A single training run on the synthetic scene cannot resolve 1 dB: held-out views differ by 3-6 dB within one checkpoint and
consecutive checkpoints of one run by +-3 dB (profiles/r04_psnr_*_100k.jsonl: at 100 K steps bf16 29.7 +- 3.7 dB, fp32 24.5 +- 5.7 dB
over 8 views - bf16 ABOVE fp32 by 5 dB - while round 2's 20 K pair had it 1.9 dB below). So: several seeds (initial weights and
pixel stream) per precision, same seeds for both, mean over the held-out views of the mean over the last 3 checkpoints per run.
  psnr_seeds.py <steps> <n_seeds> [--views V] [--rig real]    -> one JSON line: per-run values, mean +- sd per precision, the paired differences"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps, n_seeds = int(sys.argv[1]), int(sys.argv[2])
views = int(sys.argv[sys.argv.index("--views") + 1]) if "--views" in sys.argv else 8
rig = sys.argv[sys.argv.index("--rig") + 1] if "--rig" in sys.argv else "synthetic"
import numpy as np
res = {"bf16": [], "fp32": []}
for seed in range(n_seeds):
    for prec in ("bf16", "fp32"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_psnr.py"), str(steps), prec, "--views", str(views),
                              "--checkpoints", "10", "--seed", str(seed), "--quiet", "--rig", rig], capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            print(out.stderr[-2000:], file=sys.stderr)
            sys.exit(1)
        j = json.loads(line[-1])
        res[prec].append(j["mean_of_last_3_checkpoints"])
        print("seed %d %s: %.2f dB (last checkpoint %.2f +- %.2f over views; %.0f s)" % (
            seed, prec, j["mean_of_last_3_checkpoints"], j["final_val_psnr_mean"], j["final_val_psnr_sd"], j["wall_s"]), flush=True)
b, f = np.array(res["bf16"]), np.array(res["fp32"])
d = b - f
print(json.dumps({"rig": rig, "steps": steps, "seeds": n_seeds, "views": views, "bf16_psnr_per_seed": b.round(3).tolist(), "fp32_psnr_per_seed": f.round(3).tolist(),
                  "bf16_mean": float(b.mean()), "bf16_sd": float(b.std(ddof=1)) if n_seeds > 1 else None,
                  "fp32_mean": float(f.mean()), "fp32_sd": float(f.std(ddof=1)) if n_seeds > 1 else None,
                  "paired_difference_bf16_minus_fp32_mean": float(d.mean()),
                  "paired_difference_sd": float(d.std(ddof=1)) if n_seeds > 1 else None,
                  "paired_difference_standard_error": float(d.std(ddof=1) / np.sqrt(n_seeds)) if n_seeds > 1 else None}))
