"""profiles/<round>_* from what tools/profile_round.sh left under gpurun_out/ (dev; run from the repo root):
python tools/dev/make_profiles.py r03"""
import json, os, shutil, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
G, P = "gpurun_out", "profiles"
line = open("%s/%s_bench_train_bf16.json" % (G, R)).read().strip().splitlines()[-1]
json.loads(line)
open("%s/%s_bench_train_bf16.json" % (P, R), "w").write(line + "\n")
for f in ("train_bf16_kernel_stats_whole_run.csv", "step_timeline_two_streams.txt", "forward_kernel_stats.csv"):
    if os.path.exists("%s/%s_%s" % (G, R, f)):          # (the two steady-state tables are copied by hand: their headers are edited)
        shutil.copy("%s/%s_%s" % (G, R, f), "%s/%s_%s" % (P, R, f))
summ = {t: json.load(open("%s/%s_%s/summary.json" % (G, R, t))) for t in ("sdf1", "sdf1t", "step", "fwd")}

def pick(d, key):
    ks = [k for k in d if key in k]
    assert ks, (key, list(d))
    k = max(ks, key=lambda n: d[n].get("kernel_trace_us", {}).get("mean", 0) * d[n].get("kernel_trace_us", {}).get("n", 0))
    return k, d[k]
for tag, src, key in (("counters_sdf_fwd_sdf1", "sdf1", "sdf_fwd2_kernel<1, false"), ("counters_sdf_fwd_sdf1t", "sdf1t", "sdf_fwd2_kernel<1, true"),
                      ("counters_dw_gemm", "step", "dw_gemm_bf16"), ("counters_shade_fused", "fwd", "sdf_fwd2_kernel<2,")):
    k, v = pick(summ[src], key)
    json.dump({k: v}, open("%s/%s_%s.json" % (P, R, tag), "w"), indent=1)
json.dump(summ["step"], open("%s/%s_counters_step.json" % (P, R), "w"), indent=1)
open("%s/%s_counters_step_table.txt" % (P, R), "w").write(
    subprocess.run([sys.executable, "tools/dev/counters_table.py", "%s/%s_counters_step.json" % (P, R), "16"], capture_output=True, text=True, check=True).stdout)

def traffic(src, key, note, out, launches=1):
    r = subprocess.run([sys.executable, "tools/traffic_json.py", "%s/%s_%s/summary.json" % (G, R, src), key, note], capture_output=True, text=True, check=True)
    j = json.loads(r.stdout)
    if launches != 1:
        j["mean_bytes_of_one_launch"] = j["hbm_bytes_per_launch"]
        for f in ("fetch_bytes_x2", "write_bytes", "hbm_bytes_per_launch"):
            j[f] *= launches
    json.dump(j, open("%s/%s_%s.json" % (P, R, out), "w"), indent=1)
traffic("sdf1", "sdf_fwd2_kernel<1, false", "fused SDF kernel, inference launch on 65 536 points (tools/kernel_loop.py sdf1).", "traffic_sdf_fwd_bf16")
traffic("sdf1t", "sdf_fwd2_kernel<1, true", "fused SDF kernel, training-mode launch (saves) on 65 536 rows (tools/kernel_loop.py sdf1t).", "traffic_sdf_fwd_bf16_train")
traffic("step", "dw_gemm_bf16", "weight-gradient GEMM in the bench step, all launches of a step together: the counter runs use the in-order schedule (VDN_OVERLAP=0), "
        "whose two launches (SDF entries + the rest) are 2 x the mean per dispatch; the default schedule issues the rest as two launches "
        "(background network, heads) with the same bytes. The run includes the first ~600 steps' larger work lists (the steady state's lists "
        "are ~15 % smaller).", "traffic_dw_gemm_bf16", launches=2)
traffic("fwd", "sdf_fwd2_kernel<2,", "the one-launch shading kernel (vdn_shade_fused_bf16: PE -> SDF MLP -> gradient sweep -> colour head -> NeuS alpha / compositing) "
        "on the 65 536 inside samples of a 512-ray full-frame batch, inside a loop of render() calls (tools/dev/render_loop.py 60 512 0).", "traffic_shade_fused_bf16")
json.dump(summ["fwd"], open("%s/%s_counters_forward.json" % (P, R), "w"), indent=1)
print("profiles/%s_* written" % R)
