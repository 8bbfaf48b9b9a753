#!/bin/bash
# Everything profiles/ cites for one round, collected on the GPU box (run from the repo root through gpurun):
#   counters (SQ shares, FETCH/WRITE) of the fused SDF kernel's inference and training launches and of the weight-gradient
#   GEMM, and the rocprofv3 kernel trace of the headline bench command (side stream off: concurrent kernels inflate each
#   other's durations in a trace). Summaries are written under gpurun_out/<round>_*; tools/summarise_counters.py and
#   tools/traffic_json.py turn them into the files committed under profiles/.
#   usage: bash tools/profile_round.sh r02
R=${1:-r05}
export VDN_SIDE_STREAM=0 VDN_OVERLAP=0
bash tools/collect_counters.sh ${R}_sdf1 python3 tools/kernel_loop.py sdf1 65536 bf16 8
bash tools/collect_counters.sh ${R}_sdf1t python3 tools/kernel_loop.py sdf1t 65536 bf16 8
bash tools/collect_counters.sh ${R}_step python3 bench.py --headline-only --no-cpu-baseline --steps 10
# the inference path: a loop of render() calls on full-frame 512-ray batches - the one-launch shading kernel sdf_fwd2_kernel<2,...>
# (vdn_shade_fused_bf16, the north star's single launch), the sampler's passes, the background network
bash tools/collect_counters.sh ${R}_fwd python3 tools/dev/render_loop.py 60 512 0
find gpurun_out/${R}_fwd/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_forward_kernel_stats.csv
for t in sdf1 sdf1t step fwd; do
  python3 tools/summarise_counters.py gpurun_out/${R}_$t > gpurun_out/${R}_$t/summary.json
  rm -rf gpurun_out/${R}_$t/pass*/ gpurun_out/${R}_$t/trace          # raw per-dispatch csv files: tens of MB; the summary keeps the means
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# kernel trace of the headline command with nothing but the timed step's own launches in it (--no-roofline), reduced to the
# steady-state tail (tools/steady_stats.py: the work lists shrink over the first ~600 steps; `value` is decided behind that)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -- python3 bench.py --headline-only --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/${R}_trace_bench.json 2> gpurun_out/${R}_trace.log
find gpurun_out/${R}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_train_bf16_kernel_stats_whole_run.csv
find gpurun_out/${R}_trace -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/steady_stats.py {} 0.3 gpurun_out/${R}_trace_bench.json > gpurun_out/${R}_train_bf16_kernel_stats.csv
rm -rf gpurun_out/${R}_trace
# the same with both streams in use (the default schedule): kernels overlap, their durations inflate each other
unset VDN_SIDE_STREAM VDN_OVERLAP
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_trace2 -- python3 bench.py --headline-only --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/${R}_trace2_bench.json 2> gpurun_out/${R}_trace2.log
find gpurun_out/${R}_trace2 -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/steady_stats.py {} 0.3 gpurun_out/${R}_trace2_bench.json > gpurun_out/${R}_train_bf16_kernel_stats_two_streams.csv
rm -rf gpurun_out/${R}_trace2
echo profile_round done
# the bench line of the same box, and SURVEY.md 8d's whole CPU protocol (3 warm-up + 5 timed batches per figure: minutes of host time)
python3 bench.py > gpurun_out/${R}_bench_train_bf16.json 2> gpurun_out/${R}_bench.err
python3 bench.py --headline-only --no-roofline --cpu-baseline-full > gpurun_out/${R}_bench_cpu_baseline_full.json 2> gpurun_out/${R}_bench_cpu_full.err
echo profile_round complete
