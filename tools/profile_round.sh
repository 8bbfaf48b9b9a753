#!/bin/bash
# Everything profiles/ cites for one round, collected on the GPU box (run from the repo root through gpurun):
#   counters (SQ shares, FETCH/WRITE) of the fused SDF kernel's inference and training launches and of the weight-gradient
#   GEMM, and the rocprofv3 kernel trace of the headline bench command (side stream off: concurrent kernels inflate each
#   other's durations in a trace). Summaries are written under gpurun_out/<round>_*; tools/summarise_counters.py and
#   tools/traffic_json.py turn them into the files committed under profiles/.
#   usage: bash tools/profile_round.sh r02
R=${1:-r02}
export VDN_SIDE_STREAM=0
bash tools/collect_counters.sh ${R}_sdf1 python3 tools/kernel_loop.py sdf1 65536 bf16 8
bash tools/collect_counters.sh ${R}_sdf1t python3 tools/kernel_loop.py sdf1t 65536 bf16 8
bash tools/collect_counters.sh ${R}_step python3 bench.py --headline-only --no-cpu-baseline --steps 10
for t in sdf1 sdf1t step; do
  python3 tools/summarise_counters.py gpurun_out/${R}_$t > gpurun_out/${R}_$t/summary.json
  rm -rf gpurun_out/${R}_$t/pass*/ gpurun_out/${R}_$t/trace          # raw per-dispatch csv files: tens of MB; the summary keeps the means
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -- python3 bench.py --headline-only --no-cpu-baseline --steps 20 > gpurun_out/${R}_trace_bench.json 2> gpurun_out/${R}_trace.log
find gpurun_out/${R}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_train_bf16_kernel_stats.csv
rm -rf gpurun_out/${R}_trace
echo profile_round done
