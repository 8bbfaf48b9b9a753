R=r04
export VDN_SIDE_STREAM=0 VDN_OVERLAP=0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -- python3 bench.py --headline-only --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/${R}_trace_bench.json 2> gpurun_out/${R}_trace.log
find gpurun_out/${R}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_train_bf16_kernel_stats_whole_run.csv
find gpurun_out/${R}_trace -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/steady_stats.py {} 0.3 gpurun_out/${R}_trace_bench.json > gpurun_out/${R}_train_bf16_kernel_stats.csv
rm -rf gpurun_out/${R}_trace
unset VDN_SIDE_STREAM VDN_OVERLAP
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_trace2 -- python3 bench.py --headline-only --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/${R}_trace2_bench.json 2> gpurun_out/${R}_trace2.log
find gpurun_out/${R}_trace2 -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/steady_stats.py {} 0.3 gpurun_out/${R}_trace2_bench.json > gpurun_out/${R}_train_bf16_kernel_stats_two_streams.csv
rm -rf gpurun_out/${R}_trace2
python3 bench.py > gpurun_out/${R}_bench_train_bf16.json 2> gpurun_out/${R}_bench.err
echo traces done
