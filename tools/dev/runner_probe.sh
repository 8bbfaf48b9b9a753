cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/dev/runner_probe.py bf16 20 2>&1 | tail -1
python3 tools/dev/runner_probe.py fp32 8 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d gpurun_out/rp -o rp --output-format csv -- python3 tools/dev/runner_probe.py bf16 20 > gpurun_out/rp.log 2>&1
tail -1 gpurun_out/rp.log
f=$(find gpurun_out/rp -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/runner_flow_kernel_stats.csv
rm -rf gpurun_out/rp
