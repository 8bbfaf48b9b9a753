cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export VDN_SIDE_STREAM=0 VDN_OVERLAP=0
rocprofv3 --kernel-trace --stats -d gpurun_out/fp -o fp --output-format csv -- python3 bench.py --precision fp32 --headline-only --no-cpu-baseline --no-roofline --steps 10 > gpurun_out/fp.log 2>&1
tail -c 400 gpurun_out/fp.log
f=$(find gpurun_out/fp -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/fp32_kernel_stats.csv
rm -rf gpurun_out/fp
