cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python3 bench.py --headline-only --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/tl.log 2>&1
f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
python3 tools/dev/timeline.py $f -600 > gpurun_out/tl_step.txt
python3 tools/dev/timeline.py $f -601 > gpurun_out/tl_step5.txt
rm -rf gpurun_out/tl
