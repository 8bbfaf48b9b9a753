#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_run11
cd tools/dev && ./sdf2_bench 65536 10 > ../../gpurun_out/r05_run11/sdf2_harness.log 2>&1; cd ../..
grep -E "per-WG|phases|median" gpurun_out/r05_run11/sdf2_harness.log | cut -c1-220
