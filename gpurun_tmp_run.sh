#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run8; mkdir -p $O
python tools/dev/dw_split_sweep.py > $O/sweep.txt 2>&1
grep -v "Warn\|amdgpu" $O/sweep.txt
