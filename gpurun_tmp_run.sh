#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run18; mkdir -p $O
python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_variants.py tests/test_gpu_shade_fused.py -m gpu -x -q 2>&1 | grep -E "^E |^>|passed|failed|Error" | head -30 > $O/tests.txt
cat $O/tests.txt
