#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run15; mkdir -p $O
python -m pytest tests/test_gpu_train_parity.py -m gpu -x -q -k "colour_head" 2>&1 | tail -12 > $O/tests.txt
tail -6 $O/tests.txt
