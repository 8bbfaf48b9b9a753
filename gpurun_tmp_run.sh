#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run16; mkdir -p $O
python -m pytest tests/test_gpu_shade_fused.py tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -5 $O/tests.txt; tail -2 $O/smoke.txt
