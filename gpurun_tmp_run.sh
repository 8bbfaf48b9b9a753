#!/bin/bash
# GPU call 6 of round 5 (development): is the real-camera leg's 1.46 ms the rig or the process's stream count? + full suite on the shared streams
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run6; mkdir -p $O
VDN_REAL_CAMS=1 python tools/dev/step_wall.py real_cams > $O/step.txt 2>&1
python tools/dev/step_wall.py synth_cams >> $O/step.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt
grep -v "Warn\|amdgpu" $O/step.txt; tail -3 $O/tests.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_run6/bench.json'))
print('headline', d['value'], d['ms_per_step'])
for k in ('all_samples_evaluated','parity_path','wdepth','object_centric','real_cameras'):
    print(k, d[k]['value'], d[k]['ms_per_step'])
print(d['runner_flow']['bf16']['rays_per_s'], d['runner_flow']['fp32']['rays_per_s'])
PY
