#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run21; mkdir -p $O
for i in 1 2 3; do
  python tools/dev/step_wall.py normal
  VDN_MAIN_PRIORITY=1 python tools/dev/step_wall.py main_high
done > $O/step.txt 2>&1
for i in 1 2; do
  python tools/dev/step_wall.py normal_wdepth 40 6 wdepth
  VDN_MAIN_PRIORITY=1 python tools/dev/step_wall.py main_high_wdepth 40 6 wdepth
  python tools/dev/step_wall.py normal_crop 40 6 white 420
  VDN_MAIN_PRIORITY=1 python tools/dev/step_wall.py main_high_crop 40 6 white 420
done >> $O/step.txt 2>&1
VDN_MAIN_PRIORITY=1 python bench.py --no-cpu-baseline > $O/bench_prio.json 2> $O/bench_prio.err
python bench.py --no-cpu-baseline > $O/bench_normal.json 2> $O/bench_normal.err
grep -v "Warn\|amdgpu" $O/step.txt
python - <<'PY'
import json
for n in ("prio","normal"):
    d=json.load(open('gpurun_out/r05_run21/bench_%s.json'%n))
    print(n, round(d['value']), round(d['ms_per_step'],4), {k:round(d[k]['ms_per_step'],3) for k in ('all_samples_evaluated','parity_path','wdepth','object_centric','real_cameras')})
PY
