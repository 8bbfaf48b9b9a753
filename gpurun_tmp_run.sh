#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run12; mkdir -p $O
L=vdn-nerf_amd/vdn_hip
for v in base w8s3 w8s4 w8s5 base; do
  if [ $v = base ]; then python tools/dev/nerf_probe.py $v; else VDN_LIB=$L/libvdn_render_$v.so python tools/dev/nerf_probe.py $v; fi
done > $O/nerf_probe.txt 2>&1
for i in 1 2; do
  python tools/dev/step_wall.py base
  VDN_LIB=$L/libvdn_render_w8s4.so python tools/dev/step_wall.py w8s4
done > $O/step.txt 2>&1
grep -v "Warn\|amdgpu" $O/nerf_probe.txt $O/step.txt
