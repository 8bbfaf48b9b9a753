#!/bin/bash
# GPU call 3 of round 5 (development): full GPU suite on the immediate-offset DMA build, SDF kernel A/B against the per-piece addressing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run3; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
L=vdn-nerf_amd/vdn_hip
for i in 1 2; do
  python tools/dev/sdf_probe.py imm
  VDN_LIB=$L/libvdn_render_dma0.so python tools/dev/sdf_probe.py dma0
done > $O/sdf_probe.txt 2>&1
for i in 1 2; do
  python tools/dev/step_wall.py imm
  VDN_LIB=$L/libvdn_render_dma0.so python tools/dev/step_wall.py dma0_sdf_only
done > $O/step.txt 2>&1
python tools/dev/nerf_probe.py imm > $O/nerf_probe.txt 2>&1
python bench.py --headline-only --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -4 $O/tests.txt; grep -v "Warn\|amdgpu" $O/sdf_probe.txt $O/step.txt $O/nerf_probe.txt
