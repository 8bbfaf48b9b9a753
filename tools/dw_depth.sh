#!/bin/bash
# timing sweep of the dW GEMM's prefetch depth (VDN_DW_DEPTH; 1 = two workgroups per CU with one stage in flight each)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT/vdn-nerf_amd"
for D in ${@:-1 2 3 4}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../include -I csrc -DVDN_DW_DEPTH=$D -c csrc/train_dw_bf16.hip -o vdn_hip/_build/train_dw_bf16.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o vdn_hip/libvdn_render.so vdn_hip/_build/*.o
  (cd .. && python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('DEPTH=$D', d['ms_per_step'], d['roofline_dw_gemm']['kernel_ms'], d['roofline_dw_gemm']['achieved'])")
done
