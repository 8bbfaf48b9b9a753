#!/bin/bash
# timing-only ablations of the forward chain (results are wrong by construction): 0 = real kernel, 1 = no softplus,
# 2 = no MFMA, 3 = no chunk barrier
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT/vdn-nerf_amd"
for v in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../include -I csrc -DVDN_ABLATE=$v -c csrc/sdf_bf16.hip -o vdn_hip/_build/sdf_bf16.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o vdn_hip/libvdn_render.so vdn_hip/_build/*.o
  (cd .. && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$v -- python tools/kernel_loop.py sdf0 65536 bf16 8 > /dev/null 2>&1; f=$(find gpurun_out/abl_$v -name "*kernel_stats.csv" | head -1); echo "ablate=$v $(grep sdf_fwd "$f" | cut -c1-200)")
done
