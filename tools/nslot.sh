#!/bin/bash
# timing sweep of the weight-ring depth (VDN_NSLOT) on the SDF kernels; run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT/vdn-nerf_amd"
for G in ${@:-3 4 5}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../include -I csrc -DVDN_NSLOT=$G -c csrc/sdf_bf16.hip -o vdn_hip/_build/sdf_bf16.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o vdn_hip/libvdn_render.so vdn_hip/_build/*.o
  for cfg in "sdf0 8192" "sdf1 65536" "sdf1t 65536"; do
    set -- $cfg
    (cd .. && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ns_${G}_$1_$2 -- python tools/kernel_loop.py $1 $2 bf16 8 > /dev/null 2>&1
     f=$(find gpurun_out/ns_${G}_$1_$2 -name "*kernel_stats.csv" | head -1); echo "NSLOT=$G $1 $2 $(grep sdf_fwd "$f" | awk -F, '{print $(NF-4)}' | tr '\n' ' ')")
  done
done
